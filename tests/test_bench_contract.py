"""bench.py's command line (the driver's contract) and its refusal to run without a GPU: the library has no CPU path
and the bench must say so instead of measuring something else."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_the_driver_passes_exist():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


@pytest.mark.skipif(torch.cuda.is_available(), reason="this is the no-GPU behaviour")
def test_refuses_to_run_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a GPU" in (out.stderr + out.stdout)
    assert "{" not in out.stdout                                    # no JSON line: nothing was measured


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
