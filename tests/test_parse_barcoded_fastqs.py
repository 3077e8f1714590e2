"""ParseBarcodedFastqs: stLFR fastq.gz pairs -> barcode-sorted .fastb/.qualp/.bci, byte for byte against
the reference's own binary -- oracle/_ref/ParseBarcodedFastqs, built from 10X/ParseBarcodedFastqs.cc where it lies
(flags only) -- when it is there (the build container), and against the fixture it wrote (tests/golden/pbf/) anywhere.
These run the program's HOST=True path (its own host code for the pair order, the packing and the PQVec encoder), on the
CPU; the default path -- the same three steps on the MI355X (dfk_pbf_run) -- is compared with both in tests/test_gpu_pbf.py."""
import os
import subprocess

import numpy as np
import pytest

from superplus_amd import feudal
from tests.fastq_synth import make_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "superplus_amd", "ParseBarcodedFastqs")
REF = os.path.join(ROOT, "oracle", "_ref", "ParseBarcodedFastqs")


def run(binary, fq1, fq2, head, *extra, device=False):
    host = ["HOST=True"] if binary == OURS and not device else []
    return subprocess.run([binary, "FASTQS={" + fq1 + "," + fq2 + "}", "OUT_HEAD=" + head, *extra, *host], capture_output=True, text=True, timeout=600)


def same_files(a, b):
    for ext in ("fastb", "qualp", "bci"):
        assert open(f"{a}.{ext}", "rb").read() == open(f"{b}.{ext}", "rb").read(), ext


def test_fixture_written_by_the_reference_binary(tmp_path, golden_dir):
    g = os.path.join(golden_dir, "pbf")
    r = run(OURS, g + "/r_1.fq.gz", g + "/r_2.fq.gz", f"{tmp_path}/o/reads", "NUM_BUCKETS=4", "NUM_THREADS=3")
    assert r.returncode == 0, r.stderr
    same_files(f"{tmp_path}/o/reads", g + "/reads")
    # and what it wrote is a valid DF input: pairs, barcode 0 first, index ascending and complete
    _, _, rlen = feudal.read_fastb(f"{tmp_path}/o/reads.fastb")
    bci = feudal.read_bci(f"{tmp_path}/o/reads.bci")
    assert bci[0] == 0 and bci[-1] == len(rlen) == 600 and np.all(np.diff(bci) >= 0) and np.all(bci % 2 == 0)


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference binary exists in the build container only")
@pytest.mark.parametrize("pairs,seed,n_bc,ragged,extra", [
    (2000, 11, 40, True, ("NUM_BUCKETS=5",)), (1500, 12, 300, False, ()), (800, 13, 3, True, ("NUM_BUCKETS=256",)),
    (1200, 14, 25, True, ("NUM_BUCKETS=2",)), (1500, 15, 30, False, ("NUM_BUCKETS=7", "READS_PER_BC=90"))])
def test_against_the_reference_binary(tmp_path, pairs, seed, n_bc, ragged, extra):
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, pairs, seed, n_bc=n_bc, ragged=ragged)
    r = run(REF, fq1, fq2, f"{tmp_path}/ref/reads", "NUM_THREADS=1", *extra)
    assert r.returncode == 0, r.stdout + r.stderr
    o = run(OURS, fq1, fq2, f"{tmp_path}/ours/reads", "NUM_THREADS=4", *extra)
    assert o.returncode == 0, o.stderr
    same_files(f"{tmp_path}/ours/reads", f"{tmp_path}/ref/reads")


def test_one_bucket_works(tmp_path):
    """NUM_BUCKETS=1 makes the reference abort (its bucket loop never closes the only bucket: vec::back() on an empty
    vec, 10X/ParseBarcodedFastqs.cc:331-335); here it is simply one bucket: all barcodes ascending."""
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, 300, 31, n_bc=9)
    assert run(OURS, fq1, fq2, f"{tmp_path}/o/reads", "NUM_BUCKETS=1").returncode == 0
    bci = feudal.read_bci(f"{tmp_path}/o/reads.bci")
    assert bci[-1] == 600 and len(bci) >= 3


def test_refuses_what_the_reference_refuses(tmp_path):
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, 50, 3)
    make_fastq(f"{tmp_path}/b_1.fq.gz", f"{tmp_path}/b_2.fq.gz", 40, 4)
    assert run(OURS, fq1, f"{tmp_path}/b_2.fq.gz", f"{tmp_path}/o/reads").returncode != 0       # files that are not a pair
    open(f"{tmp_path}/plain.fq", "w").write("@r#0_0_0/1\nACGT\n+\nIIII\n")
    r = run(OURS, f"{tmp_path}/plain.fq", fq2, f"{tmp_path}/o/reads")
    assert r.returncode != 0 and "gz format" in r.stderr
    assert run(OURS, fq1, fq2, f"{tmp_path}/o/").returncode != 0                                  # OUT_HEAD ending in '/'


def test_output_feeds_the_df_front_end(tmp_path):
    """runall.sh:125-127: ParseBarcodedFastqs then DF, the ingest half (no GPU needed for EXIT_LOAD)."""
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, 700, 21, n_bc=30)
    assert run(OURS, fq1, fq2, f"{tmp_path}/tmp/reads", "NUM_BUCKETS=6").returncode == 0
    df = subprocess.run([os.path.join(ROOT, "superplus_amd", "DF"), f"ROOT={tmp_path}/tmp", f"LR={tmp_path}/tmp/reads.fastb", "PIPELINE=cs",
                         "ALIGN=False", "NUM_THREADS=4", "MAX_MEM_GB=640", "EXIT_LOAD=True"], capture_output=True, text=True)
    assert df.returncode == 0, df.stdout + df.stderr
    same_files(f"{tmp_path}/tmp/GapToy/1/data/frag_reads_orig", f"{tmp_path}/tmp/reads")


@pytest.mark.parametrize("mem_mb,extra", [(1.3, ("NUM_BUCKETS=8",)), (0.7, ("NUM_BUCKETS=16",)), (1.0, ("NUM_BUCKETS=9", "READS_PER_BC=400")),
                                          (0.3, ("NUM_BUCKETS=64",))])            # (the last: the unbarcoded pairs alone take several pieces)
def test_a_set_larger_than_max_mem_gb_is_done_in_groups_of_buckets(tmp_path, mem_mb, extra):
    """The reads do not fit MAX_MEM_GB: the first pass keeps barcodes and lengths, the inputs are read again once per group of
    consecutive buckets that fits (the unbarcoded pairs, which keep the file's order, in as many pieces as it takes) -- and the
    three files come out as when everything is held at once (and as the reference's binary writes them)."""
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, 3000, 41, n_bc=40, ragged=True)
    whole = run(OURS, fq1, fq2, f"{tmp_path}/whole/reads", "NUM_THREADS=3", *extra)
    assert whole.returncode == 0 and "more passes" not in whole.stderr, whole.stderr
    o = run(OURS, fq1, fq2, f"{tmp_path}/groups/reads", "NUM_THREADS=3", f"MAX_MEM_GB={mem_mb / 1024}", *extra)
    assert o.returncode == 0, o.stderr
    n_passes = int(o.stderr.split("allowed: ")[1].split()[0])
    assert n_passes >= 3, o.stderr
    same_files(f"{tmp_path}/groups/reads", f"{tmp_path}/whole/reads")
    if os.path.exists(REF):
        r = run(REF, fq1, fq2, f"{tmp_path}/ref/reads", "NUM_THREADS=1", *extra)
        assert r.returncode == 0, r.stdout + r.stderr
        same_files(f"{tmp_path}/groups/reads", f"{tmp_path}/ref/reads")


def test_refuses_an_input_that_does_not_fit_max_mem_gb(tmp_path):
    """A SINGLE bucket whose reads do not fit MAX_MEM_GB ends the run with a message saying so (and no half-written files),
    not with the OOM killer."""
    from tests.fastq_synth import make_fastq
    make_fastq(f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz", 3000, 5, n_bc=6)
    r = subprocess.run([OURS, "FASTQS={" + f"{tmp_path}/a_1.fq.gz,{tmp_path}/a_2.fq.gz" + "}", f"OUT_HEAD={tmp_path}/o", "NUM_BUCKETS=2", "MAX_MEM_GB=0.0005", "HOST=True"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "MAX_MEM_GB" in r.stderr and not os.path.exists(f"{tmp_path}/o.fastb")


def test_without_a_device_the_default_path_fails_loudly(tmp_path):
    """The product path is the device's: where there is no GPU the program says so and writes nothing (HOST=True is explicit)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, 50, 3)
    r = run(OURS, fq1, fq2, f"{tmp_path}/o/reads", device=True)
    assert r.returncode != 0 and "HOST=True" in r.stderr and not os.path.exists(f"{tmp_path}/o/reads.fastb")
