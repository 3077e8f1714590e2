"""tools/check_shard_scale.py -- the sharded pipeline against the single-GPU pipeline at sizes the test suite does
not reach (the single-GPU pipeline is the one pinned to the oracle by tests/test_gpu_parity.py).

    python tools/check_shard_scale.py --genome-mb 3100 --pairs 225000000 [--rccl] [--steps 2]

One rank owns every bucket: in-process exchanges by default, --rccl for torch.distributed at world size 1."""
import argparse, os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from superplus_amd import synth
from superplus_amd.dfk import Dfk
from superplus_amd.dist import DistDfk, run_inprocess


def summary(d):
    st = d.stats()
    return {"n_inst": st["n_inst"], "n_distinct": st["n_distinct"], "n_solid": st["n_solid"],
            "spectrum": np.asarray(d.spectrum())[:12].tolist(), "passes": st["n_passes"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=3100.0)
    ap.add_argument("--pairs", type=int, default=225_000_000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--rccl", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    genome = synth.make_genome(int(args.genome_mb * 1e6), 20261004, device=dev)
    rs = synth.make_reads(genome, args.pairs, 20261021)
    del genome
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    shard = (rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc, 0)

    d = Dfk(K=48, device=0)
    d.count_device(*shard[:6])
    want = summary(d)
    print("single GPU:", want, flush=True)
    del d
    torch.cuda.empty_cache()

    if args.rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    s = DistDfk(K=48, device=0)
    bad = 0
    for step in range(args.steps):
        if args.rccl:
            s.count_device(*shard[:6], read_id0=0)
        else:
            run_inprocess([s], [shard], pipelined=True)
        got = summary(s)
        same = all(got[k] == want[k] for k in ("n_inst", "n_distinct", "n_solid", "spectrum"))
        print(f"sharded, step {step}:", got, "SAME" if same else "DIFFERENT", flush=True)
        bad += not same
    if args.rccl:
        dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
