#!/usr/bin/env python3
"""bench.py -- k-mers/s of the DF createDict hot path on MI355X.

A "step" is one full pass of the hot path (quality-tail trim, canonical k-mer extraction,
count, MIN_FREQ/MIN_BC solid filter, spectrum, adjacency clean-up) over one synthetic stLFR
read set that is already resident in HBM when the timed region starts.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: synthetic human-scale stLFR, 900 M pairs of 2x100 bp over a
3.1 Gb random genome (1.8 G reads, 8.8e10 k-mer instances, ~3.1e9 solid 48-mers), 0.5 % substitutions,
10 % unbarcoded pairs, K=48, MIN_FREQ=3, MIN_BC=2, MIN_QUAL=7.  It is generated directly in HBM
(torch, seeded) in ~20 s.  On one GPU the library counts it in passes over ranges of its minimizer buckets (the dictionary alone
is 99 GB).  With N GPUs the SAME set is sharded by pair ranges (configs[2]; "scaling": "strong"): every
rank generates and holds 900M/N pairs, records travel to the rank owning their minimizer bucket in one
RCCL all-to-all per pass, and a second small exchange settles cross-rank adjacencies.
--pairs / --genome-mb scale it down for quick runs.

One JSON line on rank 0.  `roofline` is for the dominant kernel (k_count): algorithmic bytes
= 64 B per k-mer instance (one 32-B sector read + one 32-B sector write of the
owning table slot) divided by the kernel's duration from HIP events on the library's stream.
`cpu_baseline` times the reference's own MapReduceEngine/KmerDict (oracle/_ref/refdrv, kind
"reference") or, if that binary is absent, the C restatement (kind "port") on a bounded
sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from superplus_amd import synth  # noqa: E402
from superplus_amd.dfk import Dfk  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
B_INST = 64                    # algorithmic bytes per k-mer instance in the count kernel


def profiled_traffic(n_inst):
    """HBM bytes per k_count launch from the committed rocprofv3 --pmc passes (profiles/rNN_traffic.json,
    produced by tools/profile_gpu.sh + tools/summarize_prof.py on this same command).  Counters cannot be
    read inside this process, so the figure is quoted only when the profiled run had the same instance count."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    line = t.get("bench_line_under_profiler") or {}
    if (line.get("counts_rank0") or {}).get("n_inst") != n_inst:
        return None
    return t["hbm_bytes_per_launch"]


def cpu_baseline(rs, K, sample_reads):
    """Reference components (or the port) on the first `sample_reads` reads, all host cores."""
    from superplus_amd import feudal
    n = min(sample_reads, rs.n_reads) & ~1
    nb, nq = int(rs.base_off[n]), int(rs.pq_off[n])
    sub = dict(packed=rs.packed[:nb].cpu().numpy(), base_off=rs.base_off[: n + 1].cpu().numpy().astype(np.uint64),
               read_len=rs.read_len[:n].cpu().numpy().astype(np.uint32), pq_bytes=rs.pq_bytes[:nq].cpu().numpy(),
               pq_off=rs.pq_off[: n + 1].cpu().numpy().astype(np.uint64), bc=rs.bc[:n].cpu().numpy().astype(np.int32))
    cores = os.cpu_count() or 1
    refdrv = os.path.join(ROOT, "oracle", "_ref", "refdrv")
    if os.path.exists(refdrv):
        try:
            with tempfile.TemporaryDirectory() as d:
                feudal.write_fastb(d + "/s.fastb", sub["packed"], sub["base_off"], sub["read_len"])
                feudal.write_qualp(d + "/s.qualp", sub["pq_bytes"], sub["pq_off"])
                bc = sub["bc"].astype(np.int64)
                nb = int(bc.max()) + 1 if n else 1
                bci = np.concatenate([[0], np.cumsum(np.bincount(bc, minlength=nb))]).astype(np.int64)
                feudal.write_bci(d + "/s.bci", bci)
                os.makedirs(d + "/o")
                threads = min(cores, 32)
                subprocess.run([refdrv, "dict", str(K), d + "/s", d + "/o", "7", "3", "2", "1", str(threads)],
                               check=True, stdout=subprocess.DEVNULL, timeout=600)
                t = dict(line.split() for line in open(d + "/o/times.txt"))
                secs = sum(float(t[k]) for k in ("goodlens_s", "mr1_s", "mr2_s", "dict_s", "adj_s"))
                return {"value": float(t["instances"]) / secs, "unit": "k-mers/s", "cores": threads, "kind": "reference",
                        "sample": f"first {n} reads of the workload ({t['instances']} k-mer instances); "
                                  "createDict-equivalent = tail scan + 2 MapReduceEngine runs + Dict build + "
                                  f"recomputeAdjacencies, {secs:.2f} s"}
        except Exception as e:  # fall through to the port
            print(f"[bench] refdrv baseline failed ({e}); using the C port", file=sys.stderr)
    from oracle import pyoracle
    t0 = time.time()
    r = pyoracle.run(sub["packed"], sub["base_off"], sub["read_len"], sub["pq_bytes"], sub["pq_off"], sub["bc"], K=K,
                     threads=cores)
    secs = time.time() - t0
    return {"value": r["n_inst"] / secs, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": f"first {n} reads of the workload ({r['n_inst']} k-mer instances), {secs:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mb", type=float, default=3100.0, help="genome size, Mb")
    ap.add_argument("--pairs", type=int, default=900_000_000, help="read pairs in the whole set")
    ap.add_argument("--coverage", type=float, default=0.0, help="if > 0: pairs = coverage * genome / 200")
    ap.add_argument("--passes", type=int, default=0, help="number of bucket-range passes (0 = sized from free HBM)")
    ap.add_argument("--K", type=int, default=48)
    ap.add_argument("--minimizer", type=int, default=0)
    ap.add_argument("--inst-per-item", type=int, default=0)
    ap.add_argument("--cpu-sample-reads", type=int, default=6000000,
                    help="reads of the workload the CPU baseline runs on (about 10 s of reference code on 32 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--family-copies", type=int, default=0,
                    help="plant this many diverged copies of one 300-bp element in the genome (hot minimizer buckets)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (torch.distributed + DistDfk) even with one rank: a check of that path on one GPU")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; gloo (with --one-device) rehearses the multi-rank launch on one GPU")
    ap.add_argument("--hbm-budget-gb", type=float, default=0.0, help="HBM the library may use (0 = 90 %% of what is free)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="rehearsal on one GPU: run rank 0 of this many ranks against replicas of itself (not a benchmark result)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libdfk has no CPU path")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        if "RANK" not in os.environ:                 # --force-dist without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            dist.init_process_group(args.backend, rank=0, world_size=1, device_id=dev)
        elif args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where the bench's own small collectives live

        def allreduce(value, dtype, op):
            t = torch.tensor([value], dtype=dtype, device=cdev)
            dist.all_reduce(t, op=op)
            return t.item()

    G = int(args.genome_mb * 1e6)
    total_pairs = int(args.coverage * G / 200.0) if args.coverage > 0 else args.pairs
    lo, hi = rank * total_pairs // world, (rank + 1) * total_pairs // world          # this rank's pair range
    if args.emulate_world > 1:
        lo, hi = 0, total_pairs // args.emulate_world
    genome = synth.make_genome(G, 20261004, device=dev, family_copies=args.family_copies)   # same genome on every rank
    rs = synth.make_reads(genome, hi - lo, 20261004 + 17 * (rank + 1))
    del genome
    if multi:          # barcode ids must not collide between ranks
        stride = int(allreduce(int(rs.bc.max().item()) + 1, torch.int64, dist.ReduceOp.MAX))
        rs.bc = torch.where(rs.bc > 0, rs.bc + rank * stride, rs.bc)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()          # the library sizes its HBM budget from what is free when the context is created

    if args.emulate_world > 1:
        from superplus_amd.dist import DistDfk, ReplicaComm
        d = DistDfk(comm=ReplicaComm(args.emulate_world), K=args.K, device=local, minimizer_len=args.minimizer,
                    inst_per_item=args.inst_per_item, passes=args.passes,
                    hbm_budget_bytes=int(args.hbm_budget_gb * 1e9))
        def step():
            d.count_device(rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc, read_id0=0)
            return d.stats()
    elif not multi:
        d = Dfk(K=args.K, device=local, minimizer_len=args.minimizer, inst_per_item=args.inst_per_item, passes=args.passes,
                    hbm_budget_bytes=int(args.hbm_budget_gb * 1e9))
        def step():
            d.count_device(rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)
            return d.stats()
    else:
        from superplus_amd.dist import DistDfk
        d = DistDfk(K=args.K, device=local, minimizer_len=args.minimizer, inst_per_item=args.inst_per_item, passes=args.passes,
                    hbm_budget_bytes=int(args.hbm_budget_gb * 1e9))
        def step():
            d.count_device(rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc, read_id0=2 * lo)
            return d.stats()

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    st = None
    for _ in range(args.steps):
        st = step()
    barrier()
    elapsed = time.perf_counter() - t0
    n_inst = st["n_inst"]
    if multi:
        elapsed = float(allreduce(elapsed, torch.float64, dist.ReduceOp.MAX))
        n_inst = int(allreduce(n_inst, torch.int64, dist.ReduceOp.SUM))

    # multi-rank runs: where the last step's time went on the slowest and on the fastest rank (host clock per
    # phase of DistDfk.count_device), and the bytes a rank sent over the links
    dist_timing = None
    if hasattr(d, "timing"):
        tm = d.timing
        if multi:
            keys = [k for k in sorted(tm) if k not in ("n_passes",)]
            dist_timing = {"max_over_ranks": {k: round(float(allreduce(float(tm[k]), torch.float64, dist.ReduceOp.MAX)), 3) for k in keys},
                           "min_over_ranks": {k: round(float(allreduce(float(tm[k]), torch.float64, dist.ReduceOp.MIN)), 3) for k in keys},
                           "n_passes": tm["n_passes"]}
        else:
            dist_timing = {"rank0": {k: round(float(v), 3) for k, v in tm.items()}}
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = n_inst * args.steps / elapsed
        k_ms = st["ms_count"]
        k_inst = n_inst // world                                   # instances this rank's k_count launches handled
        achieved = B_INST * k_inst / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        full = G == 3_100_000_000 and total_pairs == 900_000_000
        out = {
            "metric": "k-mers/s (DF createDict stage: trim + canonical k-mer count + solid filter + spectrum + adjacency)",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[%d]: " % (1 if world == 1 else 2) if full else "scaled-down run: ") +
                                   f"synthetic stLFR, {total_pairs} pairs 2x100 bp over a {args.genome_mb:g} Mb random genome "
                                   f"({200.0 * total_pairs / G:.1f}x), 0.5% subst., 10% unbarcoded, K={args.K}, MIN_FREQ=3, MIN_BC=2, MIN_QUAL=7" +
                                   (f", {args.family_copies} copies of a 300-bp repeat family" if args.family_copies else ""),
                       "reads_total": 2 * total_pairs, "kmer_instances_total": n_inst, "K": args.K,
                       "parallelism": "single GPU, bucket-range passes" if world == 1 else
                                      f"{world} ranks: read shards, all-to-all of super-k-mer records by minimizer bucket"},
            "df_stage_wall_s": elapsed / args.steps,
            **({"rehearsal": f"rank 0 of {args.emulate_world} against replicas of itself; per-rank time without the transfers"}
               if args.emulate_world > 1 else {}),
            "stage_ms_rank0": {k: round(st[k], 3) for k in ("ms_trim", "ms_part_count", "ms_part_scatter", "ms_count",
                                                           "ms_fallback", "ms_adjacency", "ms_total")},
            "counts_rank0": {k: st[k] for k in ("n_reads", "n_inst", "n_records", "n_buckets", "n_items", "n_overflow_items",
                                                "n_distinct", "n_solid", "adj_probes", "hbm_bytes_peak", "n_passes")},
            "roofline": {"bound": "hbm", "kernel": "k_count", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": profiled_traffic(st["n_inst"]),
                         "algorithmic_bytes_per_launch": B_INST * k_inst // max(1, st["n_passes"]),
                         "kernel_ms_all_launches": k_ms, "launches_per_step": st["n_passes"]},
        }
        if dist_timing is not None:
            out["dist_timing_ms"] = dist_timing
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rs, args.K, args.cpu_sample_reads)
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
