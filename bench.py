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
SEED = 20261004


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


def head_of(rs, n):
    """The first n reads of a device-resident read set, as tensors (views) the library takes."""
    nb, nq = int(rs.base_off[n]), int(rs.pq_off[n])
    return (rs.packed[:nb], rs.base_off[: n + 1], rs.read_len[:n], rs.pq_bytes[:nq], rs.pq_off[: n + 1], rs.bc[:n])


def cpu_baseline_and_parity(rs, K, device, what):
    """Reference components (or the port) on the read set `rs` (the bounded sample of the workload), all host cores --
    and, since the answer is there anyway, the HIP path on the same reads checked against it: spectrum equal,
    dictionary digest equal (dfk_solid_digest on the device against the numpy form over the CPU's entries)."""
    from superplus_amd import feudal
    from superplus_amd.dfk import ENTRY_DTYPE, digest_of
    n = rs.n_reads
    packed, base_off, read_len, pq_bytes, pq_off, bc = head_of(rs, n)
    sub = dict(packed=packed.cpu().numpy(), base_off=base_off.cpu().numpy().astype(np.uint64),
               read_len=read_len.cpu().numpy().astype(np.uint32), pq_bytes=pq_bytes.cpu().numpy(),
               pq_off=pq_off.cpu().numpy().astype(np.uint64), bc=bc.cpu().numpy().astype(np.int32))
    cores = os.cpu_count() or 1
    try:                                             # (a cgroup's CPU quota is what the threads really get: 16 of the box's 256 here)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    refdrv = os.path.join(ROOT, "oracle", "_ref", "refdrv")
    base, cpu_solid, cpu_hist, ref_files = None, None, None, None
    if os.path.exists(refdrv):
        try:
            with tempfile.TemporaryDirectory() as d:
                feudal.write_fastb(d + "/s.fastb", sub["packed"], sub["base_off"], sub["read_len"])
                feudal.write_qualp(d + "/s.qualp", sub["pq_bytes"], sub["pq_off"])
                b64 = sub["bc"].astype(np.int64)
                nbc = int(b64.max()) + 1 if n else 1
                bci = np.concatenate([[0], np.cumsum(np.bincount(b64, minlength=nbc))]).astype(np.int64)
                feudal.write_bci(d + "/s.bci", bci)
                os.makedirs(d + "/o")
                threads = min(cores, 32)
                # "graph" = dict, then the rest of buildReadQGraph48 and the two steps behind it through the reference's own
                # containers and writers (oracle/ref_graph.cc): a.<K>/ with the graph, a.paths, a.paths.inv, a.countsb, a.dup --
                # what rows f-1, f-2 and f-4 are compared with below.  Its time is not part of the baseline (createDict only).
                out = subprocess.run([refdrv, "graph" if K == 48 else "dict", str(K), d + "/s", d + "/o", "7", "3", "2", "1", str(threads)],
                                     check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900).stdout
                adir = os.path.join(d, "o", f"a.{K}")
                if os.path.isdir(adir):
                    ref_files = {f: open(os.path.join(adir, f), "rb").read() for f in sorted(os.listdir(adir)) if f.startswith("a.") and os.path.isfile(os.path.join(adir, f))}   # (a.*: the stage's files; refdrv keeps its translation tables beside them)
                # a MapReduceEngine run that overflowed a buffer has dropped barcodes (MapReduceEngine.h:533-538 prints it):
                # not a reference answer (caveat 2)
                if "buffer overflow" in out:
                    raise RuntimeError("the reference's MapReduceEngine reported buffer overflows on this sample")
                t = dict(line.split() for line in open(d + "/o/times.txt"))
                phases = {k: float(t[k]) for k in ("goodlens_s", "mr1_s", "mr2_s", "dict_s", "adj_s")}
                secs = sum(phases.values())
                cpu_solid = np.fromfile(d + "/o/solid.bin", ENTRY_DTYPE)
                cpu_hist = np.loadtxt(d + "/o/spectrum.txt", dtype=np.int64, ndmin=1)
                base = {"value": float(t["instances"]) / secs, "unit": "k-mers/s", "cores": threads, "kind": "reference",
                        "sample": f"{what} ({t['instances']} k-mer instances); "
                                  "createDict-equivalent = tail scan + 2 MapReduceEngine runs + Dict build + "
                                  f"recomputeAdjacencies, {secs:.2f} s",
                        # refdrv's own clock per phase (times.txt): the tail scan is single-threaded in the driver (the reference
                        # runs it under parallelForBatch); mr1 counts the solid k-mers, mr2 collects them
                        "phases_s": {"goodlens_single_thread": phases["goodlens_s"], "mapreduce_1": phases["mr1_s"], "mapreduce_2": phases["mr2_s"],
                                     "dict_insert_single_thread": phases["dict_s"], "recompute_adjacencies": phases["adj_s"]},
                        "value_without_single_threaded_phases": float(t["instances"]) / max(1e-9, phases["mr1_s"] + phases["mr2_s"] + phases["adj_s"])}
        except Exception as e:  # fall through to the port
            print(f"[bench] refdrv baseline failed ({e}); using the C port", file=sys.stderr)
    if base is None:
        from oracle import pyoracle
        t0 = time.time()
        r = pyoracle.run(sub["packed"], sub["base_off"], sub["read_len"], sub["pq_bytes"], sub["pq_off"], sub["bc"], K=K,
                         threads=cores)
        secs = time.time() - t0
        cpu_solid, cpu_hist = r["solid"], r["hist"]
        base = {"value": r["n_inst"] / secs, "unit": "k-mers/s", "cores": cores, "kind": "port",
                "sample": f"{what} ({r['n_inst']} k-mer instances), {secs:.2f} s"}
    # the HIP path on the same reads
    d = Dfk(K=K, device=device)
    d.count_device(packed, base_off, read_len, pq_bytes, pq_off, bc)
    hist = np.asarray(d.spectrum())
    ok_hist = len(hist) == len(cpu_hist) and bool(np.array_equal(hist, cpu_hist))
    ok_dict = d.solid_count() == len(cpu_solid) and d.digest() == digest_of(cpu_solid)
    parity = {"status": "ok" if (ok_hist and ok_dict) else "MISMATCH", "against": base["kind"], "reads": n,
              "solid": int(d.solid_count()), "spectrum_equal": ok_hist, "dictionary_digest_equal": ok_dict}
    if ref_files:
        # rows f-1, f-2, f-4 on the same reads: every file of the reference's a.<K>/ byte for byte
        with tempfile.TemporaryDirectory() as t:
            g = d.graph_build(); d.graph_write(t)
            p = d.paths_build_device(packed, base_off, read_len, pq_bytes, pq_off)
            d.paths_write(t + "/a.paths"); d.paths_index_write(t); n_dup = d.dups_write(t + "/a.dup")
            same = {f: os.path.exists(os.path.join(t, f)) and open(os.path.join(t, f), "rb").read() == b for f, b in ref_files.items()}
        parity["graph_paths_files"] = {"equal": sorted(f for f, ok in same.items() if ok), "differ": sorted(f for f, ok in same.items() if not ok),
                                       "hbv_edges": g["n_edges"], "reads_placed": p["n_placed"], "dup_pairs": int(n_dup)}
        if not all(same.values()): parity["status"] = "MISMATCH"
    d.close()
    return base, parity


def bind_to_gpu_node(local):
    """This process's threads (and those of the stage it starts) on the CPUs of the GPU's NUMA node, as a launcher's
    `numactl --cpunodebind` would: what it writes into the page cache -- the DF leg's input files -- then lies on that node's
    memory, and the stage's transfer lanes (which bind themselves the same way, df_main.cc) read it at the local rate rather
    than over the socket link.  BENCH_NO_BIND=1 leaves the scheduler alone.  -> what was done, for the JSON."""
    if os.environ.get("BENCH_NO_BIND"): return {"bound": False, "why": "BENCH_NO_BIND"}
    try:
        bus = torch.cuda.get_device_properties(local).pci_bus_id if hasattr(torch.cuda.get_device_properties(local), "pci_bus_id") else None
        dom = getattr(torch.cuda.get_device_properties(local), "pci_domain_id", 0)
        dev_id = getattr(torch.cuda.get_device_properties(local), "pci_device_id", 0)
        if bus is None: return {"bound": False, "why": "no PCI bus id"}
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev_id:02x}.0/numa_node"
        node = int(open(path).read())
        if node < 0: return {"bound": False, "why": "no NUMA node recorded for the device"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus)
        return {"bound": True, "numa_node": node, "cpus": len(cpus)}
    except Exception as e:                      # (an unusual sysfs: measure unbound, and say so)
        return {"bound": False, "why": str(e)[:120]}


def timed_steps(d, shard, steps, warmup=1):
    """-> (seconds per step, stats of the last step) for a single-GPU context."""
    for _ in range(warmup):
        d.count_device(*shard)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        d.count_device(*shard)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, d.stats()


def leg_summary(secs, st):
    return {"kmers_per_s": st["n_inst"] / secs, "step_s": round(secs, 4), "n_passes": st["n_passes"], "n_inst": st["n_inst"],
            "n_solid": st["n_solid"], "n_items": st["n_items"], "n_overflow_items": st["n_overflow_items"],
            "ms_count": round(st["ms_count"], 2), "ms_part_count": round(st["ms_part_count"], 2),
            "ms_fallback": round(st["ms_fallback"], 2)}


def write_read_files(rs, head, workers=8):
    """A device-resident read set as head.{fastb,qualp,bci} (feudal files, written in slabs by a few threads)."""
    from concurrent.futures import ThreadPoolExecutor
    from superplus_amd import feudal
    n = rs.n_reads
    var_b, var_q = int(rs.base_off[n]), int(rs.pq_off[n])

    def plan(var_len, has_fixed):
        var_tab = 24 + var_len
        fixed_off = var_tab + 8 * (n + 1)
        return var_tab, fixed_off, fixed_off + (4 * n if has_fixed else 0)

    jobs = []          # (fd, file offset, tensor-producing thunk)
    fds = []
    SLAB = 64 << 20

    def add(fd, off, t, bias=0):
        step = max(1, SLAB // t.element_size())
        for a in range(0, t.numel(), step):
            jobs.append((fd, off + a * t.element_size(), t, a, min(t.numel(), a + step), bias))

    for ext, var, offs, fixed, hdr in ((".fastb", rs.packed[:var_b], rs.base_off, rs.read_len, (4, 16, 1)),
                                       (".qualp", rs.pq_bytes[:var_q], rs.pq_off, None, (0, 8, 1))):
        var_tab, fixed_off, size = plan(var.numel(), fixed is not None)
        fd = os.open(head + ext, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
        os.ftruncate(fd, size)
        os.pwrite(fd, feudal.header(n, hdr[0], hdr[1], hdr[2], var_tab, fixed_off), 0)
        add(fd, 24, var)
        add(fd, var_tab, offs, bias=24)                     # the table holds absolute file offsets
        if fixed is not None:
            add(fd, fixed_off, fixed)
        fds.append(fd)

    def run(job):
        fd, off, t, a, b, bias = job
        x = t[a:b]
        if bias:
            x = x + bias
        buf = x.cpu().numpy().tobytes() if x.dtype != torch.uint8 else x.cpu().numpy()
        os.pwrite(fd, buf, off)

    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(run, jobs))
    for fd in fds:
        os.close(fd)
    feudal.write_bci(head + ".bci", rs.bci)


def host_memory_limits():
    """What the box lets this command hold in RAM (files on /dev/shm count): the cgroup's limit and use, and the room on
    /dev/shm.  gpurun ends a command at about 90 % of the cgroup limit (its 270 GiB of a 300 GiB cgroup), so that is the cap."""
    out = {}
    for name, path in (("cgroup_memory_max", "/sys/fs/cgroup/memory.max"), ("cgroup_memory_current", "/sys/fs/cgroup/memory.current")):
        try:
            v = open(path).read().strip()
            out[name] = None if v == "max" else int(v)
        except OSError:
            out[name] = None
    try:
        st = os.statvfs("/dev/shm"); out["dev_shm_free"] = st.f_bavail * st.f_frsize
    except OSError:
        out["dev_shm_free"] = None
    try:
        out["mem_available"] = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
    except (OSError, IndexError):
        out["mem_available"] = None
    cap = min(x for x in (0.9 * out["cgroup_memory_max"] if out["cgroup_memory_max"] else None, out["mem_available"], 1e18) if x is not None)
    out["usable"] = int(cap - (out["cgroup_memory_current"] or 0))
    return out


def pbf_leg(args, local):
    """ParseBarcodedFastqs on `--pbf-pairs` synthetic stLFR pairs (2x100 bp over a 4.6 Mb genome: BASELINE configs[0]'s shape):
    wall time of the device path and of the host path, the device's own times, and that the three files are identical."""
    import re, shutil
    root = tempfile.mkdtemp(prefix="dfk_pbf_", dir=args.df_dir if os.path.isdir(args.df_dir) else None)
    try:
        fq = [root + "/r_1.fq.gz", root + "/r_2.fq.gz"]
        t0 = time.perf_counter()
        synth.write_fastq_pair(fq[0], fq[1], args.pbf_pairs, SEED + 3, threads=args.df_threads)
        t_gen = time.perf_counter() - t0
        exe = os.path.join(ROOT, "superplus_amd", "ParseBarcodedFastqs")
        res = {}
        for name, extra_args in (("device", [f"DEVICE={local}"]), ("host", ["HOST=True"])):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "FASTQS={" + fq[0] + "," + fq[1] + "}", f"OUT_HEAD={root}/{name}/reads", f"NUM_THREADS={args.df_threads}", "NUM_BUCKETS=10", *extra_args],
                               capture_output=True, text=True, timeout=1200)
            res[name] = {"wall_s": round(time.perf_counter() - t0, 3)}
            if r.returncode != 0:
                return {"error": f"{name} path exited {r.returncode}: {r.stderr[-300:]}"}
            m = re.search(r"device: pair order ([\d.]+) ms, packing \+ PQVec ([\d.]+) ms, with transfers ([\d.]+) ms", r.stderr)
            if m: res[name]["device_ms"] = {"pair_order": float(m.group(1)), "pack_and_pqvec": float(m.group(2)), "with_transfers": float(m.group(3))}
        same = all(open(f"{root}/device/reads.{e}", "rb").read() == open(f"{root}/host/reads.{e}", "rb").read() for e in ("fastb", "qualp", "bci"))
        return {"pairs": args.pbf_pairs, "workload": f"{args.pbf_pairs} pairs 2x100 bp over a 4.6 Mb genome, fastq.gz, {max(1, args.pbf_pairs // 20)} barcodes, 10 % unbarcoded (BASELINE configs[0]'s shape)",
                "device_path": res["device"], "host_path": dict(res["host"], threads=args.df_threads), "files_identical": same,
                "note": "both walls include inflating and line-splitting the two .gz files (one host thread each) and writing the three files; device_ms is what the device path spends in dfk_pbf_run",
                "fastq_written_in_s": round(t_gen, 1)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cgroup_cpu_stat():
    """/sys/fs/cgroup/cpu.stat of this process's cgroup as {name: int} ({} where there is none)."""
    try:
        rel = open("/proc/self/cgroup").read().strip().split("::", 1)[1].strip()
        for d in ("/sys/fs/cgroup" + rel, "/sys/fs/cgroup"):
            if os.path.exists(d + "/cpu.stat"):
                return {a: int(b) for a, b in (l.split() for l in open(d + "/cpu.stat"))}
    except Exception:
        pass
    return {}


def df_stage_wall(args, dev, local):
    """BASELINE.json's other half: the DF stage's wall-clock, measured as SURVEY 8(d) defines it -- process start of
    `DF ROOT=... LR=...` (the unchanged runall.sh:127 command line) to its exit, with every output written.  With the
    graph (the default): everything 10X/DF.cc:300-561 leaves behind for the next stage -- frag_reads_orig.*, the side files,
    the spectrum JSON and a.<K>/ with the graph, the read paths, the inverted paths index and the duplicate marks (no
    kmers.kvec: the reference removes its own, BuildReadQGraph48.cc:303).  Files live on /dev/shm (RAM-backed); the leg runs
    BASELINE configs[1] at full size when the command's memory allowance holds inputs + outputs, scaled down otherwise."""
    import shutil
    G = int(args.genome_mb * 1e6)
    total_pairs = int(args.coverage * G / 200.0) if args.coverage > 0 else args.pairs
    mem = host_memory_limits()
    # bytes in RAM per read: input files 50.4, their frag_reads_orig copies 50.4 (none when hard-linked), lens 2, and with the
    # graph a.paths 20, a.paths.inv 8, a.dup 0.5; kmers.kvec (GRAPH=False) 32 per solid k-mer = 55; DF's own vectors 6
    per_read_link = 50.4 + 2 + 6 + (28.5 if args.df_graph else 55.0)
    per_read_copy = per_read_link + 50.4
    want = min(total_pairs, args.df_pairs) if args.df_pairs else total_pairs
    link = False
    if 2 * want * per_read_copy * 1.25 > mem["usable"]:
        link = True                                                      # frag_reads_orig.{fastb,qualp} as hard links to the inputs (LINK_READS=True)
    pairs = want
    if 2 * pairs * per_read_link * 1.25 > mem["usable"]:
        pairs = max(1000, int(mem["usable"] / 1.25 / per_read_link / 2))
    Gd = max(1000, int(G * pairs / total_pairs))
    root = tempfile.mkdtemp(prefix="dfk_df_", dir=args.df_dir if os.path.isdir(args.df_dir) else None)
    # a second, smaller input for the stage as the reference's DF leaves it by default -- frag_reads_orig.{fastb,qualp} COPIED
    # (10X/DfTools.cc:164-167), not linked: as many pairs as fit the allowance with the copies (half of configs[1] on this box)
    pairs_c = min(pairs // 2, max(1000, int(mem["usable"] / 1.25 / per_read_copy / 2)))
    try:
        genome = synth.make_genome(Gd, SEED, device=dev)
        rs = synth.make_reads(genome, pairs, SEED + 17)
        del genome
        t0 = time.perf_counter()
        write_read_files(rs, root + "/reads")
        t_files = time.perf_counter() - t0
        in_bytes = sum(os.path.getsize(root + "/reads" + e) for e in (".fastb", ".qualp", ".bci"))
        have_c = link and args.df_copies and not (args.df_gpus > 1 or args.df_transport)
        if have_c:                                               # the first pairs_c pairs of the same reads: the same genome at a lower coverage
            write_read_files(rs.head(pairs_c), root + "/half")
        del rs
        held = torch.cuda.memory_reserved(local)
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        # The device memory this process has just given back (the read generator's, ~230 GB) is wiped by the driver before
        # anybody gets it again, at about 33 GB/s (tools/vram_alloc_cost.hip: a 180-GiB hipMalloc right behind the release of
        # one takes 5.4 s): started at once, the stage's own allocations wait for the harness's leftovers.  A pipeline's DF
        # starts on a quiet device.  BOTH clocks are reported: the run started at once (cold) and the run started after the wait.
        quiesce = float(os.environ.get("BENCH_QUIESCE_S", min(15.0, held / 25e9)))
        # transfer lanes per copy: four (each keeps a DMA in flight and spins on it; three copies run side by side in the stage's last
        # phase, and the box gives the command 16 CPUs -- measured at full size: 16 lanes 17.1 s, 8: 16.7, 6: 17.0, 4: 15.6-16.2)
        # ... the upload runs alone and is bound by its lanes' copies out of the page cache: eight (4 lanes 2.05-2.08 s, 6: 1.90-1.96, 8: 1.85-1.92)
        env = dict(os.environ, DFK_HOST_THREADS=os.environ.get("DFK_HOST_THREADS", str(min(args.df_threads, 4))),
                   DFK_UPLOAD_THREADS=os.environ.get("DFK_UPLOAD_THREADS", str(min(args.df_threads, 8))))

        def run_stage(head, n_pairs, linked, wait_s):
            """one run of the child process on the files `head`.*; the work directory is removed afterwards"""
            cmd = [os.path.join(ROOT, "superplus_amd", "DF"), f"ROOT={root}", f"LR={head}.fastb", "PIPELINE=cs", "ALIGN=False",
                   f"NUM_THREADS={args.df_threads}", "MAX_MEM_GB=640", f"DEVICE={local}", f"K={args.K}",
                   "GRAPH=True" if args.df_graph else "GRAPH=False"]   # (GRAPH=True: rows f-1, f-2, f-4 -- edges + HBV + read paths + paths index + duplicate marks -> a.<K>/)
            if linked: cmd.append("LINK_READS=True")
            if args.df_gpus > 1 or args.df_transport:
                # the C++ multi-GPU host (df_shard.h): DF forks one rank per GPU and moves the records over RCCL itself;
                # `loopback` runs the ranks as threads on ONE GPU (the rehearsal a one-GPU box allows)
                cmd.append(f"NUM_GPUS={max(1, args.df_gpus)}")
                if args.df_transport == "loopback": env["DF_TRANSPORT"] = "loopback"
                elif args.df_gpus <= 1: env["DF_FORCE_SHARDED"] = "1"
            if os.environ.get("DF_TASKSET"): cmd = ["taskset", "-c", os.environ["DF_TASKSET"]] + cmd      # (an experiment's switch: bind the stage's threads)
            if env.get("DF_WRAP"): cmd = env["DF_WRAP"].split() + cmd          # (a DF_VARIANTS run under a profiler: "rocprofv3 --kernel-trace --stats -d dir --")
            time.sleep(wait_s)
            cpu0 = cgroup_cpu_stat()
            t0 = time.perf_counter(); e0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
            wall = time.perf_counter() - t0; e1 = time.time()
            cpu1 = cgroup_cpu_stat()
            if os.environ.get("DFK_TRACE") and os.path.isdir(os.path.join(ROOT, "gpurun_out")):      # the child's trace, for whoever asked for it
                with open(os.path.join(ROOT, "gpurun_out", "df_child_trace.txt"), "a") as f: f.write(r.stderr)
            w = root + "/GapToy/1"
            if r.returncode != 0:
                shutil.rmtree(root + "/GapToy", ignore_errors=True)
                return {"error": f"DF exited {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
            timing, digests = {}, None
            for line in r.stdout.splitlines():
                if line.startswith("DF_MAIN_EPOCH "): timing["spawn_to_main_s"] = round(float(line.split()[1]) - e0, 3)
                if line.startswith("DF_EXIT_EPOCH "): timing["exit_to_reaped_s"] = round(e1 - float(line.split()[1]), 3)
                if line.startswith("DF_DIGESTS "): digests = json.loads(line[len("DF_DIGESTS "):])
                if line.startswith("DF_TIMING "):
                    o = json.loads(line[len("DF_TIMING "):])
                    if "rank0" in o:                     # the C++ sharded host: rank 0's phases, beside the parent's line
                        timing["shard"] = dict(o["rank0"], ranks=o.get("ranks"))
                        for k in ("kmer_instances", "solid"): timing.setdefault(k, o.get(k))
                    else: timing.update(o)
            # what the stage WROTE: hard links to the inputs (LINK_READS) are not output
            out_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(w) for f in fs
                            if not (linked and os.stat(os.path.join(dp, f)).st_nlink > 1))
            shutil.rmtree(root + "/GapToy", ignore_errors=True)
            # the CPU time the run took in all its threads, and how long the cgroup's CPU quota held them back (cpu.stat deltas)
            cpu = {k[:-5] + "_s": round((cpu1[k] - cpu0[k]) / 1e6, 3) for k in ("usage_usec", "user_usec", "system_usec", "throttled_usec") if k in cpu0 and k in cpu1}
            return {"wall_s": round(wall, 3), "waited_before_s": round(wait_s, 1), "pairs": n_pairs, "timing": timing, "digests": digests, "output_bytes": out_bytes, "cpu": cpu,
                    "frag_reads_orig": "hard links to the inputs (LINK_READS=True: byte-identical files, no second copy in RAM)" if linked else "copies of the inputs (the reference's default)"}

        cold = run_stage(root + "/reads", pairs, link, 0.0)                              # started at once, behind the harness's own release
        if "error" in cold: return cold
        # (the child has just released its own ~270 GB: the same wait again before the run that is the headline)
        main = run_stage(root + "/reads", pairs, link, quiesce)
        if "error" in main: return main
        # an experiment's switch: more runs on the same files, each with its own environment ("A=1,B=2;C=3"), to gpurun_out/df_variants.txt
        for spec in filter(None, os.environ.get("DF_VARIANTS", "").split(";")):
            saved = dict(env)
            env.update(kv.split("=", 1) for kv in spec.split(",") if "=" in kv)
            v = run_stage(root + "/reads", pairs, link, quiesce)
            env.clear(); env.update(saved)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                with open(os.path.join(ROOT, "gpurun_out", "df_variants.txt"), "a") as f: f.write(json.dumps({"env": spec, "wall_s": v.get("wall_s"), "cpu": v.get("cpu"), "timing": v.get("timing"), "error": v.get("error")}) + "\n")
        for e in (".fastb", ".qualp", ".bci"): os.unlink(root + "/reads" + e)
        copies = run_stage(root + "/half", pairs_c, False, quiesce) if have_c else None
        wall, timing = main["wall_s"], main["timing"]
        full = pairs == 900_000_000 and Gd == 3_100_000_000
        return {"df_stage_wall_s": wall,
                "df_stage_wall_cold_s": cold["wall_s"],
                "df_stage_wall_copies_s": (wall if not link else None) if not copies else copies.get("wall_s"),
                "workload": ("BASELINE configs[1]" if full else "BASELINE configs[1] scaled to fit the host memory cap") +
                            f": {pairs} pairs 2x100 bp over a {Gd / 1e6:g} Mb random genome ({200.0 * pairs / Gd:.1f}x), K={args.K}",
                "definition": "wall time of the child process `DF ROOT= LR= PIPELINE=cs ALIGN=False NUM_THREADS= MAX_MEM_GB=640` "
                              "(runall.sh:127) from start to exit: map inputs, frag_reads_orig.* (" + main["frag_reads_orig"] + "), lens/qhist/dti, upload, "
                              "createDict on the GPU, spectrum JSON" + (", then what buildReadQGraph48's second half, writePathsIndex and MarkDups leave in a.<K>/ "
                              "(10X/DF.cc:541-561: graph, read paths, paths index, duplicate marks)" if args.df_graph else ", kmers.kvec") +
                              " -- ingest + StageBuildGraph, not the other seven DF stages.  df_stage_wall_s: started on a quiet device "
                              "(device_quiesce_s after the previous holder of the HBM let go); df_stage_wall_cold_s: started at once; "
                              "df_stage_wall_copies_s: frag_reads_orig.{fastb,qualp} written as copies (the reference's default), on `copies.pairs` pairs",
                "frag_reads_orig": main["frag_reads_orig"],
                "device_quiesce_s": round(quiesce, 1),   # waited before the stage started: the driver wiping what the previous process had released
                "host_memory": dict(mem, estimated_need=int(2 * pairs * (per_read_link if link else per_read_copy))),
                "kmers_per_s_whole_stage": (timing.get("kmer_instances", 0) / wall) if wall > 0 else None,
                "cpu_time": main.get("cpu"),             # CPU seconds over all threads / seconds the cgroup's CPU quota held them back
                "breakdown_s": {k: timing.get(k) for k in ("spawn_to_main_s", "open_validate_s", "ingest_outputs_s", "upload_s", "count_s",
                                                           "spectrum_kvec_write_s", "qual_hist_s", "destroy_s", "background_join_s", "total_s", "exit_to_reaped_s")},
                "graph": ({"graph_s": timing.get("graph_s"), "device_s": timing.get("graph_device_s"), "host_s": timing.get("graph_host_s"),
                           "write_s": timing.get("graph_write_s"), "edges": timing.get("graph_edges"), "vertices": timing.get("graph_vertices")}
                          if args.df_graph else None),
                "paths": ({"paths_s": timing.get("paths_s"), "device_s": timing.get("paths_device_s"), "write_s": timing.get("paths_write_s"),
                           "reads_placed": timing.get("reads_placed"), "path_edges": timing.get("path_edges"),
                           "paths_index_s": timing.get("paths_index_s"), "mark_dups_s": timing.get("mark_dups_s"), "dup_pairs": timing.get("dup_pairs")} if args.df_graph else None),
                # content digests of a.paths / a.paths.inv / a.countsb / a.dup and the graph's identities (dfk_paths_digest): the two
                # full-size runs must agree
                "digests": main["digests"], "digests_equal_cold_run": main["digests"] == cold["digests"],
                "cold": {k: cold[k] for k in ("wall_s", "waited_before_s")} | {"breakdown_s": {k: cold["timing"].get(k) for k in ("upload_s", "count_s", "graph_s", "paths_s", "total_s")}},
                "copies": None if not copies else ({"error": copies["error"]} if "error" in copies else
                          {k: copies[k] for k in ("wall_s", "waited_before_s", "pairs", "output_bytes", "frag_reads_orig")} | {"breakdown_s": {k: copies["timing"].get(k) for k in ("upload_s", "count_s", "graph_s", "paths_s", "ingest_outputs_s", "background_join_s", "total_s")}}),
                "host": ("C++ sharded host, %d rank(s), transport %s" % (max(1, args.df_gpus), args.df_transport or "rccl")) if (args.df_gpus > 1 or args.df_transport) else "single GPU (dfk_count)",
                "shard_times_s": timing.get("shard"),
                "input_bytes": in_bytes, "output_bytes": main["output_bytes"], "files_on": root, "host_threads": args.df_threads,
                "solid": timing.get("solid"), "kmer_instances": timing.get("kmer_instances"),
                "input_files_written_in_s": round(t_files, 2)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


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
    ap.add_argument("--cpu-sample-reads", type=int, default=3000000,
                    help="reads of the sample the CPU baseline (and the parity check) runs on (about 30 s of reference code on 32 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the reported-only legs (K sweep, repeat-rich genome, DF stage wall-clock)")
    ap.add_argument("--legs", default="ksweep,repeat,rehearsal,pbf,df", help="which reported-only legs run (comma list of ksweep, repeat, rehearsal, pbf, df)")
    ap.add_argument("--pbf-pairs", type=int, default=2_000_000, help="pairs of the ParseBarcodedFastqs leg (configs[0] is ~5 M; 2 M keeps the default run short)")
    ap.add_argument("--extra-steps", type=int, default=3, help="timed steps of each reported-only leg")
    ap.add_argument("--family-copies", type=int, default=0,
                    help="plant this many diverged copies of one 300-bp element in the genome (hot minimizer buckets)")
    ap.add_argument("--low-complexity", type=float, default=0.0, help="share of the genome overwritten with microsatellite stretches")
    ap.add_argument("--ragged-quals", type=float, default=0.0, help="share of the reads with per-base (nBits=2) quality blocks")
    ap.add_argument("--df-pairs", type=int, default=0,
                    help="pairs of the DF-stage wall-clock leg (0 = the whole set); scaled down by itself when inputs + outputs, "
                         "which live in RAM (/dev/shm), would not fit the command's memory allowance (cgroup memory.max)")
    ap.add_argument("--df-dir", default="/dev/shm", help="where the DF leg's files go")
    ap.add_argument("--df-threads", type=int, default=16, help="NUM_THREADS of the DF leg (the box's CPU share for one GPU)")
    ap.add_argument("--sharded-one", action="store_true",
                    help="run the whole set through the SHARDED pipeline with one rank (class-count scan without scattered atomics, "
                         "slices, two-level LDS regroup): the alternative single-GPU design, for comparison (DESIGN.md section 9)")
    ap.add_argument("--df-graph", action=argparse.BooleanOptionalAction, default=True,
                    help="DF leg: GRAPH=True -- the graph, the read paths, the paths index and the duplicate marks in a.<K>/ (rows f-1, f-2, f-4); "
                         "--no-df-graph: ingest + count + kmers.kvec only")
    ap.add_argument("--df-copies", action=argparse.BooleanOptionalAction, default=True,
                    help="DF leg: also run the stage with frag_reads_orig.{fastb,qualp} as copies (the reference's default) on as many pairs as fit")
    ap.add_argument("--df-gpus", type=int, default=1, help="DF leg: NUM_GPUS of the C++ multi-GPU host (DF forks one rank per GPU, RCCL directly)")
    ap.add_argument("--df-transport", default="", choices=["", "rccl", "loopback"],
                    help="DF leg: run the C++ sharded host even with one rank (rccl), or all ranks as threads on one GPU (loopback)")
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
    affinity = bind_to_gpu_node(local)
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
    genome = synth.make_genome(G, SEED, device=dev, family_copies=args.family_copies,
                               low_complexity_frac=args.low_complexity)             # same genome on every rank
    rs = synth.make_reads(genome, hi - lo, SEED + 17 * (rank + 1), ragged_frac=args.ragged_quals)
    del genome
    if multi:          # barcode ids must not collide between ranks
        stride = int(allreduce(int(rs.bc.max().item()) + 1, torch.int64, dist.ReduceOp.MAX))
        rs.bc = torch.where(rs.bc > 0, rs.bc + rank * stride, rs.bc)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()          # the library sizes its HBM budget from what is free when the context is created
    shard = (rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)

    kw = dict(K=args.K, device=local, minimizer_len=args.minimizer, inst_per_item=args.inst_per_item, passes=args.passes,
              hbm_budget_bytes=int(args.hbm_budget_gb * 1e9))
    if args.emulate_world > 1 or args.sharded_one:
        from superplus_amd.dist import DistDfk, ReplicaComm
        d = DistDfk(comm=ReplicaComm(max(1, args.emulate_world)), **kw)
        def step():
            d.count_device(*shard, read_id0=0)
            return d.stats()
    elif not multi:
        d = Dfk(**kw)
        def step():
            d.count_device(*shard)
            return d.stats()
    else:
        from superplus_amd.dist import DistDfk
        d = DistDfk(**kw)
        def step():
            d.count_device(*shard, read_id0=2 * lo)
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
    each = []                                  # (dfk_count_device returns when the step is done: host clock per step)
    for _ in range(args.steps):
        t1 = time.perf_counter()
        st = step()
        each.append(round(1e3 * (time.perf_counter() - t1), 1))
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
    d.close()                                        # the legs below make their own contexts
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = n_inst * args.steps / elapsed
        k_ms = st["ms_count"]
        k_inst = n_inst // world                                   # instances this rank's k_count launches handled
        achieved = B_INST * k_inst / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        full = G == 3_100_000_000 and total_pairs == 900_000_000 and not (args.family_copies or args.low_complexity or args.ragged_quals)
        traffic = profiled_traffic(st["n_inst"])
        if args.emulate_world > 1:
            label = (f"REHEARSAL, not a benchmark configuration: rank 0 of {args.emulate_world} holding 1/{args.emulate_world} of the set "
                     "(every peer replaced by a replica of this rank; no transfers): ")
        elif full:
            label = "BASELINE configs[%d]: " % (1 if world == 1 else 2)
        else:
            label = "scaled-down run: "
        out = {
            "metric": "k-mers/s (DF createDict stage: trim + canonical k-mer count + solid filter + spectrum + adjacency)",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": label +
                                   f"synthetic stLFR, {total_pairs} pairs 2x100 bp over a {args.genome_mb:g} Mb random genome "
                                   f"({200.0 * total_pairs / G:.1f}x), 0.5% subst., 10% unbarcoded, K={args.K}, MIN_FREQ=3, MIN_BC=2, MIN_QUAL=7" +
                                   (f", {args.family_copies} copies of a 300-bp repeat family" if args.family_copies else "") +
                                   (f", {100 * args.low_complexity:g}% microsatellites" if args.low_complexity else "") +
                                   (f", {100 * args.ragged_quals:g}% reads with per-base quality blocks" if args.ragged_quals else ""),
                       "reads_total": 2 * total_pairs, "kmer_instances_total": n_inst, "K": args.K,
                       "parallelism": "single GPU, bucket-range passes" if world == 1 else
                                      f"{world} ranks: read shards, all-to-all of super-k-mer records by minimizer bucket"},
            "host_affinity": affinity,                # (rank 0's; bind_to_gpu_node)
            "step_ms_each_rank0": each,
            "step_wall_s": elapsed / args.steps,      # one pass of the hot path, inputs resident in HBM (NOT the DF stage's wall-clock: see df_stage)
            **({"rehearsal": f"rank 0 of {args.emulate_world} against replicas of itself; per-rank time without the transfers"}
               if args.emulate_world > 1 else {}),
            "stage_ms_rank0": {k: round(st[k], 3) for k in ("ms_trim", "ms_part_count", "ms_part_scatter", "ms_count",
                                                           "ms_fallback", "ms_adjacency", "ms_total")},
            "counts_rank0": {k: st[k] for k in ("n_reads", "n_inst", "n_records", "n_buckets", "n_items", "n_overflow_items",
                                                "n_distinct", "n_solid", "adj_probes", "hbm_bytes_peak", "n_passes")},
            # `achieved` is the contract's figure: MODEL bytes (64 B per instance, an HBM-resident table) over the kernel's
            # time.  This design keeps the table in LDS: the kernel's physical HBM traffic is `traffic` (counters), ~17x less.
            "roofline": {"bound": "hbm", "kernel": "k_count", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "achieved_is": "algorithmic (model) bytes / kernel time, SURVEY 8d; not a measured HBM rate",
                         "physical_hbm_gbs": (traffic / (k_ms * 1e-3 / max(1, st["n_passes"])) / 1e9) if traffic and k_ms > 0 else None,
                         "algorithmic_bytes_per_launch": B_INST * k_inst // max(1, st["n_passes"]),
                         "kernel_ms_all_launches": k_ms, "launches_per_step": st["n_passes"]},
        }
        if dist_timing is not None:
            out["dist_timing_ms"] = dist_timing
        if world == 1 and not args.no_cpu_baseline and args.emulate_world <= 1:
            # The bounded sample: the workload at its own coverage over a proportionally smaller genome, so that the
            # reference's reduce sees the workload's k-mer multiplicities (the FIRST 6 M reads of the 1.8 G are 0.19x
            # of the genome: not one solid k-mer -- nothing to compare, and no dictionary for the CPU to build).
            sp = min(total_pairs, max(1, args.cpu_sample_reads // 2))
            Gs = max(2000, int(G * sp / total_pairs))
            try:
                gs = synth.make_genome(Gs, SEED + 5, device=dev, family_copies=args.family_copies * Gs // G,
                                       low_complexity_frac=args.low_complexity)
                rss = synth.make_reads(gs, sp, SEED + 6, ragged_frac=args.ragged_quals)
                del gs
                out["cpu_baseline"], out["parity_check"] = cpu_baseline_and_parity(
                    rss, args.K, local, f"the workload's generator at the workload's coverage over a {Gs / 1e6:g} Mb genome: {sp} pairs")
                del rss
            except Exception as e:                       # the headline number must not depend on a reported-only leg
                out["cpu_baseline_error"] = repr(e)
        if world == 1 and not args.no_extras and args.emulate_world <= 1 and not multi:
            extra = {}
            out["extra"] = extra
            # C5: the K sweep on the same reads
            sweep = {}
            legs = set(args.legs.split(","))
            for K2 in ((40, 48, 60) if "ksweep" in legs else ()):
                if K2 == args.K:
                    sweep[str(K2)] = {"kmers_per_s": value, "step_s": round(elapsed / args.steps, 4), "n_passes": st["n_passes"],
                                      "n_inst": st["n_inst"], "n_solid": st["n_solid"], "n_items": st["n_items"],
                                      "n_overflow_items": st["n_overflow_items"], "ms_count": round(st["ms_count"], 2),
                                      "ms_part_count": round(st["ms_part_count"], 2), "ms_fallback": round(st["ms_fallback"], 2)}
                    continue
                try:
                    d2 = Dfk(**dict(kw, K=K2))
                    secs, s2 = timed_steps(d2, shard, args.extra_steps)
                    sweep[str(K2)] = leg_summary(secs, s2)
                    d2.close()
                except Exception as e:
                    sweep[str(K2)] = {"error": repr(e)}
            extra["k_sweep"] = sweep
            # rows f-1 / f-2 / f-4 of a sharded run, rehearsed on one GPU: every rank builds the whole graph, then paths ITS pair range
            # and sorts / groups ITS share of the pairs and keys (df_shard.h).  Here: the whole set counted, the graph built, then
            # rank 0's share at G = 2 and 8 -- the first 1/G of the reads pathed over the whole graph, their index pairs sorted, their
            # duplicates marked.  No exchange (a real run adds one all-to-all of 8-byte pairs and one of 16-byte keys).
            if "rehearsal" in legs:
                fr = {}
                try:
                    d4 = Dfk(**kw)
                    d4.count_device(*shard)
                    n_all = rs.n_reads
                    for Gr in (8, 2):
                        t1 = time.perf_counter(); g4 = d4.graph_build(); torch.cuda.synchronize(); t_graph = time.perf_counter() - t1
                        n = (n_all // 2 // Gr) * 2
                        nb, nq = int(rs.base_off[n].item()), int(rs.pq_off[n].item())
                        t1 = time.perf_counter()
                        p4 = d4.paths_build_device(rs.packed[:nb + 8], rs.base_off[: n + 1], rs.read_len[:n], rs.pq_bytes[:nq + 8], rs.pq_off[: n + 1])
                        t_paths = time.perf_counter() - t1
                        t1 = time.perf_counter(); d4.paths_index_write(None); t_index = time.perf_counter() - t1
                        t1 = time.perf_counter(); d4.dups_write(None); t_dups = time.perf_counter() - t1
                        fr[str(Gr)] = {"reads_of_this_rank": n, "graph_s": round(t_graph, 3), "path_reads_s": round(t_paths, 3), "paths_index_s": round(t_index, 3),
                                       "mark_dups_s": round(t_dups, 3), "reads_placed": p4["n_placed"], "graph_edges": g4["n_edges"]}
                    d4.close()
                    extra["f_rows_rehearsal"] = dict(fr, note="REHEARSAL on one GPU, not a multi-GPU measurement: the device time of rank 0's share of rows f-1 (whole graph, every rank), "
                                                              "f-2 (its reads) and f-4 (its pairs and keys) at G ranks, no files written, no exchange (G = 1, the whole set, is what df_stage times)")
                except Exception as e:
                    extra["f_rows_rehearsal"] = dict(fr, error=repr(e))
                    d4 = None
                torch.cuda.synchronize(); torch.cuda.empty_cache()
            del shard, rs
            torch.cuda.empty_cache()
            # a genome with repeats: ~10 % in one diverged 300-bp family, 1 % microsatellites; a quarter of the reads
            # carry per-base quality blocks (k_trim's bit-unpack path)
            fam = int(0.10 * G / 300)
            try:
                if "repeat" not in legs:
                    raise RuntimeError("leg not selected")
                genome = synth.make_genome(G, SEED + 1, device=dev, family_copies=fam, low_complexity_frac=0.01)
                rs2 = synth.make_reads(genome, total_pairs, SEED + 18, ragged_frac=0.25)
                del genome
                torch.cuda.synchronize(); torch.cuda.empty_cache()
                d2 = Dfk(**kw)
                secs, s2 = timed_steps(d2, (rs2.packed, rs2.base_off, rs2.read_len, rs2.pq_bytes, rs2.pq_off, rs2.bc), args.extra_steps)
                extra["repeat_genome"] = dict(leg_summary(secs, s2), ms_trim=round(s2["ms_trim"], 2),
                                              workload=f"{fam} diverged copies of a 300-bp element (10 % of the genome), 1 % microsatellites, "
                                                       "25 % of the reads with per-base (nBits=2) quality blocks; otherwise the headline set")
                d2.close()
                del rs2
            except Exception as e:
                extra["repeat_genome"] = {"error": repr(e)}
                d2 = rs2 = genome = None
            torch.cuda.synchronize(); torch.cuda.empty_cache()
            # row (e) without a multi-GPU node: rank 0 of G on its 1/G of the headline set against replicas of itself
            # (ReplicaComm: real volumes and coverage, no bytes moved) -- the per-rank compute time of a G-GPU run
            reh = {}
            for Gr in ((2, 8) if "rehearsal" in legs else ()):
                try:
                    from superplus_amd.dist import DistDfk, ReplicaComm
                    genome = synth.make_genome(G, SEED, device=dev)
                    rs3 = synth.make_reads(genome, total_pairs // Gr, SEED + 17)
                    del genome
                    torch.cuda.synchronize(); torch.cuda.empty_cache()
                    d3 = DistDfk(comm=ReplicaComm(Gr), **kw)
                    sh3 = (rs3.packed, rs3.base_off, rs3.read_len, rs3.pq_bytes, rs3.pq_off, rs3.bc)
                    d3.count_device(*sh3, read_id0=0)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.extra_steps):
                        d3.count_device(*sh3, read_id0=0)
                    torch.cuda.synchronize()
                    secs = (time.perf_counter() - t1) / args.extra_steps
                    s3 = d3.stats()
                    reh[str(Gr)] = {"per_rank_step_s": round(secs, 4), "pairs_per_rank": total_pairs // Gr, "n_passes": s3["n_passes"],
                                    "ms_scan": round(s3["ms_part_count"], 1), "ms_scatter_regroup": round(s3["ms_part_scatter"], 1),
                                    "ms_count": round(s3["ms_count"], 1), "ms_adjacency": round(s3["ms_adjacency"], 1),
                                    "bytes_sent_to_peers": d3.timing.get("bytes_sent_to_peers"),
                                    "kmers_per_s_if_all_ranks_took_this_long": round(st["n_inst"] / secs)}
                    d3.close()
                    del rs3, sh3, d3
                except Exception as e:
                    reh[str(Gr)] = {"error": repr(e)}
                torch.cuda.synchronize(); torch.cuda.empty_cache()
            if reh:
                extra["sharded_rehearsal"] = dict(reh, note="REHEARSAL on one GPU, not a multi-GPU measurement: rank 0 of G against replicas of "
                                                            "itself; the all-to-all is a device copy (a real run hides it under the counts, DESIGN.md section 6)")
            # row f-3: ParseBarcodedFastqs (the stage in front of DF, runall.sh:125) on a synthetic fastq.gz pair of configs[0]'s shape:
            # the program's default path (pair order, 2-bit packing, PQVec encoder on the device) beside its own HOST=True path
            if "pbf" in legs:
                try:
                    extra["parse_barcoded_fastqs"] = pbf_leg(args, local)
                except Exception as e:
                    extra["parse_barcoded_fastqs"] = {"error": repr(e)}
            try:
                if "df" not in legs:
                    raise RuntimeError("leg not selected")
                out["df_stage"] = df_stage_wall(args, dev, local)
            except Exception as e:
                out["df_stage"] = {"error": repr(e)}
            for k in ("df_stage_wall_s", "df_stage_wall_cold_s", "df_stage_wall_copies_s"):
                out[k] = out["df_stage"].get(k)
            out["df_stage_device_quiesce_s"] = out["df_stage"].get("device_quiesce_s")
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
