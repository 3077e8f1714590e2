#!/usr/bin/env python3
"""tools/timeline.py TRACE_DIR -- one step of bench.py as a timeline, from a rocprofv3 --kernel-trace CSV:
per stream (queue), the kernels longer than 0.5 ms and the idle gaps longer than 0.5 ms, for the LAST step
(from its k_trim launch to its last kernel).  Run on the GPU box right after the profile; prints text."""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0],
                 r.get("Queue_Id") or r.get("Stream_Id") or "?"))
rows.sort()
trims = [i for i, r in enumerate(rows) if "k_trim" in r[2]]
lo = trims[-1]
step = rows[lo:]
t0 = step[0][0]
end = max(r[1] for r in step)
print(f"step: {(end - t0) / 1e6:.1f} ms, {len(step)} kernels")
queues = {}
for r in step:
    queues.setdefault(r[3], []).append(r)
for q, ks in queues.items():
    busy = sum(e - s for s, e, _, _ in ks)
    print(f"\nqueue {q}: {len(ks)} kernels, busy {busy / 1e6:.1f} ms")
    prev_end = t0
    acc_small, n_small = 0, 0
    for s, e, name, _ in ks:
        gap = s - prev_end
        if gap > 500_000:
            if n_small:
                print(f"            ... {n_small} short kernels, {acc_small / 1e6:.2f} ms")
                acc_small, n_small = 0, 0
            print(f"  {(prev_end - t0) / 1e6:9.2f}  -- idle {gap / 1e6:.2f} ms --")
        if e - s > 500_000:
            if n_small:
                print(f"            ... {n_small} short kernels, {acc_small / 1e6:.2f} ms")
                acc_small, n_small = 0, 0
            print(f"  {(s - t0) / 1e6:9.2f}  {name[:60]:60s} {(e - s) / 1e6:8.2f} ms")
        else:
            acc_small += e - s; n_small += 1
        prev_end = max(prev_end, e)
    if n_small:
        print(f"            ... {n_small} short kernels, {acc_small / 1e6:.2f} ms")
