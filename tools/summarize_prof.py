#!/usr/bin/env python3
"""tools/summarize_prof.py ROUND -- fold the rocprofv3 CSVs that tools/profile_gpu.sh left under
gpurun_out/prof_ROUND/ into small committed files under profiles/:

  profiles/ROUND_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (dfk kernels + totals)
  profiles/ROUND_hbm_counters.csv   per-kernel mean FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
  profiles/ROUND_traffic.json       per-launch HBM bytes of the dominant kernel, corrected as
                                    MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and reads
                                    exactly half of a 16 B/lane coalesced stream on gfx950; WRITE_SIZE
                                    is exact for 16-B stores) -- bench.py quotes it as roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = list(csv.DictReader(open(newest(src + "/trace/*/*_kernel_stats.csv"))))
with open(os.path.join(dst, f"{R}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
    for r in stats:
        if "dfk::" in r["Name"]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for r in csv.DictReader(open(newest(src + f"/{which}/*/*_counter_collection.csv"))):
        if "dfk::" in r["Kernel_Name"]:
            ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{R}_hbm_counters.csv"), "w", newline="") as f:
    w = csv.writer(f)
    # The guide's x2 on FETCH_SIZE is calibrated for wide coalesced streams (16 B per lane) only.  These kernels read
    # their bulk that way (one uint4 of a record or entry per lane); for the others (4/8-byte loads, scattered probes,
    # atomics) the width is uncalibrated: the raw figure is a lower bound and the doubled one an upper bound.
    WIDE = ("k_count<", "k_hot_split<", "k_big_insert<", "k_boundary_list", "k_fill_holes", "k_digest")
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "read_bytes_raw", "read_bytes_x2", "x2_correction", "write_bytes"])
    for k, v in ctr.items():
        fs = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        ws = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        wide = any(t in k for t in WIDE)
        w.writerow([k, len(v["FETCH_SIZE"]), f"{fs:.1f}", f"{ws:.1f}", int(fs * 1024), int(2 * fs * 1024),
                    "calibrated: 16 B/lane streaming reads" if wide else "uncalibrated access width: read bytes lie between raw and x2", int(ws * 1024)])

dom = [k for k in ctr if "k_count<" in k][0]
bench = [l for l in open(os.path.join(src, "bench_trace.log")) if l.startswith("{")]
line = json.loads(bench[-1]) if bench else {}
runs = int(line.get("steps", 3)) + int(line.get("warmup", 1))               # bench steps executed under the profiler
passes = int((line.get("counts_rank0") or {}).get("n_passes", 1)) or 1
# k_count is launched once per bucket-range pass (plus tiny relaunches for split items): total its traffic over the
# whole run and divide by the number of main launches
fs_tot, ws_tot = sum(ctr[dom]["FETCH_SIZE"]), sum(ctr[dom]["WRITE_SIZE"])
main = runs * passes
tot_ns = [float(r["TotalDurationNs"]) for r in stats if "k_count<" in r["Name"]][0]
json.dump({"round": R, "kernel": dom, "main_launches": main, "avg_ns_per_main_launch_rocprof": tot_ns / main,
           "fetch_size_kib_total": fs_tot, "write_size_kib_total": ws_tot,
           "hbm_bytes_per_launch": int((2 * fs_tot + ws_tot) * 1024 / main),
           "correction": "read bytes = 2 x FETCH_SIZE KiB (gfx950, 16 B/lane coalesced stream); write bytes = WRITE_SIZE KiB",
           "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras",
           "bench_line_under_profiler": line or None},
          open(os.path.join(dst, f"{R}_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{R}_kernel_stats.csv")).read())
print(open(os.path.join(dst, f"{R}_hbm_counters.csv")).read())
