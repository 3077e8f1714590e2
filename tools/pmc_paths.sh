#!/bin/bash
# tools/pmc_paths.sh -- on the GPU box from the repo root: HBM read / write counters (separate passes, as MI355X_MICROARCH.md
# prescribes) for the kernels of rows f-1, f-2 and f-4 on a 200 M-pair set, next to their durations.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_paths
rm -rf $OUT; mkdir -p $OUT
CMD="python3 tools/paths_prof.py 690 200000000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_paths/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out
fs, ws = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
dur = {}
for f in glob.glob("gpurun_out/pmc_paths/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)): dur[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
with open("gpurun_out/pmc_paths/summary.csv", "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "launches", "total_ms", "FETCH_SIZE_KiB_total", "WRITE_SIZE_KiB_total", "read_GBps_raw", "read_GBps_x2", "write_GBps"])
    for k, (n, ns) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        if "dfk::" not in k or ("graph" not in k and "path" not in k and "filter" not in k and "dup" not in k and "pidx" not in k and "rs_" not in k and "qual" not in k): continue
        f, wr = sum(fs.get(k, [])), sum(ws.get(k, []))
        s = ns / 1e9
        w.writerow([k[:90], n, f"{ns / 1e6:.1f}", f"{f:.0f}", f"{wr:.0f}", f"{f * 1024 / s / 1e9:.0f}", f"{2 * f * 1024 / s / 1e9:.0f}", f"{wr * 1024 / s / 1e9:.0f}"])
print(open("gpurun_out/pmc_paths/summary.csv").read())
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
