#!/bin/bash
# tools/profile_r04.sh -- round 4's committed profiles (GPU box, from the repo root):
#   tools/profile_gpu.sh r04 + summarize        kernel stats, HBM counters, k_count's traffic of the default bench command
#   timeline of the last step                   tools/timeline.py on the kernel trace
#   ParseBarcodedFastqs' device path            kernel stats of one run on 2 M synthetic pairs
set -e
export TMPDIR=/tmp
tools/profile_gpu.sh r04
python3 tools/summarize_prof.py r04 > gpurun_out/prof_r04/summary.txt
T=$(ls -t gpurun_out/prof_r04/trace/*/*_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 tools/timeline.py gpurun_out/prof_r04/trace > profiles/r04_timeline.txt || echo "no kernel trace kept" > profiles/r04_timeline.txt
mkdir -p gpurun_out/pbf_prof
python3 -c "
from superplus_amd import synth
synth.write_fastq_pair('gpurun_out/pbf_prof/r_1.fq.gz', 'gpurun_out/pbf_prof/r_2.fq.gz', 2000000, 20261007, threads=16)"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pbf_prof/trace -- superplus_amd/ParseBarcodedFastqs "FASTQS={gpurun_out/pbf_prof/r_1.fq.gz,gpurun_out/pbf_prof/r_2.fq.gz}" OUT_HEAD=gpurun_out/pbf_prof/o/reads NUM_THREADS=16 NUM_BUCKETS=10 > gpurun_out/pbf_prof/run.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/pbf_prof/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
with open("profiles/r04_pbf_kernel_stats.csv", "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["# rocprofv3 --kernel-trace --stats -- superplus_amd/ParseBarcodedFastqs on 2 M synthetic pairs (2x100 bp, 100 k barcodes): the device path's kernels"])
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct"])
    for r in rows:
        w.writerow([r["Name"].replace("void ", "").split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
print(open("profiles/r04_pbf_kernel_stats.csv").read())
PY
tail -3 gpurun_out/pbf_prof/run.log
rm -rf gpurun_out/pbf_prof/o gpurun_out/pbf_prof/*.gz
find gpurun_out/prof_r04 gpurun_out/pbf_prof -name '*_kernel_trace.csv' -delete
# (gpurun brings back gpurun_out/ only: what this wrote under profiles/ travels in a copy)
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_* gpurun_out/profiles_r04/
head -30 profiles/r04_timeline.txt
