#!/bin/bash
# tools/profile_gpu.sh ROUND -- run on the GPU box (via gpurun) from the repo root.
# Three separate rocprofv3 passes over the same bench command: kernel trace + stats, then the
# two HBM traffic counters on their own (they do not fit one pass: MI355X_MICROARCH.md, PMC slots).
set -e
R=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/bench_write.log 2>&1
# the per-launch kernel traces are large and not needed once the stats exist (gpurun returns at most 64 MiB)
find $OUT/fetch $OUT/write -name '*_kernel_trace.csv' -delete
find $OUT/trace -name '*_kernel_trace.csv' -size +20M -delete
find $OUT -name '*.csv' | head -50
