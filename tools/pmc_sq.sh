#!/bin/bash
# tools/pmc_sq.sh TAG [DFK_LIB] -- where k_count's wave cycles go, from SQ counters on the 100 Mb slice (GPU box).
# Three rocprofv3 --pmc passes (8 SQ slots each, --kernel-trace only); per dfk kernel the counters summed over its dispatches.
export TMPDIR=/tmp
TAG=${1:-sq}; LIB=${2:-}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
[ -f gpurun_out/counters_gfx950.txt ] || rocprofv3 -L > gpurun_out/counters_gfx950.txt 2>&1
CMD="python3 bench.py --genome-mb 100 --pairs 15000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
pick() { for c in "$@"; do grep -qw "$c" gpurun_out/counters_gfx950.txt && echo -n "$c "; done; }
A=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA)
B=$(pick SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAVES)
C=$(pick SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN)
echo "pass a: $A"; echo "pass b: $B"; echo "pass c: $C"
DFK_LIB=$LIB rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/a -- $CMD > $OUT/bench_a.log 2>&1 || { tail -5 $OUT/bench_a.log; exit 1; }
DFK_LIB=$LIB rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/b -- $CMD > $OUT/bench_b.log 2>&1 || { tail -5 $OUT/bench_b.log; exit 1; }
[ -n "$C" ] && { DFK_LIB=$LIB rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/c -- $CMD > $OUT/bench_c.log 2>&1 || tail -5 $OUT/bench_c.log; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], '*', '**', '*_counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'dfk::' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
with open(os.path.join(sys.argv[1], 'summary.txt'), 'w') as out:
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        if v.get('SQ_WAVE_CYCLES', 0) < 1e7: continue
        line = k[:60] + '\n   ' + '\n   '.join(f"{c:26s} {x:.4e}" + (f"  ({x / v['SQ_WAVE_CYCLES']:.3f} of wave cycles)" if c.startswith(('SQ_WAIT', 'SQ_ACTIVE', 'SQ_INST_CYCLES')) else '') for c, x in sorted(v.items()))
        print(line); out.write(line + '\n')
PY
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*_counter_collection.csv' -size +8M -delete
