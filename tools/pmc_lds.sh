#!/bin/bash
# tools/pmc_lds.sh TAG -- LDS counters of the 100 Mb slice bench per kernel: two rocprofv3 --pmc passes (GPU box).
# Prints, per dfk kernel, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE etc. summed over its dispatches.
export TMPDIR=/tmp
TAG=${1:-lds}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --genome-mb 100 --pairs 15000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/a -- $CMD > $OUT/bench_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $CMD > $OUT/bench_b.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], '*', '**', '*_counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'dfk::' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
with open(os.path.join(sys.argv[1], 'summary.txt'), 'w') as out:
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
        if v.get('GRBM_GUI_ACTIVE', 0) < 1e6: continue
        line = k[:70] + ' ' + ' '.join(f"{c}={x:.4e}" for c, x in sorted(v.items()))
        if v.get('SQ_LDS_IDX_ACTIVE'): line += f" conflict/idx_active={v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE']:.3f}"
        if v.get('SQ_ACTIVE_INST_LDS'): line += f" conflict/active_inst_lds={v['SQ_LDS_BANK_CONFLICT']/v['SQ_ACTIVE_INST_LDS']:.3f}"
        print(line); out.write(line + '\n')
PY
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*_counter_collection.csv' -size +8M -delete
