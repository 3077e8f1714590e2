#!/bin/bash
# tools/pmc_slice.sh COUNTERS... -- SQ counters of the 100 Mb slice bench, one rocprofv3 --pmc pass (GPU box)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_slice
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -- python3 bench.py --genome-mb 100 --pairs 15000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = max(glob.glob('gpurun_out/pmc_slice/*/*_counter_collection.csv'), key=lambda p: __import__('os').path.getmtime(p))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'dfk::' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    if max(v.values()) > 1e6: print(k[:60], {c: f"{x:.3e}" for c, x in v.items()})
PY
