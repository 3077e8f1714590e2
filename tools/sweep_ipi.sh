#!/bin/bash
# tools/sweep_ipi.sh -- instances per work item on the 100 Mb slice (GPU box)
for ipi in ${@:-2560 3072 3584 4096 4608 5120}; do
  echo "inst_per_item=$ipi $(python bench.py --genome-mb 100 --pairs 15000000 --steps 3 --warmup 1 --no-cpu-baseline --inst-per-item $ipi 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), d["stage_ms_rank0"]["ms_count"], d["stage_ms_rank0"]["ms_fallback"], d["counts_rank0"]["n_items"], d["counts_rank0"]["n_overflow_items"])')"
done
