#!/bin/bash
# tools/sweep.sh -- parameter sweeps on the GPU box (run through gpurun from the repo root)
for ipi in 3000 4500 6144 8000 10000; do
  echo "inst_per_item=$ipi $(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inst-per-item $ipi 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), d["stage_ms"], d["counts"]["n_items"], d["counts"]["n_overflow_items"])')"
done
for m in 10 12 13 15 16; do
  echo "minimizer=$m $(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --minimizer $m 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), d["stage_ms"], d["counts"]["n_records"], d["counts"]["adj_probes"])')"
done
