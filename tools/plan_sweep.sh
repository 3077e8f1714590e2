for cfg in "1.1 10" "1.3 16" "1.1 10" "1.3 16" "1.15 10" "1.1 11"; do set -- $cfg; DFK_PLAN_GROWTH=$1 DFK_PLAN_FIRST=$2 python bench.py --no-cpu-baseline --no-extras --steps 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', d['ms_per_step'], d['counts_rank0']['n_passes'], d['stage_ms_rank0']['ms_count'], d['stage_ms_rank0']['ms_part_scatter'])
" >> gpurun_out/plan_sweep.txt; done
