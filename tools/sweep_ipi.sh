rm -f gpurun_out/ipi.txt
for n in 0 2304 2688 3328 3840; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 2 --inst-per-item $n 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['counts_rank0']; s=d['stage_ms_rank0']; print('ipi=$n', round(d['ms_per_step'],1), 'count', s['ms_count'], 'items', c['n_items'], 'overflow', c['n_overflow_items'], 'passes', c['n_passes'])
" >> gpurun_out/ipi.txt
done
