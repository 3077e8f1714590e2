# tools/cfg_sweep.sh -- k_count table size / waves per workgroup variants (built with -DDFK_LOG2S= -DDFK_NWAVES= into variants/) at full scale
rm -f gpurun_out/cfg_sweep.txt
for v in "variants/libdfk_s11_w12.so DFK_S2_SAME_PRIO=1" "variants/libdfk_s11_w10.so DFK_S2_SAME_PRIO=1" "variants/libdfk_s11_w12.so DFK_DEFER_SWEEP_BELOW=0" "variants/libdfk_s11_w10.so DFK_DEFER_SWEEP_BELOW=0"; do
  set -- $v
  env DFK_LIB=$1 $2 python3 bench.py --no-cpu-baseline --no-extras --steps 2 2>&1 | python3 -c "
import sys,json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['counts_rank0']; s=d['stage_ms_rank0']; ok=True; print('[$v]', round(d['ms_per_step'],1), 'count', s['ms_count'], 'scatter', s['ms_part_scatter'], 'passes', c['n_passes'], 'solid', c['n_solid'])
if not ok: print('[$v] failed')
" >> gpurun_out/cfg_sweep.txt
done
