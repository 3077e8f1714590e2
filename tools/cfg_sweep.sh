# tools/cfg_sweep.sh -- k_count waves per workgroup / sweep list size variants (built with -DDFK_NWAVES= -DDFK_SWEEP_READS=
# into variants/) at full scale: step time, k_count time, sweep time
rm -f gpurun_out/cfg_sweep.txt
for v in "superplus_amd/libdfk.so" "variants/libdfk_norc.so" "variants/libdfk_w10.so" "variants/libdfk_w10_norc.so" "variants/libdfk_w10_r4.so" "variants/libdfk_w12_r4.so"; do
  env DFK_LIB=$v python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys,json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['counts_rank0']; s=d['stage_ms_rank0']; ok=True; print('[$v]', round(d['ms_per_step'],1), 'count', round(s['ms_count'],1), 'scatter', round(s['ms_part_scatter'],1), 'scan', round(s['ms_part_count'],1), 'passes', c['n_passes'], 'solid', c['n_solid'])
if not ok: print('[$v] failed')
" >> gpurun_out/cfg_sweep.txt
done
cat gpurun_out/cfg_sweep.txt
