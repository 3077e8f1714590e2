# tools/cfg_sweep.sh -- build variants of libdfk (k_count waves per workgroup, sweep list size, per-run classes) into variants/
# and run each at full scale: step time, k_count time, sweep time.  Round 3 (gpurun_out/cfg_sweep.txt):
#   default (8 waves)               1508 ms   count 967   sweeps 863 (9 passes)
#   -DDFK_RUN_CLASSES               1504      count 963   sweeps 898 (11 passes: 81.7 ms a sweep instead of 95.9)
#   -DDFK_NWAVES=10                 2050      count 881   sweeps 1295: they no longer run under the counts
#   -DDFK_NWAVES=10 + run classes   2006      count 881   sweeps 1017
#   -DDFK_NWAVES=12 2-KB lists      1947      count 818   sweeps 633 (alone: after the counts, not under them)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -Wall -Wno-unused-function -Wno-pass-failed"
mkdir -p variants gpurun_out
for v in "rc:-DDFK_RUN_CLASSES" "w10:-DDFK_NWAVES=10" "w10_rc:-DDFK_NWAVES=10 -DDFK_RUN_CLASSES" "w12_r4:-DDFK_NWAVES=12 -DDFK_SWEEP_READS=4"; do
  n=${v%%:*}; f=${v#*:}; [ -f variants/libdfk_$n.so ] || /opt/rocm/bin/hipcc $FL $f -shared -o variants/libdfk_$n.so superplus_amd/csrc/dfk.hip
done
rm -f gpurun_out/cfg_sweep.txt
for v in superplus_amd/libdfk.so variants/libdfk_rc.so variants/libdfk_w10.so variants/libdfk_w10_rc.so variants/libdfk_w12_r4.so; do
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
