# tools/defer_test.sh -- does launching every range's sweep behind k_count cost time, and does it remove the slow passes a profiler provokes?
export TMPDIR=/tmp
for v in 0.045 1.0; do
  DFK_DEFER_SWEEP_BELOW=$v python3 bench.py --no-cpu-baseline --no-extras --steps 4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('defer_below=$v plain', d['step_ms_each_rank0'], d['stage_ms_rank0']['ms_count'], d['stage_ms_rank0']['ms_part_scatter'])
" >> gpurun_out/defer_test.txt
done
for v in 0.045 1.0; do
  rm -rf gpurun_out/defer_prof_$v
  DFK_DEFER_SWEEP_BELOW=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/defer_prof_$v -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('defer_below=$v rocprof', d['step_ms_each_rank0'], d['stage_ms_rank0']['ms_count'], d['stage_ms_rank0']['ms_part_scatter'])
" >> gpurun_out/defer_test.txt
  find gpurun_out/defer_prof_$v -name '*_kernel_trace.csv' -delete
done
