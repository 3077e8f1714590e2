rm -f gpurun_out/xfer.txt
for c in 524288 1048576 2097152; do
  DFK_XFER_CHUNK=$c python3 bench.py --no-cpu-baseline --legs df --steps 1 --warmup 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)['df_stage']; print('chunk=$c', d.get('df_stage_wall_s'), d.get('breakdown_s'))
" >> gpurun_out/xfer.txt
done
