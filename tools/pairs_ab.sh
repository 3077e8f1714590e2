#!/bin/bash
# A/B of k_count's pair extraction (two consecutive k-mers of a record per lane) against one k-mer per lane:
# parity first, then the 100 Mb slice (with the phase-clock builds) and configs[1] at full size.
#   variants/libdfk_pairs.so      hipcc ... -DDFK_PAIRS -DDFK_PAIRS_SERIAL
#   variants/libdfk_phase.so        hipcc ... -DDFK_PHASE_TIMES          (make phase)
#   variants/libdfk_pairs_phase.so  both
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_parity.log 2>&1 || { tail -30 gpurun_out/ab_parity.log; exit 1; }
tail -2 gpurun_out/ab_parity.log
SL="--genome-mb 100 --pairs 15000000 --no-cpu-baseline --no-extras --steps 10 --warmup 2"
for v in "" variants/libdfk_pairs.so variants/libdfk_phase.so variants/libdfk_pairs_phase.so; do
  n=$(basename "${v:-default}" .so)
  DFK_LIB=$v python bench.py $SL > gpurun_out/ab_slice_$n.log 2> gpurun_out/ab_slice_$n.err || { tail -5 gpurun_out/ab_slice_$n.err; exit 1; }
  python - "$n" gpurun_out/ab_slice_$n.log <<'PY'
import json, sys
o = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "slice ms_per_step %.2f" % o["ms_per_step"], {k: round(v, 2) for k, v in o["stage_ms_rank0"].items()})
PY
  grep -A14 "wave cycles by phase" gpurun_out/ab_slice_$n.err | tail -15
done
for v in "" variants/libdfk_pairs.so; do
  n=$(basename "${v:-default}" .so)
  DFK_LIB=$v python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > gpurun_out/ab_full_$n.log 2> gpurun_out/ab_full_$n.err || { tail -5 gpurun_out/ab_full_$n.err; exit 1; }
  python - "$n" gpurun_out/ab_full_$n.log <<'PY'
import json, sys
o = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "FULL ms_per_step %.1f" % o["ms_per_step"], o["step_ms_each_rank0"], {k: round(v, 1) for k, v in o["stage_ms_rank0"].items()}, "frac %.3f" % o["roofline"]["frac"])
PY
done
