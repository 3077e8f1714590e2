#!/bin/bash
# tools/try_lib.sh LIB... -- bench the 100 Mb slice with alternative builds of libdfk.so (GPU box)
for lib in "$@"; do
  cp superplus_amd/libdfk.so /tmp/libdfk_keep.so
  cp "$lib" superplus_amd/libdfk.so
  echo "$lib: $(python bench.py --genome-mb 100 --pairs 15000000 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), d["stage_ms_rank0"])')"
  cp /tmp/libdfk_keep.so superplus_amd/libdfk.so
done
