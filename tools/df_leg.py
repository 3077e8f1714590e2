#!/usr/bin/env python3
"""Print the DF-stage leg of a bench.py JSON line (tools/df_leg.py gpurun_out/b.log)."""
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith('{'): continue
    j = json.loads(l)
    d = j.get('extra', {}).get('df_stage') or j.get('df_stage') or next(v for v in j.get('extra', {}).values() if isinstance(v, dict) and 'df_stage_wall_s' in v)
    for k, v in d.items():
        if k in ('workload', 'definition', 'host_memory', 'digests', 'frag_reads_orig', 'files_on'): continue
        print(k, json.dumps(v)[:700])
