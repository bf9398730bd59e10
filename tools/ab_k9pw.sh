#!/bin/bash
# cfg2 9-1-5 per-layer times, product library, three runs (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 1 2 3; do
  timeout -k 10 120 python $R/tools/bench_variants.py --only cfg2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{') and not l.startswith('['): continue
    ds=json.loads(l); ds=ds if isinstance(ds,list) else [ds]
    for d in ds:
        print(d['workload'][:22], d['ms_per_step'], {k:v['us'] for k,v in d['layers'].items()})
" || exit 1
done
