#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for z in 1 0; do
FASTQUICK_ZLIB_INFLATE=$z timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget > $O/r4o_$z.json 2> $O/r4o_$z.err
python3 -c "
import json
d=json.loads(open('$O/r4o_$z.json').read().strip().splitlines()[-1])
fe=d['front_end']
print('zlib_only=$z tokenise_inflate', fe.get('tokenise_inflate_pairs_per_s'), fe.get('tokenise_inflate'))
for k,v in (fe.get('cli_e2e_steady') or {}).items():
    if isinstance(v,dict): print('  ', k, v.get('pairs_per_s'), v.get('wall_s'), v.get('notices'))
"
done
