#!/bin/bash
# search-stage experiments on one resident on-target batch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 900 python tools/exp_gap.py 4194304 "-" "-" 2>&1 | grep -v "^reads made" | cut -c1-330
timeout 900 python tools/exp_gap.py 1048576 "-" "gap_skip_bound=0" 2>&1 | grep -v "^reads made" | cut -c1-330
timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget --no-front-end > $O/r4p_ont1.json 2>$O/r4p.err
python3 -c "
import json
d=json.loads(open('$O/r4p_ont1.json').read().strip().splitlines()[-1])
print('ontarget 1 stream value %.4g ms_per_step %.1f'%(d['value'],d['ms_per_step']), {k:(round(v.get('frac_of_hbm_peak',0),3),round(v.get('avg_launch_ms',0),2)) for k,v in d['kernel_rooflines'].items() if isinstance(v,dict)})
print(d['roofline'].get('device_ms_per_call'))
"
} > $O/r4p.txt 2>&1
cat $O/r4p.txt
