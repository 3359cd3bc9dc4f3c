#!/bin/bash
# two on-target streams: persistent search kernels that leave a few wave slots free for the other stream's small kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
{
for cfg in "2 0" "2 15" "2 14" "2 12" "3 0" "3 15" "1 15" "2 0"; do
  set -- $cfg
  T=""; [ "$2" != "0" ] && T="--tune gap_waves_per_cu=$2"
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs $1 --steps 4 --warmup 1 $Q $T > $O/r4q.json 2>> $O/r4q.err
  python3 -c "
import json
d=json.loads(open('$O/r4q.json').read().strip().splitlines()[-1])
k=d['kernel_rooflines']
print('ctxs $1 waves_per_cu $2: value %.4g ms/step %.1f  nogap %.2f full %.2f width %.2f ms  gap frac %.3f' % (d['value'], d['ms_per_step'], k['fq_gap_nogap']['avg_launch_ms'], k['fq_gap_full']['avg_launch_ms'], k['fq_width']['avg_launch_ms'], k['fq_gap']['frac_of_hbm_peak']))"
done
} > $O/r4q.txt 2>&1
cat $O/r4q.txt
