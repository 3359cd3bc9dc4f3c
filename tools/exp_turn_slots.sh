# (experiment) cfg 3 -- 100k markers, WGS mix, sixteen streams -- with K search stages side by side instead of sixteen: bench.py --tune device_turn_min=100000,device_turn_slots=K
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
Q="--markers 100000 --steps 6 --warmup 2 --no-cpu-baseline --no-resident --no-front-end --ontarget-tput-ctxs 0 --no-ontarget --no-host-budget"
for K in 0 1 2 3 4 6 8; do
  if [ $K = 0 ]; then T=""; else T="--tune device_turn_min=100000,device_turn_slots=$K"; fi
  python3 bench.py $Q $T 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('slots', '$K', 'value %.3e' % d['value'], 'ms_per_step', d['ms_per_step'], 'fq_prep frac', d['roofline']['frac'])"
done
