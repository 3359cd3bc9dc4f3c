Q="--no-resident --no-ontarget --no-cpu-baseline --no-front-end --ontarget-tput-ctxs 0 --no-host-budget --workdir /tmp/fq_bench"
sum() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$1', 'value %.3e' % d['value'], 'ms/step', d['ms_per_step'], 'prep avg ms', r['avg_launch_ms'], 'frac', r['frac'], 'stage', d['stage_ms_per_call'])
"; }

python bench.py --steps 8 --tune prep_priority=1 $Q 2>/dev/null | sum prio10k

python bench.py --steps 6 --warmup 2 --markers 100000 --tune prep_priority=1 $Q 2>/dev/null | sum prio100k
