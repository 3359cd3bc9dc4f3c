Q="--no-resident --no-ontarget --no-cpu-baseline --no-front-end --ontarget-tput-ctxs 0 --no-host-budget --workdir /tmp/fq_bench"
sum() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'value %.3e' % d['value'], 'ms/step', d['ms_per_step'], 'cpu/call', d['host_cpu_ms_per_call'], 'wall/call', d['wall_ms_per_call'], 'wait', d['device_wait_ms_per_call'])
"; }
python bench.py --steps 8 $Q 2>/dev/null | sum full
python bench.py --steps 8 --host-cpus 2 $Q 2>/dev/null | sum cpus2
python bench.py --steps 8 --host-cpus 4 $Q 2>/dev/null | sum cpus4
python bench.py --steps 8 --host-cpus 2 --ctxs 8 --pairs 8388608 $Q 2>/dev/null | sum cpus2_8x8M
python bench.py --steps 8 --host-cpus 2 --ctxs 8 $Q 2>/dev/null | sum cpus2_8x4M
