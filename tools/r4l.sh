mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4l_gputests.log 2>&1; echo rc=$?; tail -4 gpurun_out/r4l_gputests.log | cut -c1-200
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for cfg in "2 device_turns=2" "2 device_turns=3" "3 device_turns=3" "4 device_turns=3"; do
set -- $cfg
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs $1 --steps 4 --warmup 2 $Q --tune $2 > gpurun_out/r4l.json 2> gpurun_out/r4l.err
python -c "
import json
d = json.loads(open('gpurun_out/r4l.json').read().strip().splitlines()[-1])
print('ontarget ctxs $1 $2 value %.4g ms_per_step %.1f host_ms_per_call %s dev %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call'), d['roofline']['device_ms_per_call']))"
done
