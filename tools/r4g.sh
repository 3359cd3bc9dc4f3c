mkdir -p gpurun_out
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for cfg in "2 device_turns=0" "2 device_turns=1" "3 device_turns=0" "3 device_turns=1" "4 device_turns=0"; do
set -- $cfg
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs $1 --steps 3 --warmup 1 $Q --tune $2 > gpurun_out/r4g.json 2> gpurun_out/r4g.err
python -c "
import json
d = json.loads(open('gpurun_out/r4g.json').read().strip().splitlines()[-1])
print('ontarget ctxs $1 $2 value %.4g ms_per_step %.1f host_ms_per_call %s dev %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call'), d['roofline']['device_ms_per_call']))"
done
