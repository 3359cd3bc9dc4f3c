mkdir -p gpurun_out
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for t in "device_turn_min=1048576" "device_turn_min=0" "device_turns=0"; do
timeout 900 python bench.py --markers 100000 --steps 5 --warmup 2 $Q --tune $t > gpurun_out/r4e_100k_$t.json 2> gpurun_out/r4e_100k.err
python -c "
import json
d = json.loads(open('gpurun_out/r4e_100k_$t.json').read().strip().splitlines()[-1])
print('100k $t value %.4g ms_per_step %.1f host_ms %s dev %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call'], d['roofline']['device_ms_per_call']))"
done
for n in 8 24; do
timeout 900 python bench.py --markers 100000 --ctxs $n --steps 5 --warmup 2 $Q > gpurun_out/r4e_100k_c$n.json 2>> gpurun_out/r4e_100k.err
python -c "
import json
d = json.loads(open('gpurun_out/r4e_100k_c$n.json').read().strip().splitlines()[-1])
print('100k ctxs $n value %.4g ms_per_step %.1f host_ms %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call']))"
done
