mkdir -p gpurun_out
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for n in 2 3 4; do
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs $n --steps 4 --warmup 2 $Q > gpurun_out/r4m.json 2> gpurun_out/r4m.err
python -c "
import json
d = json.loads(open('gpurun_out/r4m.json').read().strip().splitlines()[-1])
print('ontarget ctxs $n value %.4g ms_per_step %.1f host_ms %s cpu_ms %s wait_ms %s gap %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call'), d.get('host_cpu_ms_per_call'), d.get('device_wait_ms_per_call'), d['kernel_rooflines']['fq_gap']['frac_of_hbm_peak']))"
done
timeout 900 python bench.py --mix ontarget --pairs 1048576 --ctxs 16 --steps 3 --warmup 1 $Q > gpurun_out/r4m.json 2> gpurun_out/r4m.err
python -c "
import json
d = json.loads(open('gpurun_out/r4m.json').read().strip().splitlines()[-1])
print('ontarget 16 x 1M value %.4g ms_per_step %.1f host_ms %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call')))"
timeout 900 python bench.py --steps 10 --warmup 3 $Q > gpurun_out/r4m.json 2> gpurun_out/r4m.err
python -c "
import json
d = json.loads(open('gpurun_out/r4m.json').read().strip().splitlines()[-1])
print('wgs 16 value %.4g ms_per_step %.1f host_ms %s cpu_ms %s wait_ms %s wall %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call'), d.get('host_cpu_ms_per_call'), d.get('device_wait_ms_per_call'), d['wall_ms_per_call']))"
