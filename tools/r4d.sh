mkdir -p gpurun_out
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for n in 3 4; do
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs $n --steps 3 --warmup 1 $Q > gpurun_out/r4d_ont_c$n.json 2> gpurun_out/r4d_ont_c$n.err
python -c "
import json
d = json.loads(open('gpurun_out/r4d_ont_c$n.json').read().strip().splitlines()[-1])
print('ontarget ctxs $n value %.4g ms_per_step %.1f host_ms_per_call %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call')))"
done
timeout 600 python bench.py --ctxs 1 --steps 6 --warmup 3 $Q --tune trace=1 > gpurun_out/r4d_wgs1.json 2> gpurun_out/r4d_wgs1.err
tail -32 gpurun_out/r4d_wgs1.err | grep -v arena
python -c "
import json
d = json.loads(open('gpurun_out/r4d_wgs1.json').read().strip().splitlines()[-1])
print('wgs 1 stream value %.4g ms_per_step %.2f host_ms %s dev %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call'], d['roofline']['device_ms_per_call']))"
for n in 4 8; do
timeout 600 python bench.py --ctxs $n --steps 10 --warmup 3 $Q > gpurun_out/r4d_wgs$n.json 2> gpurun_out/r4d_wgs$n.err
python -c "
import json
d = json.loads(open('gpurun_out/r4d_wgs$n.json').read().strip().splitlines()[-1])
print('wgs $n streams value %.4g ms_per_step %.2f host_ms %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call']))"
done
bash tools/experiment.sh stats r4d wgs_mainleg --steps 20 --warmup 5 $Q | head -24
bash tools/experiment.sh stats r4d 100k --markers 100000 --steps 6 --warmup 2 --no-cpu-baseline --no-resident --no-front-end --ontarget-tput-ctxs 0 | head -12
bash tools/experiment.sh stats r4d 76bp_ontarget --mix ontarget --read-len 76 --pairs 1048576 --ctxs 2 --steps 3 --warmup 1 $Q | head -12
for t in 100k 76bp_ontarget; do python -c "
import json
d = json.loads(open('gpurun_out/r4d_${t}_bench.json').read().strip().splitlines()[-1])
print('$t value %.4g ms_per_step %.1f' % (d['value'], d['ms_per_step']), {k: (v['avg_launch_ms'], v['frac_of_hbm_peak']) for k, v in d['kernel_rooflines'].items()})
if 'ontarget' in d: print('   ontarget', d['ontarget']['value'], {k: (v['avg_launch_ms'], v['frac_of_hbm_peak']) for k, v in d['ontarget']['kernel_rooflines'].items()})"; done
