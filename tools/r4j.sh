mkdir -p gpurun_out
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fresh" > gpurun_out/r4j_gputests.log 2>&1; tail -2 gpurun_out/r4j_gputests.log
for t in width_both_strands=0 width_both_strands=1; do
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 $Q --tune $t > gpurun_out/r4j_ont1.json 2> gpurun_out/r4j_ont1.err
python -c "
import json
d = json.loads(open('gpurun_out/r4j_ont1.json').read().strip().splitlines()[-1])
print('ontarget 1 stream $t value %.4g ms_per_step %.1f width %s' % (d['value'], d['ms_per_step'], d['kernel_rooflines']['fq_width']))"
done
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 4 --warmup 2 $Q > gpurun_out/r4j_ont2.json 2> gpurun_out/r4j_ont2.err
python -c "
import json
d = json.loads(open('gpurun_out/r4j_ont2.json').read().strip().splitlines()[-1])
print('ontarget 2 streams value %.4g ms_per_step %.1f dev %s' % (d['value'], d['ms_per_step'], d['roofline']['device_ms_per_call']))"
timeout 900 python bench.py --markers 100000 --steps 5 --warmup 2 $Q > gpurun_out/r4j_100k.json 2> gpurun_out/r4j_100k.err
python -c "
import json
d = json.loads(open('gpurun_out/r4j_100k.json').read().strip().splitlines()[-1])
print('100k value %.4g ms_per_step %.1f dev %s' % (d['value'], d['ms_per_step'], d['roofline']['device_ms_per_call']))"
