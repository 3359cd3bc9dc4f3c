mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fresh or option" > gpurun_out/r4h_gputests.log 2>&1; tail -3 gpurun_out/r4h_gputests.log
timeout 900 python tests/fuzz_parity.py --seeds 80 --start 270000 --adversarial > gpurun_out/r4h_soak_adv.log 2>&1; echo "adversarial rc=$? OK=$(grep -c ' OK ' gpurun_out/r4h_soak_adv.log) FAIL=$(grep -c FAIL gpurun_out/r4h_soak_adv.log)"
timeout 900 python tests/fuzz_parity.py --seeds 80 --start 271000 > gpurun_out/r4h_soak.log 2>&1; echo "parity rc=$? OK=$(grep -c ' OK ' gpurun_out/r4h_soak.log) FAIL=$(grep -c FAIL gpurun_out/r4h_soak.log)"
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for t in sw_serial_reverse=0 sw_serial_reverse=1; do
timeout 600 python bench.py --ctxs 1 --steps 6 --warmup 3 $Q --tune $t > gpurun_out/r4h_wgs1.json 2> gpurun_out/r4h_wgs1.err
python -c "
import json
d = json.loads(open('gpurun_out/r4h_wgs1.json').read().strip().splitlines()[-1])
print('wgs 1 stream $t value %.4g ms_per_step %.2f host_ms %s dev %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call'], d['roofline']['device_ms_per_call']))"
done
for n in 4 8; do
timeout 600 python bench.py --ctxs $n --steps 10 --warmup 3 $Q > gpurun_out/r4h_wgs$n.json 2> gpurun_out/r4h_wgs$n.err
python -c "
import json
d = json.loads(open('gpurun_out/r4h_wgs$n.json').read().strip().splitlines()[-1])
print('wgs $n streams value %.4g ms_per_step %.2f host_ms %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call']))"
done
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 $Q > gpurun_out/r4h_ont1.json 2> gpurun_out/r4h_ont1.err
python -c "
import json
d = json.loads(open('gpurun_out/r4h_ont1.json').read().strip().splitlines()[-1])
print('ontarget 1 stream value %.4g ms_per_step %.1f dev %s' % (d['value'], d['ms_per_step'], d['roofline']['device_ms_per_call']))"
