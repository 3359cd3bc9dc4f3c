mkdir -p gpurun_out
( time timeout 1500 python bench.py > gpurun_out/r4f_default_bench.json 2> gpurun_out/r4f_default_bench.err ) 2> gpurun_out/r4f_time.txt
cat gpurun_out/r4f_time.txt; tail -3 gpurun_out/r4f_default_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4f_default_bench.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.2f host_ms/call %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call']))
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'frac', 'dominant_by_device_time', 'device_ms_per_call')})
o = d['ontarget']
print('ontarget value %.4g ms/step %.1f host_ms %s distinct %s' % (o['value'], o['ms_per_step'], o['host_ms_per_call'], o.get('distinct_batches')))
print('tput', o.get('throughput'))
print('front', json.dumps(d.get('front_end'))[:1800])
print('cpu', json.dumps(d.get('cpu_baseline'))[:300])
PY
