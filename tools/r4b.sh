mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cli_multidevice.py tests/test_gpu_parity.py -m gpu -x -q -k "multidevice or fq_list or golden" > gpurun_out/r4b_gputests.log 2>&1; tail -3 gpurun_out/r4b_gputests.log
timeout 900 python bench.py > gpurun_out/r4b_default_bench.json 2> gpurun_out/r4b_default_bench.err; tail -3 gpurun_out/r4b_default_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4b_default_bench.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.2f host_ms/call %s wall_ms/call %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call'], d['wall_ms_per_call']))
print('roofline', json.dumps(d['roofline'])[:600])
o = d['ontarget']
print('ontarget value %.4g ms/step %.1f host_ms %s dev %s' % (o['value'], o['ms_per_step'], o['host_ms_per_call'], o['device_ms_per_call']))
print('ontarget rooflines', {k: (v['avg_launch_ms'], v['frac_of_hbm_peak']) for k, v in o['kernel_rooflines'].items()})
print('tput', o.get('throughput'))
print('front', d.get('front_end', {}).get('cli_e2e'))
PY
bash tools/experiment.sh stats r4b ont4m --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget --no-front-end
bash tools/experiment.sh timeline r4b 2
bash tools/experiment.sh trace r4b 4194304
