mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4c_gputests.log 2>&1; tail -4 gpurun_out/r4c_gputests.log
bash tools/experiment.sh stats r4c ont4m --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget --no-front-end | head -14
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 4 --warmup 2 --no-cpu-baseline --no-front-end > gpurun_out/r4c_bench_ont.json 2> gpurun_out/r4c_bench_ont.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4c_bench_ont.json').read().strip().splitlines()[-1])
print('ontarget 2 streams value %.4g ms_per_step %.1f host_ms_per_call %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call')))
print(d['roofline'].get('device_ms_per_call'))
PY
