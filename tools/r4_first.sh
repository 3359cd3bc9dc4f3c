#!/bin/bash
# round 4, first device run of the device-resident records: GPU test tier, the host-phase trace of one on-target call, the on-target bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_gputests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r4a_gputests.log
tail -5 gpurun_out/r4a_gputests.log
timeout 600 python tools/gap_paths.py 4194304 trace=1 > gpurun_out/r4a_trace_ont4m.txt 2>&1
tail -45 gpurun_out/r4a_trace_ont4m.txt
timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 4 --warmup 2 --no-cpu-baseline --no-front-end > gpurun_out/r4a_bench_ont.json 2> gpurun_out/r4a_bench_ont.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r4a_bench_ont.json').read().strip().splitlines()[-1])
    print('ontarget value %.3e ms_per_step %.1f host_ms_per_call %s stage %s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call'), d.get('stage_ms_per_call')))
except Exception as e:
    print('bench parse failed', e)
PY
tail -5 gpurun_out/r4a_bench_ont.err
