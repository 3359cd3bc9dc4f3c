mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp7.txt
: > $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $O
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for c in 1 8 16; do
  timeout 600 python bench.py --ctxs $c --steps 20 --warmup 5 $Q > gpurun_out/r3/exp7_ctx$c.json 2>> gpurun_out/r3/exp7.err
  python - <<PY >> $O
import json
d=json.loads(open("gpurun_out/r3/exp7_ctx$c.json").read().strip().splitlines()[-1])
print("ctxs $c value %.4g ms/step %.2f stage_ms_per_call %s host %.1f wall %.1f" % (d["value"], d["ms_per_step"], d["stage_ms_per_call"], d["host_ms_per_call"], d["wall_ms_per_call"]))
PY
done
cat $O
