export GPU_MAX_HW_QUEUES=20
mkdir -p gpurun_out/r3
timeout 1500 python bench.py --steps 6 --warmup 2 --no-resident --no-ontarget --no-cpu-baseline > gpurun_out/r3/exp5_bench.json 2> gpurun_out/r3/exp5_bench.err
tail -5 gpurun_out/r3/exp5_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3/exp5_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print("front_end", json.dumps(d.get("front_end"), indent=1))
PY
