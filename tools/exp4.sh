export GPU_MAX_HW_QUEUES=20
mkdir -p gpurun_out/r3
nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|^CPU\(s\)" 
timeout 1500 python bench.py > gpurun_out/r3/exp4_bench.json 2> gpurun_out/r3/exp4_bench.err
tail -3 gpurun_out/r3/exp4_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3/exp4_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print("front_end", json.dumps(d.get("front_end"), indent=1))
print("ontarget", d["ontarget"]["value"], json.dumps(d["ontarget"]["kernel_rooflines"]))
PY
