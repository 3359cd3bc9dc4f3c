# GPU tier + soaks on the device + default bench (round 3, first full pass)
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gputests_a.txt 2>&1; tail -4 gpurun_out/r3/gputests_a.txt
timeout 500 python tests/fuzz_consumers_vs_reference.py --device 0 --seeds 400 --start 140000 --budget 330 2>&1 | cut -c1-160 > gpurun_out/r3/soak_consumers_gpu.txt; tail -2 gpurun_out/r3/soak_consumers_gpu.txt
timeout 400 python tests/fuzz_parity.py --seeds 400 --start 150000 2>&1 | cut -c1-160 > gpurun_out/r3/soak_parity_gpu.txt; tail -2 gpurun_out/r3/soak_parity_gpu.txt; grep -c "OK  " gpurun_out/r3/soak_parity_gpu.txt; grep -c FAIL gpurun_out/r3/soak_parity_gpu.txt
timeout 300 python tests/fuzz_parity.py --se --seeds 400 --start 151000 2>&1 | cut -c1-160 > gpurun_out/r3/soak_parity_se_gpu.txt; grep -c "OK  " gpurun_out/r3/soak_parity_se_gpu.txt; grep -c FAIL gpurun_out/r3/soak_parity_se_gpu.txt
timeout 900 python bench.py > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err; tail -2 gpurun_out/r3/bench_a.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3/bench_a.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
o=d["ontarget"]; print("ontarget", o["value"], o["ms_per_step"], o["host_ms_per_call"], {k:(v["avg_launch_ms"], v["frac_of_hbm_peak"]) for k,v in o["kernel_rooflines"].items()})
print("front_end", {k:v for k,v in d["front_end"].items() if k.endswith("per_s")})
PY
