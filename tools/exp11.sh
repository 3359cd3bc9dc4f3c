mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp11.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or cli or bench_call or ontarget_call" 2>&1 | tail -2 >> $O
for ht in -1 16 32; do
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --ontarget-tput-ctxs 0 --no-front-end --tune host_threads=$ht > gpurun_out/r3/exp11_ht$ht.json 2>> gpurun_out/r3/exp11.err
  python - <<PY >> $O
import json
t=open("gpurun_out/r3/exp11_ht$ht.json").read().strip().splitlines()
d=json.loads(t[-1])
print("1 stream host_threads $ht: value %.4g ms/step %.1f host_ms_per_call %.1f" % (d["value"], d["ms_per_step"], d["host_ms_per_call"]))
PY
done
timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-front-end > gpurun_out/r3/exp11_2s.json 2>> gpurun_out/r3/exp11.err
python - <<PY >> $O
import json
d=json.loads(open("gpurun_out/r3/exp11_2s.json").read().strip().splitlines()[-1])
print("2 streams: value %.4g ms/step %.1f host %.1f ; tput leg %s" % (d["value"], d["ms_per_step"], d["host_ms_per_call"], d.get("throughput")))
PY
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-resident --no-ontarget > gpurun_out/r3/exp11_wgs.json 2>> gpurun_out/r3/exp11.err
python - <<PY >> $O
import json
d=json.loads(open("gpurun_out/r3/exp11_wgs.json").read().strip().splitlines()[-1])
print("wgs: value %.4g ms/step %.1f host %.1f front_end %s" % (d["value"], d["ms_per_step"], d["host_ms_per_call"], {k:v for k,v in d["front_end"].items() if k.endswith("per_s")}))
print(d["front_end"]["cli_e2e"])
PY
cat $O
