mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp10.txt
: > $O
for ht in 0 24 32 48 64; do
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --ontarget-tput-ctxs 0 --no-front-end --tune host_threads=$ht > gpurun_out/r3/exp10_ht$ht.json 2>> gpurun_out/r3/exp10.err
  python - <<PY >> $O
import json
d=json.loads(open("gpurun_out/r3/exp10_ht$ht.json").read().strip().splitlines()[-1])
print("host_threads $ht: value %.4g ms/step %.1f host_ms_per_call %.1f" % (d["value"], d["ms_per_step"], d["host_ms_per_call"]))
PY
done
cat $O
bash tools/final_collect.sh r3a
