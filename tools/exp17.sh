bash tools/final_collect.sh r3b
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in 2 3 4; do
  timeout 900 python bench.py --mix ontarget --pairs 4194304 --ctxs $n --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget --no-front-end > gpurun_out/r3b_ont_ctx$n.json 2> gpurun_out/r3b_ont_ctx$n.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3b_ont_ctx$n.json").read().strip().splitlines()[-1])
print("ctxs $n: value %.4g ms/step %.1f host_ms_per_call %.1f roofline %s" % (d["value"], d["ms_per_step"], d.get("host_ms_per_call", -1), json.dumps(d.get("roofline"))[:300]))
PY
done
