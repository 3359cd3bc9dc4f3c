# refine: one task per wavefront (default) vs one task per lane (refine_lanes=1) at the on-target call shape
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O; cd $R
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for rl in 0 1; do
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 2 --warmup 1 $Q --tune refine_lanes=$rl > $O/exp27_$rl.json 2>> $O/exp27.err
  python - <<PY
import json
d=json.loads(open("$O/exp27_$rl.json").read().strip().splitlines()[-1])
print("refine_lanes=$rl: value %.4g ms/step %.1f stage %s refine_tasks %s" % (d["value"], d["ms_per_step"], json.dumps(d["stage_ms_per_call"]), d["work_per_call"]["refine_tasks"]))
PY
done
