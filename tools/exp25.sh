# device flag hipDeviceScheduleBlockingSync (default now) vs FASTQUICK_SPIN_WAIT=1: CPU time of the device waits, throughput; quota-aware host threads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for sp in 0 1; do
  if [ $sp = 1 ]; then export FASTQUICK_SPIN_WAIT=1; else unset FASTQUICK_SPIN_WAIT; fi
  timeout 600 python tools/gap_paths.py 4194304 trace=1 2>&1 | tail -32 | grep "A: width\|call \|B3\|flatten" | head -6 | cut -c1-120
  timeout 600 python bench.py --steps 16 --warmup 4 $Q > $O/exp25_wgs_$sp.json 2>> $O/exp25.err
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 3 --warmup 1 $Q > $O/exp25_ont2_$sp.json 2>> $O/exp25.err
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 3 --steps 3 --warmup 1 $Q > $O/exp25_ont3_$sp.json 2>> $O/exp25.err
  python - <<PY
import json
for tag in ("wgs", "ont2", "ont3"):
    d=json.loads(open("$O/exp25_%s_$sp.json" % tag).read().strip().splitlines()[-1])
    print("spin=$sp %s: value %.4g ms/step %.1f host_ms_per_call %.1f wall_ms_per_call %.1f" % (tag, d["value"], d["ms_per_step"], d.get("host_ms_per_call", -1), d.get("wall_ms_per_call", -1)))
PY
done
