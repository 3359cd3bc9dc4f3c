# host threads per call when two / three on-target streams share the host: what the box really gives (cgroup quota, affinity) and the throughput per setting
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "nproc: $(nproc)"; grep Cpus_allowed_list /proc/self/status; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"; lscpu | grep -i "numa\|socket\|thread\|model name" ; uptime; } > $O/exp22_host.txt 2>&1
cat $O/exp22_host.txt
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for cfg in "2 12" "2 16" "2 20" "2 24" "2 32" "3 8" "3 12" "3 16" "1 32" "1 24"; do
  set -- $cfg
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs $1 --steps 3 --warmup 1 $Q --tune host_threads=$2 > $O/exp22_c$1_t$2.json 2>> $O/exp22.err
  python - <<PY
import json
d=json.loads(open("$O/exp22_c$1_t$2.json").read().strip().splitlines()[-1])
print("ctxs $1 host_threads $2: value %.4g ms/step %.1f host_ms_per_call %.1f" % (d["value"], d["ms_per_step"], d.get("host_ms_per_call", -1)))
PY
done
