set -x
export GPU_MAX_HW_QUEUES=20
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp2.txt
: > $O
FQ_LIB_EXPERIMENT=fastquick_amd/libfastquick_amd_instr.so timeout 600 python tools/exp_gap.py 1048576 - >> $O 2>&1
timeout 900 python tools/exp_gap.py 4194304 gap_waves_per_cu=8 gap_refill_min=8 gap_refill_min=1 >> $O 2>&1
for dt in 0 1 2; do
  timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --ontarget-tput-ctxs 0 --tune device_turns=$dt > gpurun_out/r3/exp2_bench_dt$dt.json 2>> $O
done
cat $O
for dt in 0 1 2; do python - <<PY
import json
d=json.loads(open("gpurun_out/r3/exp2_bench_dt$dt.json").read().strip().splitlines()[-1])
print("dt$dt", d["value"], d["ms_per_step"], json.dumps(d.get("roofline")), json.dumps(d.get("kernel_rooflines", d.get("config"))))
PY
done
