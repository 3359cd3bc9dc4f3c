mkdir -p gpurun_out
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for t in gap_long_pops=256 gap_long_pops=128 gap_long_pops=64 gap_long_pops=32 gap_long_pops=512 "gap_long_pops=128,gap_coop_waves=2048" ; do
timeout 600 python bench.py --ctxs 1 --steps 8 --warmup 3 $Q --tune $t > gpurun_out/r4i_wgs1.json 2> gpurun_out/r4i_wgs1.err
python -c "
import json
d = json.loads(open('gpurun_out/r4i_wgs1.json').read().strip().splitlines()[-1])
print('wgs 1 stream $t value %.4g ms_per_step %.2f host_ms %s gap %.2f retries %s' % (d['value'], d['ms_per_step'], d['host_ms_per_call'], d['roofline']['device_ms_per_call']['gap_full'], d['work_per_call']['tier_retries']))"
done
timeout 600 python bench.py --ctxs 1 --steps 4 --warmup 3 $Q --tune trace=1 > gpurun_out/r4i_wgs_trace.json 2> gpurun_out/r4i_wgs_trace.err
tail -24 gpurun_out/r4i_wgs_trace.err | grep -v arena
