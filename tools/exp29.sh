R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 600 python tools/gap_paths.py 4194304 trace=1 2>&1 | tail -32 | grep -v arena | cut -c1-120 | tail -24
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs 2 --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget --no-front-end 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ont2 value %.4g ms/step %.1f host %.1f'%(d['value'],d['ms_per_step'],d['host_ms_per_call']))"
