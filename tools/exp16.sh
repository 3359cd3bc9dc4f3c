# four steps per trip on one-row intervals (FqJump) in the round without gap children: off / on; parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1500 python tools/exp_gap.py 4194304 gap_jump=0 - > $O/exp16_gap.txt 2>&1
timeout 600 python tools/exp_gap.py 1048576 gap_jump=0 - >> $O/exp16_gap.txt 2>&1
grep -v "^reads made" $O/exp16_gap.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
