# steps ahead per trip in the round without gap children (FqGapLane::chain_ahead): 0 / 2 (default) / 3 / 4 / 6
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1500 python tools/exp_gap.py 4194304 gap_chain_ahead=0 - gap_chain_ahead=3 gap_chain_ahead=4 gap_chain_ahead=6 > $O/exp14_gap.txt 2>&1
timeout 600 python tools/exp_gap.py 1048576 - >> $O/exp14_gap.txt 2>&1
grep -v "^reads made" $O/exp14_gap.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
