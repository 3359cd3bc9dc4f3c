# the final kernels under the old knobs once more: wavefronts per CU of the first round, its refill group, wavefronts of the second
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1500 python tools/exp_gap.py 4194304 - gap_waves_per_cu=12 gap_round1_refill=32 gap_round1_refill=48 gap_round2_waves=2304 gap_round2_waves=2816 gap_round2_lane_major=0 > $O/exp28_gap.txt 2>&1
grep -v "^reads made" $O/exp28_gap.txt | cut -c1-330
