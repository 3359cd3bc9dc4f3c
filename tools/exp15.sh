# second round with the stock-option kernel (128 VGPRs): wavefronts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1500 python tools/exp_gap.py 4194304 - gap_round2_waves=2048 gap_round2_waves=3072 gap_round2_waves=3584 gap_round2_waves=4096 gap_round2_waves=3072,gap_refill_min=8 > $O/exp15_gap.txt 2>&1
grep -v "^reads made" $O/exp15_gap.txt | cut -c1-330
