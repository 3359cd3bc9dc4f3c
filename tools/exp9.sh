mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp9.txt
: > $O
timeout 1800 python tools/exp_gap.py 4194304 - gap_round1_refill=48 gap_round1_refill=32 gap_waves_per_cu=12 gap_refill_min=32 >> $O 2>&1
cat $O
