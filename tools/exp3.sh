export GPU_MAX_HW_QUEUES=20
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp3.txt
: > $O
timeout 1500 python tools/exp_gap.py 4194304 gap_round2_waves=0 gap_round2_waves=2560 gap_round2_waves=2048 gap_round2_waves=1536 gap_round2_waves=1024 gap_round2_waves=2048,gap_refill_min=32 >> $O 2>&1
cat $O
