mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp8.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fresh" 2>&1 | tail -2 >> $O
timeout 1500 python tools/exp_gap.py 4194304 gap_round2_lane_major=0 - gap_round2_waves=2560 gap_round2_waves=3072 gap_round2_waves=1536 >> $O 2>&1
timeout 600 python tools/exp_gap.py 1048576 gap_round2_lane_major=0 - >> $O 2>&1
cat $O
