mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp6.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or fresh" 2>&1 | tail -3 >> $O
timeout 1200 python tools/exp_gap.py 4194304 - gap_round2_waves=1792 gap_round2_waves=1536 gap_round2_waves=1280 >> $O 2>&1
timeout 600 python tools/exp_gap.py 1048576 - >> $O 2>&1
cat $O
