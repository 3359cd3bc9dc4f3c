set -x
export GPU_MAX_HW_QUEUES=20
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or fresh" 2>&1 | tail -5 > gpurun_out/r3/exp1_parity.txt
timeout 1200 python tools/exp_gap.py 4194304 - gap_no_order=1 > gpurun_out/r3/exp1_gap4m.txt 2>&1
timeout 600 python tools/exp_gap.py 1048576 - >> gpurun_out/r3/exp1_gap4m.txt 2>&1
cat gpurun_out/r3/exp1_parity.txt gpurun_out/r3/exp1_gap4m.txt
