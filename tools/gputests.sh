export GPU_MAX_HW_QUEUES=20
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gputests.txt 2>&1
tail -15 gpurun_out/r3/gputests.txt
