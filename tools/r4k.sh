mkdir -p gpurun_out
timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -v > gpurun_out/r4k_gputests.log 2>&1; echo rc=$?
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/r4k_gputests.log | tail -3
grep -n "Fatal\|Segmentation\|File \"" gpurun_out/r4k_gputests.log | head -20
tail -5 gpurun_out/r4k_gputests.log | cut -c1-300
