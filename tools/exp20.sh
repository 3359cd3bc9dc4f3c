# A/B on one box: search starting on strand 1's root (default) vs both roots pushed; unpack by 16-byte pieces; host fills by threads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python tools/exp_gap.py 4194304 - > $O/exp20_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_pb.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp20_gap.txt 2>&1
timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp20_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_pb.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp20_gap.txt 2>&1
grep -v "^reads made" $O/exp20_gap.txt | cut -c1-330
timeout 600 python tools/gap_paths.py 4194304 trace=1 2>&1 | tail -34 | grep -v arena > $O/exp20_trace.txt
cat $O/exp20_trace.txt | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or golden or bench_call" 2>&1 | tail -2
