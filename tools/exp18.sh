# next entry fetched ahead in the round without gap children too (build with -DFQ_NOGAP_AHEAD=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python tools/exp_gap.py 4194304 - > $O/exp18_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_fa.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp18_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_fa.so timeout 900 python tools/exp_gap.py 1048576 - >> $O/exp18_gap.txt 2>&1
grep -v "^reads made" $O/exp18_gap.txt | cut -c1-330
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_fa.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nogap or pipeline or bench_call or prefix" 2>&1 | tail -2
