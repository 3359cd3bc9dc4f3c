# k_width: LDS by seed length (16 KB per block), 5 / 6 / 8 wavefronts per SIMD; first pop skipped in the search
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python tools/exp_gap.py 4194304 - > $O/exp19_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_w6.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp19_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_w8.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp19_gap.txt 2>&1
grep -v "^reads made" $O/exp19_gap.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
