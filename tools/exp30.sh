# A/B on one box, alternating: children's rows computed only where they exist (variant library) vs all eight every step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
for i in 1 2; do
timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp30_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_kk.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp30_gap.txt 2>&1
done
grep -v "^reads made" $O/exp30_gap.txt | cut -c1-330
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_kk.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
