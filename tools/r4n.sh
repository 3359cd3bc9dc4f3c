#!/bin/bash
# search-loop variants (FQ_GAP_V bits) and the pop-batching knob on one resident on-target batch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
P=${1:-4194304}
{
for v in 0 1 2 4 7; do
  echo "## FQ_GAP_V=$v"
  FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfq_xv$v.so timeout 600 python tools/exp_gap.py $P "-" "gap_pop_min=16" "gap_pop_min=32" "gap_pop_min=32,gap_pop_min2=16" 2>&1 | grep -v "^reads made" | cut -c1-330
done
echo "## path sets per trip, first round (instrumented build, mode 3)"
FQ_INSTR_RAW=1 FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_instr.so timeout 600 python tools/exp_gap.py $P "-" "gap_pop_min=32" 2>&1 | grep -v "^reads made" | cut -c1-330
} > $O/r4n.txt 2>&1
cat $O/r4n.txt
