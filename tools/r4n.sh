#!/bin/bash
# search-loop variants (FQ_GAP_V bits) on one resident on-target batch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
P=${1:-4194304}
{
for v in 8 11; do
  echo "## FQ_GAP_V=$v"
  FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfq_xv$v.so timeout 900 python tools/exp_gap.py $P "-" "gap_fast_pct=50" "gap_fast_pct=75" "gap_fast_pct=90" "gap_fast_pct=100" "gap_fast_pct=75,gap_fast_pct2=75" "gap_fast_pct=75,gap_fast_pct2=50" 2>&1 | grep -v "^reads made" | cut -c1-330
done
} > $O/r4n2.txt 2>&1
cat $O/r4n2.txt
