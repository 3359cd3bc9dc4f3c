#!/bin/bash
# search-stage experiments on one resident on-target batch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 900 python tools/exp_gap.py 4194304 "gap_skip_bound=0" "-" "gap_skip_hard=1" "gap_skip_hard=-1" "gap_skip_hard=100" "gap_skip_bound=0" "-" 2>&1 | grep -v "^reads made" | cut -c1-330
timeout 900 python tools/exp_gap.py 1048576 "gap_skip_bound=0" "-" "gap_skip_hard=1" "gap_skip_hard=100" 2>&1 | grep -v "^reads made" | cut -c1-330
} > $O/r4n4.txt 2>&1
cat $O/r4n4.txt
