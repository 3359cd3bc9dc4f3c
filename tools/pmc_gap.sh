#!/bin/bash
# PMC passes over the search kernel on the on-target mix (one pass per counter group: rocprofv3 takes at most 8 SQ / 4 TCC counters).
# usage: tools/pmc_gap.sh <tag> [tuning]   -> gpurun_out/pmc_<tag>_<group>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAVES"
G3="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
G4="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1_g$i -o p -- python3 $R/tools/gap_paths.py 524288 $2 > $R/gpurun_out/pmc_$1_g$i.log 2>&1
done
