#!/bin/bash
# SQ counters of the front end's kernels (rocprofv3 --pmc, one pass per counter group, with --kernel-trace only): instruction mix and where the
# wavefronts' cycles go.  usage: tools/frontend_pmc.sh [tag] [records]  -> gpurun_out/<tag>_pmc_fe_*
TAG=${1:-r5}; N=${2:-1000000}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
g1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
g2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
g3="SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
i=0
for g in "$g1" "$g2" "$g3"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/${TAG}_pmc_fe_g$i -o p -- python3 $R/tools/frontend_bench.py --records $N --levels 1 --repeats 2 > $O/${TAG}_pmc_fe_g$i.json 2> $O/${TAG}_pmc_fe_g$i.err
  find $O/${TAG}_pmc_fe_g$i -name '*kernel_trace.csv' -delete
done
python3 $R/tools/pmc_quick.py k_inflate $O/${TAG}_pmc_fe_g1 $O/${TAG}_pmc_fe_g2 $O/${TAG}_pmc_fe_g3 | tee $O/${TAG}_pmc_fe_summary.txt
