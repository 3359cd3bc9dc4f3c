#!/bin/bash
# One gpurun call's worth of evidence for profiles/: the default bench under rocprofv3 --kernel-trace --stats, the on-target
# bench the same way, and separate FETCH_SIZE / WRITE_SIZE passes for both (MI355X_MICROARCH.md: one --pmc pass per counter,
# never combined with the trace domains gpurun refuses).  usage: tools/collect_round.sh <tag>   -> gpurun_out/<tag>_*
# Copy what is to be judged into profiles/ afterwards (tools/pmc_summarize.py writes profiles/pmc_traffic.json).
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${name}_prof -o p -- python3 $R/bench.py "$@" > $O/${TAG}_${name}_bench.json 2> $O/${TAG}_${name}_bench.err
  cp $(find $O/${TAG}_${name}_prof -name '*kernel_stats.csv' | head -1) $O/${TAG}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/${TAG}_${name}_prof
}
pmc() {   # name, counter, bench args...
  local name=$1 ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${TAG}_pmc_${name}_$ctr -o p -- python3 $R/bench.py "$@" > $O/${TAG}_pmc_${name}_$ctr.json 2> $O/${TAG}_pmc_${name}_$ctr.err
  find $O/${TAG}_pmc_${name}_$ctr -name '*kernel_trace.csv' -delete
}
prof wgs --steps 20 --warmup 5
prof ont --mix ontarget --pairs 1048576 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline
Q="--no-cpu-baseline --no-resident --no-ontarget"
for c in FETCH_SIZE WRITE_SIZE; do
  pmc wgs $c --steps 4 --warmup 2 $Q
  pmc ont $c --mix ontarget --pairs 1048576 --ctxs 1 --steps 2 --warmup 1 $Q
done
