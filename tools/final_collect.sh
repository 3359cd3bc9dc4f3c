# The round's closing evidence in one gpurun call: GPU test tier, smoke(), the default bench, the host-phase trace of one on-target
# call, the kernel timeline of two on-target streams, the profiled bench runs (rocprofv3 --kernel-trace --stats) and the FETCH_SIZE /
# WRITE_SIZE passes -- one --pmc pass per counter, with --kernel-trace only.  usage: tools/final_collect.sh [tag]   -> gpurun_out/<tag>_*
TAG=${1:-fin}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/${TAG}_gpu_tests.txt
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/${TAG}_smoke.txt 2>&1
( time timeout 1200 python bench.py > $O/${TAG}_default_bench.json 2> $O/${TAG}_default_bench.err ) 2> $O/${TAG}_default_bench.time
bash tools/experiment.sh trace $TAG 4194304 > /dev/null
bash tools/experiment.sh timeline $TAG 2 > /dev/null
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
bash tools/experiment.sh stats $TAG wgs --steps 20 --warmup 5 --no-front-end --no-cpu-baseline > /dev/null
bash tools/experiment.sh stats $TAG wgs_mainleg --steps 20 --warmup 5 $Q > /dev/null
bash tools/experiment.sh stats $TAG ont4m --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 $Q > /dev/null
bash tools/experiment.sh stats $TAG ont --mix ontarget --pairs 1048576 --ctxs 1 --steps 3 --warmup 1 $Q > /dev/null
bash tools/experiment.sh stats $TAG 100k --markers 100000 --steps 6 --warmup 2 --no-cpu-baseline --no-resident --no-front-end --ontarget-tput-ctxs 0 > /dev/null
bash tools/experiment.sh stats $TAG 76bp_ontarget --mix ontarget --read-len 76 --pairs 1048576 --ctxs 2 --steps 3 --warmup 1 $Q > /dev/null
cd /tmp && export TMPDIR=/tmp
pmc() {   # name, counter, bench args...
  local name=$1 ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${TAG}_pmc_${name}_$ctr -o p -- python3 $R/bench.py "$@" > $O/${TAG}_pmc_${name}_$ctr.json 2> $O/${TAG}_pmc_${name}_$ctr.err
  find $O/${TAG}_pmc_${name}_$ctr -name '*kernel_trace.csv' -delete
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc wgs $c --steps 4 --warmup 2 $Q
  pmc ont $c --mix ontarget --pairs 1048576 --ctxs 1 --steps 2 --warmup 1 $Q
  pmc ont4m $c --mix ontarget --pairs 4194304 --ctxs 1 --steps 2 --warmup 1 $Q
done
cat $O/${TAG}_gpu_tests.txt $O/${TAG}_smoke.txt $O/${TAG}_default_bench.time | tail -8
