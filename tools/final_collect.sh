# The round's closing evidence in one gpurun call: GPU test tier, smoke(), the default bench, the host-phase trace of one on-target
# call, the profiled bench runs (rocprofv3 --kernel-trace --stats) and the FETCH_SIZE / WRITE_SIZE passes -- one --pmc pass per
# counter, with --kernel-trace only.  usage: tools/final_collect.sh [tag]   -> gpurun_out/<tag>_*
TAG=${1:-fin}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/${TAG}_gpu_tests.txt
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/${TAG}_smoke.txt 2>&1
timeout 900 python bench.py > $O/${TAG}_default_bench.json 2> $O/${TAG}_default_bench.err
timeout 600 python tools/gap_paths.py 4194304 trace=1 2>&1 | tail -34 | grep -v arena > $O/${TAG}_trace.txt
cd /tmp && export TMPDIR=/tmp
prof() {
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
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
prof wgs --steps 20 --warmup 5 --no-front-end
prof wgs_mainleg --steps 20 --warmup 5 $Q
prof ont4m --mix ontarget --pairs 4194304 --ctxs 1 --steps 3 --warmup 1 $Q
prof ont --mix ontarget --pairs 1048576 --ctxs 1 --steps 3 --warmup 1 $Q
for c in FETCH_SIZE WRITE_SIZE; do
  pmc wgs $c --steps 4 --warmup 2 $Q
  pmc ont $c --mix ontarget --pairs 1048576 --ctxs 1 --steps 2 --warmup 1 $Q
  pmc ont4m $c --mix ontarget --pairs 4194304 --ctxs 1 --steps 2 --warmup 1 $Q
done
cat $O/${TAG}_gpu_tests.txt $O/${TAG}_smoke.txt | tail -5
