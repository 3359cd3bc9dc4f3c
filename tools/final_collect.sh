R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/fin_gpu_tests.txt
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/fin_smoke.txt 2>&1
python bench.py > $O/fin_default_bench.json 2> $O/fin_default_bench.err
python tools/gap_paths.py 4194304 trace=1 2>&1 | tail -34 | grep -v arena > $O/fin_trace.txt
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/fin_${name}_prof -o p -- python3 $R/bench.py "$@" > $O/fin_${name}_bench.json 2> $O/fin_${name}_bench.err
  cp $(find $O/fin_${name}_prof -name '*kernel_stats.csv' | head -1) $O/fin_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/fin_${name}_prof
}
prof wgs --steps 20 --warmup 5
prof ont --mix ontarget --pairs 1048576 --ctxs 1 --steps 3 --warmup 1 --no-cpu-baseline
cat $O/fin_gpu_tests.txt $O/fin_smoke.txt | tail -5
