# CPU time per host phase (process-wide core-ms beside the wall time) of one on-target call and of WGS-mix calls
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 600 python tools/gap_paths.py 4194304 trace=1 2>&1 | tail -34 | grep -v arena > $O/exp23_trace_ont.txt
timeout 600 python tools/gap_paths.py 4194304 trace=1,host_threads=16 2>&1 | tail -34 | grep -v arena > $O/exp23_trace_ont16.txt
timeout 600 python bench.py --steps 2 --warmup 1 --ctxs 1 --no-cpu-baseline --no-resident --no-ontarget --no-front-end --tune trace=1 > $O/exp23_wgs.json 2> $O/exp23_trace_wgs.txt
cat $O/exp23_trace_ont.txt | cut -c1-120
echo ---- 16 threads
cat $O/exp23_trace_ont16.txt | cut -c1-120 | tail -30
echo ---- wgs
tail -45 $O/exp23_trace_wgs.txt | cut -c1-120
