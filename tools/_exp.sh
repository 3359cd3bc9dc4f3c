timeout 600 python -m pytest tests/test_device_frontend.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 2 --warmup 1 --no-resident --no-ontarget --no-cpu-baseline --front-end-copies 0 --ontarget-tput-ctxs 0 --no-host-budget --workdir /tmp/fq_bench > /dev/null 2>&1
F=/tmp/fq_bench/front_end
for i in 1 2; do for e in 1 2; do cat $F/reads_$e.fq.gz > /dev/null; done; done
bash tools/cli_trace.sh /tmp/fq_bench 64 qc 2>&1 | tail -2
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
FASTQUICK_FE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5c_fe_solo -o fe -- python3 $OLDPWD/tools/frontend_stream.py $F/trace_1.fq.gz $F/trace_2.fq.gz --repeats 1 > $O/r5c_fe_solo.json 2> $O/r5c_fe_solo.err
find $O/r5c_fe_solo -name '*kernel_trace.csv' -delete
head -9 $O/r5c_fe_solo/fe_kernel_stats.csv | cut -c1-110
cd $OLDPWD
python3 tools/frontend_stream.py $F/trace_1.fq.gz $F/trace_2.fq.gz --repeats 3 | cut -c1-420
