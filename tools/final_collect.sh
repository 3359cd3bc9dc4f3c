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
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end --no-host-budget"
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
# ---- the device front end: kernel durations of a 67 M-pair BGZF stream (rocprofv3 --kernel-trace --stats; kernels one after the other so that each
#      duration is the kernel's own, then the product's overlapped order), the command line's time marks, the member decoder's SQ counters
cd $R
python bench.py --steps 2 --warmup 1 --no-resident --no-ontarget --no-cpu-baseline --front-end-copies 0 --ontarget-tput-ctxs 0 --no-host-budget --workdir /tmp/fq_bench > /dev/null 2>&1
bash tools/cli_trace.sh /tmp/fq_bench 64 qc > $O/${TAG}_cli_trace.txt 2>&1
F=/tmp/fq_bench/front_end
cd /tmp && export TMPDIR=/tmp
FASTQUICK_FE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_fe_solo -o fe -- python3 $R/tools/frontend_stream.py $F/trace_1.fq.gz $F/trace_2.fq.gz --repeats 1 > $O/${TAG}_fe_solo.json 2> $O/${TAG}_fe_solo.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_fe_overlap -o fe -- python3 $R/tools/frontend_stream.py $F/trace_1.fq.gz $F/trace_2.fq.gz --repeats 1 > $O/${TAG}_fe_overlap.json 2> $O/${TAG}_fe_overlap.err
find $O/${TAG}_fe_solo $O/${TAG}_fe_overlap -name '*kernel_trace.csv' -delete
cd $R
# ---- the command line on on-target input under rocprofv3 --kernel-trace --stats: the consumers' kernels (fq_emit.h, fq_deflate.h) beside the alignment's
cd /tmp && export TMPDIR=/tmp
FQ_PROFILE_DIR=$O/${TAG}_cliont_prof FQ_BENCH_DIR=/tmp/fq_e2e timeout 600 python3 $R/tools/cli_ontarget.py 1048576 8 150 > $O/${TAG}_cli_ontarget_profiled.json 2> $O/${TAG}_cli_ontarget_profiled.err
for m in sam_out bam_and_qc; do f=$(find $O/${TAG}_cliont_prof/$m -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_cli_ontarget_${m}_kernel_stats.csv; done
rm -rf $O/${TAG}_cliont_prof
# ... the same with every call waiting for its consumers' kernels (FASTQUICK_EMIT_SYNC=1): the consumers' kernels ALONE on the device, not beside the next call's
FASTQUICK_EMIT_SYNC=1 FQ_PROFILE_DIR=$O/${TAG}_cliont_alone FQ_BENCH_DIR=/tmp/fq_e2e timeout 600 python3 $R/tools/cli_ontarget.py 1048576 8 150 > $O/${TAG}_cli_ontarget_alone.json 2> $O/${TAG}_cli_ontarget_alone.err
for m in sam_out bam_and_qc; do f=$(find $O/${TAG}_cliont_alone/$m -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_cli_ontarget_alone_${m}_kernel_stats.csv; done
rm -rf $O/${TAG}_cliont_alone
# ... how busy the device is over the steady calls of a 33.5 M-pair run (tools/kernel_busy.py over the kernel and copy traces), and the calls' stages (FASTQUICK_CTX_TRACE)
FQ_PROFILE_COPIES=1 FQ_PROFILE_DIR=$O/${TAG}_cliont_busy FQ_BENCH_DIR=/tmp/fq_e2e timeout 900 python3 $R/tools/cli_ontarget.py 1048576 32 150 > $O/${TAG}_cli_ontarget_busy.json 2> $O/${TAG}_cli_ontarget_busy.err
for m in sam_out bam_and_qc; do [ -d $O/${TAG}_cliont_busy/$m ] && python3 $R/tools/kernel_busy.py $O/${TAG}_cliont_busy/$m --mid 0.4 --gaps 16 > $O/${TAG}_cli_ontarget_busy_$m.txt 2>&1; done
rm -rf $O/${TAG}_cliont_busy
FASTQUICK_TRACE=1 FASTQUICK_CTX_TRACE=1 FQ_BENCH_DIR=/tmp/fq_e2e timeout 900 python3 $R/tools/cli_ontarget.py 1048576 32 150 > $O/${TAG}_cli_ontarget_trace.json 2> $O/${TAG}_cli_ontarget_trace.err
cd $R
bash tools/frontend_pmc.sh $TAG 2000000 > /dev/null 2>&1
python3 tools/frontend_bench.py --records 4000000 > $O/${TAG}_inflate_kernel.txt 2>&1
cat $O/${TAG}_gpu_tests.txt $O/${TAG}_smoke.txt $O/${TAG}_default_bench.time | tail -8
