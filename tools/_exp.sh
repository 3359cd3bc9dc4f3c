timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_tables or streams_run" 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-resident --no-ontarget --no-cpu-baseline --front-end-copies 0 --ontarget-tput-ctxs 0 --no-host-budget --workdir /tmp/fq_bench > /dev/null 2>&1
bash tools/cli_trace.sh /tmp/fq_bench 64 qc 2>&1 | tail -14
FASTQUICK_TRACE=1 fastquick_amd/bin/FASTQuick_amd align --index_prefix $(ls /tmp/fq_bench/*.FASTQuick.fa | head -1 | sed 's/.FASTQuick.fa$//') --fastq_1 /tmp/fq_bench/front_end/trace_1.fq.gz --fastq_2 /tmp/fq_bench/front_end/trace_2.fq.gz --out_prefix /tmp/fq_bench/front_end/t2 --sam_out --read_len 151 2>&1 >/dev/null | grep "index load"
