python bench.py --steps 2 --warmup 1 --no-resident --no-ontarget --no-cpu-baseline --front-end-copies 0 --ontarget-tput-ctxs 0 --no-host-budget --workdir /tmp/fq_bench > /dev/null 2>&1
F=/tmp/fq_bench/front_end
bash tools/cli_trace.sh /tmp/fq_bench 64 qc 2>&1 | grep "index released\|front end on the device"
for v in "" pad20 pad16; do
  L=$PWD/fastquick_amd/libfastquick_amd${v:+_$v}.so
  echo "variant ${v:-default}"
  FQ_LIB_EXPERIMENT=$L python3 tools/frontend_stream.py $F/trace_1.fq.gz $F/trace_2.fq.gz --repeats 3 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    s = d['stats']; print('  stream', d['s'], 'infl', s['ms_inflate'], 'tok', s['ms_tokenise'], 'wait_reader', s['ms_wait_reader'])
"
  P=$(ls /tmp/fq_bench/*.FASTQuick.fa | head -1); P=${P%.FASTQuick.fa}
  for r in 1 2; do
  LD_PRELOAD=$L FASTQUICK_TRACE=1 fastquick_amd/bin/FASTQuick_amd align --index_prefix $P --fastq_1 $F/trace_1.fq.gz --fastq_2 $F/trace_2.fq.gz --out_prefix $F/t3 --sam_out --read_len 151 --t 32 2>&1 >/dev/null | grep "index released\|front end on the device" | cut -c1-200
  done
done
