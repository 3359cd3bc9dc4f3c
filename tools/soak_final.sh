# closing soak of a round on its final build: oracle vs device (stage dumps + SAM), adversarial / ragged / single-end variants, consumers vs the real
# reference where it is built, the device front end vs the host reader + packer and its member decoder vs zlib (round 5)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${SOAK_TAG:-r5}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() { # name, cmd...
  local name=$1; shift
  timeout 900 "$@" > $O/soak_$name.log 2>&1
  echo "$name: rc=$? OK=$(grep -c ' OK ' $O/soak_$name.log) FAIL=$(grep -c 'FAIL' $O/soak_$name.log) last: $(tail -1 $O/soak_$name.log | cut -c1-100)"
}
{ run parity python tests/fuzz_parity.py --seeds 200 --start ${SOAK_BASE:-400000}
  run adversarial python tests/fuzz_parity.py --seeds 60 --start $((${SOAK_BASE:-400000}+1000)) --adversarial
  run ragged python tests/fuzz_parity.py --seeds 40 --start $((${SOAK_BASE:-400000}+1500)) --ragged
  run se python tests/fuzz_parity.py --seeds 120 --start $((${SOAK_BASE:-400000}+2000)) --se
  run frontend python tests/fuzz_frontend.py --seeds 1500 --start $((${SOAK_BASE:-400000}+5000))
  run consumers python tests/fuzz_consumers_vs_reference.py --seeds 400 --start $((${SOAK_BASE:-400000}+10000)) --device 0 --budget 300
} > $O/soak_final.txt 2>&1
cat $O/soak_final.txt
