#!/bin/bash
# FASTQuick_amd align with FASTQUICK_TRACE=1 on a synthetic WGS-mix BGZF pair (made by bench.py's front-end leg in --workdir): where a run's wall time goes.
# usage: tools/cli_trace.sh <workdir> [copies]
W=${1:-/tmp/fq_bench}; C=${2:-64}
R=$(cd "$(dirname "$0")/.." && pwd)
F=$W/front_end
for e in 1 2; do rm -f $F/trace_$e.fq.gz; for i in $(seq $C); do head -c -28 $F/reads_$e.fq.gz >> $F/trace_$e.fq.gz; done; done
P=$(ls $W/*.FASTQuick.fa | head -1); P=${P%.FASTQuick.fa}
if [ "${3:-}" = qc ]; then   # the QC consumer's inputs beside the index (bench.py removes them after its steady-state leg)
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from fastquick_amd import synth
ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
pre = "$P.FASTQuick.fa"
synth.write_qc_inputs(pre, ref); synth.write_param(pre, ref, 1000)
open(pre + ".genome.fa.fai", "w").write("1\t%d\t3\t60\t61\n" % len(ref.genome))
PY
fi
[ "${4:-}" = stream ] && python3 $R/tools/frontend_stream.py $F/trace_1.fq.gz $F/trace_2.fq.gz
FASTQUICK_TRACE=1 $R/fastquick_amd/bin/FASTQuick_amd align --index_prefix $P --fastq_1 $F/trace_1.fq.gz --fastq_2 $F/trace_2.fq.gz --out_prefix $F/trace_out --sam_out --read_len 151 --t 32 2>&1 >/dev/null | grep -v "sequences are\|call done\|consumers done\|index load"
