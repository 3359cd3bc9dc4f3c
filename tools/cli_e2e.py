#!/usr/bin/env python3
"""End-to-end command-line run on synthetic on-target reads: FASTQ files -> BAM + QC files, with the wall time of the whole command
next to the library's own device / host figures (the consumers -- StatCollector restatement, BAM writer -- run on the host).
usage: tools/cli_e2e.py [pairs] [extra CLI args...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastquick_amd import api, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
wd = os.environ.get("FQ_BENCH_DIR", "/tmp/fq_e2e")
os.makedirs(wd, exist_ok=True)
pre = os.path.join(wd, "e2e")
fa = pre + ".FASTQuick.fa"
ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
if not os.path.exists(fa + ".rsa"):
    ref.write_fasta(fa)
    api.build_index(fa)
synth.write_qc_inputs(fa, ref)
synth.write_param(fa, ref, 1000)
with open(fa + ".genome.fa.fai", "w") as fh:
    fh.write("1\t%d\t3\t60\t61\n" % len(ref.genome))
rb = synth.make_reads(ref, pairs, on_target=1.0, seed=77)
f1, f2 = rb.write_fastq(os.path.join(wd, "reads"))
exe = os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd")
for extra in ([], ["--sam_out"]):
    cmd = [exe, "align", "--index_prefix", pre, "--fastq_1", f1, "--fastq_2", f2, "--out_prefix", os.path.join(wd, "out")] + extra + sys.argv[2:]
    t0 = time.perf_counter()
    run = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    note = [l for l in run.stderr.decode(errors="replace").splitlines() if "device time" in l or "consumers" in l or "index staged" in l]
    print("%-10s rc %d  %d pairs in %.2f s = %.0f pairs/s   %s" % (" ".join(extra) or "BAM", run.returncode, pairs, dt, pairs / dt, " | ".join(note)))
