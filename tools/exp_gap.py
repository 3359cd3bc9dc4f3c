#!/usr/bin/env python3
"""Search-stage experiments on the on-target mix: one resident batch, several tuning variants one after the other; per variant the
durations of the two rounds of the search stage (kernel begin/end timestamps) and a digest of the hit-derived results, which must
not depend on the variant.  Usage: exp_gap.py PAIRS "k=v,k=v" "k=v" ...   ("-" = defaults)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fastquick_amd import api, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
variants = sys.argv[2:] or ["-"]
wd = os.environ.get("FQ_BENCH_DIR", "/tmp/fq_bench")
os.makedirs(wd, exist_ok=True)
markers = int(os.environ.get("FQ_MARKERS", "10000"))
pre = os.path.join(wd, "m%d.FASTQuick.fa" % markers)
ref = synth.make_reference(n_markers=markers, n_long=markers // 10, seed=12345)
if not os.path.exists(pre + ".rsa"):
    ref.write_fasta(pre)
    api.build_index(pre)
t0 = time.perf_counter()
rb = synth.make_reads(ref, pairs, on_target=1.0, seed=3000)
print("reads made in %.1f s" % (time.perf_counter() - t0), flush=True)
ix = api.Index(pre, device=0)
for var in variants:
    tune = {} if var == "-" else {k: int(v) for k, v in (kv.split("=") for kv in var.split(","))}
    al = api.Aligner(ix, max_pairs=pairs, tuning=tune)
    al.upload(rb.seq, rb.qual, rb.lens, None)
    al.align_resident()
    al.align_resident()
    al.reset_stats()
    t0 = time.perf_counter()
    res = al.align_resident()
    dt = time.perf_counter() - t0
    s = al.stats()
    reads = s["reads_searched"]
    km, kl = s["kernel_ms"], s["kernel_launches"]
    import ctypes
    nrec = 2 * res.n_survivors
    dig = hashlib.sha1(ctypes.string_at(res.rec, nrec * ctypes.sizeof(api.Result))).hexdigest()[:12]
    print("%-40s call %.1f ms | nogap %.2f ms (%d) full %.2f ms (%d) stage %.2f | width %.2f | pops/read %.1f gap touches %d (nogap %d) | maxtrips %d wave trips %d lane trips %d | retries %d | %s"
          % (var, 1e3 * dt, km[8], kl[8], km[7], kl[7], km[2], km[1], s["stack_pops"] / max(1, reads), s["gap_occ_touches"], s["gap_nogap_touches"],
             s["max_wave_trips"], s["wave_trips"], s["lane_trips"], s["tier_retries"], dig), flush=True)
    d = s["dbg"]
    if d[0] and os.environ.get("FQ_INSTR_RAW"):
        print("   INSTR raw", list(d[:16]), flush=True)
    elif d[0]:
        print("   INSTR trips %d  step clocks/trip %.0f | pop trips %d: %.0f clk | occ-only %d: %.0f | rec-fetch %d: %.0f | sparse(<=2 lanes) %d: %.0f clk | lifetime sum %.0f Mclk"
              % (d[0], 16.0 * d[2] / max(1, d[0]), d[7], 16.0 * d[6] / max(1, d[7]), d[9], 16.0 * d[8] / max(1, d[9]), d[11], 16.0 * d[10] / max(1, d[11]), d[13], 16.0 * d[12] / max(1, d[13]), 16.0 * d[14] / 1e6), flush=True)
    al.close()
ix.close()
