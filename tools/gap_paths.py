#!/usr/bin/env python3
"""Path statistics of the lane search kernel on the on-target mix (instrumented build: `make -C fastquick_amd/csrc instr`,
FQ_LIB_EXPERIMENT=fastquick_amd/libfastquick_amd_instr.so).  Prints how often each path of a loop iteration ("trip") is
executed by a wavefront and for how many lanes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fastquick_amd import api, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
tune = dict(kv.split("=") for kv in sys.argv[2].split(",")) if len(sys.argv) > 2 else {}
wd = os.environ.get("FQ_BENCH_DIR", "/tmp/fq_bench")
os.makedirs(wd, exist_ok=True)
markers = int(os.environ.get("FQ_MARKERS", "10000"))     # a smaller reference: Occ tables that fit the L2 with room to spare
pre = os.path.join(wd, "m%d.FASTQuick.fa" % markers)
ref = synth.make_reference(n_markers=markers, n_long=markers // 10, seed=12345)
if not os.path.exists(pre + ".rsa"):
    ref.write_fasta(pre)
    api.build_index(pre)
rb = synth.make_reads(ref, pairs, on_target=1.0, seed=3000)
ix = api.Index(pre, device=0)
al = api.Aligner(ix, max_pairs=pairs, tuning={k: int(v) for k, v in tune.items()})
al.upload(rb.seq, rb.qual, rb.lens, None)
al.align_resident()
al.align_resident()     # (the second call consolidates the pinned staging arena: not a steady-state call either)
al.reset_stats()
t0 = time.perf_counter()
al.align_resident()
dt = time.perf_counter() - t0
s = al.stats()
d = s["dbg"]
reads = s["reads_searched"]
print("pairs %d reads %d call %.1f ms gap kernel %.2f ms  pops/read %.1f pushes/read %.1f" % (pairs, reads, 1e3 * dt, s["kernel_ms"][7] + s["kernel_ms"][8], s["stack_pops"] / reads, s["stack_pushes"] / reads))
print("search stage on the device (first launch begin to last launch end, per call): %.2f ms" % (s["kernel_ms"][2] / max(1, 1)))
print("wave trips %d (%.1f per 64 reads)  lane trips %d  active lanes / trip %.1f" % (s["wave_trips"], s["wave_trips"] / (reads / 64.0), s["lane_trips"], s["lane_trips"] / max(1, s["wave_trips"])))
if d[0] and not os.environ.get("FQ_INSTR_TIMING"):
    print("instrumented trips %d, mean active %.1f" % (d[0], d[1] / d[0]))
    for name, k in (("0", 2), ("1-8", 3), ("9-16", 4), ("17-32", 5), ("33-48", 6), ("49-64", 7)):
        print("  trips with %5s active lanes: %5.1f %%" % (name, 100.0 * d[k] / d[0]))
    for name, k in (("pop", 8), ("tail", 9), ("expand", 10)):
        print("  %-6s path executed in %5.1f %% of trips, %5.1f lanes when executed (%.1f lane-trips per read)" % (name, 100.0 * d[k] / d[0], d[k + 3] / max(1, d[k]), d[k + 3] / reads))
    print("  all three paths in %.1f %% of trips; hit collection in %.1f %%" % (100.0 * d[14] / d[0], 100.0 * d[15] / d[0]))
if os.environ.get("FQ_INSTR_TIMING"):     # library built with `make instr INSTR_MODE=2`: dbg[2..5] are timings, not the histogram
    print("TIMING: step %.1f Mcycles/wave-sum, collect_hits %.1f (x16 shader clocks), hits with shadow sweep %d in %d trips; trips %d, hit trips %d" % (d[2] / 1e6, d[3] / 1e6, d[4], d[5], d[0], d[15]))
    print("TIMING: trips with a popping lane: %d, %.0f clocks each; trips without: %d, %.0f clocks each" % (d[7], 16.0 * d[6] / max(1, d[7]), d[0] - d[7], 16.0 * (d[2] - d[6]) / max(1, d[0] - d[7])))
    print("TIMING: no pop, no record fetch: %d trips, %.0f clocks each; no pop, record fetch: %d, %.0f; pop and record fetch: %d, %.0f" % (d[9], 16.0 * d[8] / max(1, d[9]), d[11], 16.0 * d[10] / max(1, d[11]), d[13], 16.0 * d[12] / max(1, d[13])))
    print("TIMING: wavefront lifetimes summed %.1f Mcycles x16 (step + collect = %.1f)" % (d[14] / 1e6, (d[2] + d[3]) / 1e6))
