#!/usr/bin/env python3
"""Scratch check: run a hard on-target batch with the wavefront-per-read tier forced on (FQ_GAP_LONG_POPS) and compare with the oracle."""
import os, sys, filecmp, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fastquick_amd import api, synth
import oracle_binding as ob
lib = api.load_library()
d = tempfile.mkdtemp()
pre = os.path.join(d, "ref.FASTQuick.fa")
ref = synth.make_reference(n_markers=300, n_long=40, seed=5, repeat_every=7, tandem_every=11)
ref.write_fasta(pre); api.build_index(pre)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
rb = synth.make_reads(ref, n, on_target=0.9, seed=3, sub_rate=0.03, del_frac=0.08, ins_frac=0.07, n_rate=0.003, chimera_frac=0.03)
ix = api.Index(pre, device=0)
al = api.Aligner(ix, api.default_opts(lib), max_pairs=4096, debug=True)
t = time.time(); api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, 4096, d + "/g.stages", d + "/g.sam"); print("gpu", time.time() - t)
oa = ob.OracleAligner(pre, ob.default_opts())
oa.align(rb.names, rb.seq, rb.qual, rb.lens, d + "/o.stages", d + "/o.sam", batch=4096)
df = [x for x in ob.diff_stage_files(d + "/o.stages", d + "/g.stages") if not x.startswith("line count")]
print("diffs", len(df), df[:3], "sam equal", filecmp.cmp(d + "/o.sam", d + "/g.sam", shallow=False))
st = al.stats()
print({k: st[k] for k in ("stack_pops", "gap_occ_touches", "tier_retries", "reads_searched", "max_pops_per_read", "max_wave_trips", "kernel_ms")})
print("oracle pops", oa.counters()["stack_pops"])
