"""Host pipeline + kernel bodies, run through the host-loop backend (tests/emu, test infrastructure), must reproduce
the REAL reference's golden stage dumps and SAM text.  This is the CPU-tier check of the host logic; the GPU tier
(test_gpu_parity.py) runs the same comparisons through the HIP library."""
import filecmp
import os
import subprocess

import pytest

import golden_util
import oracle_binding as ob
from fastquick_amd import api

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return api.load_library(os.path.join(EMU_DIR, "libfq_emu.so"))


@pytest.mark.parametrize("mode", ["lanes", "wave", "threads", "packed", "packed_bulk", "nogap", "generic_opts", "pipeline"])
@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_emulated_pipeline_matches_reference_golden(tag, mode, golden_cases, emu_lib):
    if mode not in ("lanes", "wave") and tag not in ("basic", "repeat", "edge", "qc", "trim76", "isize", "wide"):
        pytest.skip("this mode runs on seven of the cases here (CPU tier budget); the GPU tier runs every mode on every case")
    tuning = {}
    if mode == "wave":   # every search handed to the wavefront-per-read path (a one-lane wavefront here: its sequential rounds)
        tuning = {"gap_long_pops": 1, "gap_long_always": 1}
    if mode == "threads":   # the per-pair host phases split over threads even for these small inputs
        tuning = {"host_par_min": 1}
    if mode == "packed_bulk":   # packed input with the whole body uploaded and gathered on the device (many survivors)
        tuning = {"packed_bulk_min": 0, "md_mask_min": 0}      # (and the rows compared with the reference piece-wise before MD, as large calls do)
    if mode == "nogap":         # every launch begins with the round that searches without gap children (device-filling launches do)
        tuning = {"gap_nogap_min": 0}
    if mode == "generic_opts":  # ... with the search kernels that read the options from the launch (the default ones are compiled for FASTQuick's option block)
        tuning = {"gap_nogap_min": 0, "gap_generic_opts": 1}
    if mode == "pipeline":      # ... in segments, each segment's second round issued beside the next segment's first (large calls do)
        tuning = {"gap_nogap_min": 0, "gap_pipeline_min": 0, "gap_pipeline_segs": 3}
    if mode == "packed":        # ... with the survivors' rows gathered on the host (few survivors)
        tuning = {"packed_bulk_min": 1 << 30, "md_mask_min": 0}
    g = golden_cases[tag]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"], host_threads=3 if mode == "threads" else 0), max_pairs=max(16, g["batch"]), debug=True, tuning=tuning)
    st, sam = os.path.join(g["dir"], "emu.stages"), os.path.join(g["dir"], "emu.sam")
    api.align_stream(al, names, seq, qual, lens, g["batch"], st, sam, packed=mode.startswith("packed"))
    stats = al.stats()
    al.close()
    ix.close()
    diffs = [d for d in ob.diff_stage_files(g["stages"], st)]
    assert not diffs, "\n".join(diffs)
    assert filecmp.cmp(g["sam"], sam, shallow=False)
    if tag in ("basic", "repeat", "qc"):
        assert stats["pairs_on_device"] > 0, "the pairing kernel body (fq_pair_thread) was not exercised"
    if tag == "wide":      # SA intervals of >= 1000 rows: paired on the host through the (k,l) cache (libbwa/bwape.h:105, src/BwtMapper.cpp:815-843)
        assert stats["host_pairs"] > 300, "the pairs of the 1,292-fold repeat must have gone through the position cache"


@pytest.mark.parametrize("mode,tuning", [("lanes", {}), ("nogap", {"gap_nogap_min": 0}), ("wave", {"gap_long_pops": 1, "gap_long_always": 1}), ("handover", {"gap_long_pops": 8})])
def test_work_counters_match_the_oracle(mode, tuning, emu_lib, tmp_path):
    """What bench.py prices the search kernels with (48 B x gap_occ_touches) is the number of blocks bwt_match_gap reads in the
    reference, whatever the scheduling; pops are the reference's too (the round without gap children: at most one fewer per read)."""
    from fastquick_amd import synth
    ref = synth.make_reference(n_markers=60, n_long=6, seed=32, repeat_every=2, tandem_every=7)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    rb = synth.make_reads(ref, 600, on_target=0.95, seed=42, sub_rate=0.03, del_frac=0.08, ins_frac=0.07, n_rate=0.004, indel_len_max=3, chimera_frac=0.06)
    oa = ob.OracleAligner(pre)
    oa.align(rb.names, rb.seq, rb.qual, rb.lens, None, None, batch=200)
    oc = oa.counters()
    oa.close()
    ix = api.Index(pre, lib=emu_lib)
    al = api.Aligner(ix, max_pairs=200, tuning=tuning)
    api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, 200, None, None)
    gs = al.stats()
    al.close(); ix.close()
    assert gs["gap_occ_touches"] == oc["occ_gap_touches"] > 0
    assert gs["filter_probes"] == oc["filter_probes"]
    if mode.startswith("nogap"):
        assert oc["stack_pops"] - gs["reads_searched"] <= gs["stack_pops"] <= oc["stack_pops"]
    else:
        assert gs["stack_pops"] == oc["stack_pops"]


def test_a_rejected_ragged_packed_batch_leaves_nothing_behind(golden_cases, emu_lib):
    """A ragged packed batch with a read below the supported length is refused (FQ_ELIMIT); the batches after it on the same context
    align as if it had never been seen (the device counters of the refused call are cleared), and a packed batch object that is
    packed again is a new batch to the context (its serial changes)."""
    g = golden_cases["trim76"]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    n = 64
    seq, qual, lens, names = seq[:, :n].copy(), qual[:, :n].copy(), lens[:, :n].copy(), list(names[:n])
    lens[0, 5] -= 3      # ragged
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"]), max_pairs=n)
    good = api.HostPacked(seq, qual, lens, names, lib=emu_lib)
    res = al.align_packed(good)
    sam0, bases0 = al.sam_text(), res.n_bases
    bad_lens = lens.copy()
    bad_lens[1, 7] = 10
    bad = api.HostPacked(seq, qual, bad_lens, names, lib=emu_lib)
    with pytest.raises(api.FastquickError) as e:
        al.align_packed(bad)
    assert "read length outside" in str(e.value)
    al2 = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"]), max_pairs=n)     # a fresh stream for comparison
    for _ in range(2):
        al.prefetch(good)
        good.repack()                                     # same object, packed again: the prefetched copy must not be taken for it
        res = al.align_packed(good)
        assert res.n_bases == bases0
        res2 = al2.align_packed(good)
        assert res2.n_bases == bases0
    bad.free(); good.free()
    al.close(); al2.close(); ix.close()
    assert len(sam0) > 0


def test_option_limits_are_rejected(golden_cases, emu_lib):
    import ctypes as C
    ix = api.Index(golden_cases["basic"]["prefix"], lib=emu_lib)
    for kw in (dict(max_gapo=4), dict(max_gape=16), dict(seed_len=65), dict(s_mm=0), dict(fnr=-1.0, max_diff=31)):
        h = C.c_void_p()
        assert emu_lib.fq_ctx_create(ix.h, C.byref(api.default_opts(emu_lib, **kw)), 16, C.byref(h)) == -1, kw
    ix.close()


def test_read_length_limit_is_loud(golden_cases, emu_lib):
    import numpy as np
    ix = api.Index(golden_cases["basic"]["prefix"], lib=emu_lib)
    al = api.Aligner(ix, max_pairs=4)
    seq = np.full((2, 2, 40), ord("A"), dtype=np.uint8)
    lens = np.array([[40, 10], [40, 40]], dtype=np.int32)      # 10 < FQ_LMIN (15)
    with pytest.raises(api.FastquickError):
        al.align(seq, seq, lens, [b"a", b"b"])
    with pytest.raises(api.FastquickError):                    # batch larger than the context was sized for
        al.align(np.zeros((2, 8, 40), np.uint8), np.zeros((2, 8, 40), np.uint8), np.full((2, 8), 40, np.int32), [b"x"] * 8)
    al.close()
    ix.close()


def test_empty_batch(golden_cases, emu_lib):
    import numpy as np
    ix = api.Index(golden_cases["basic"]["prefix"], lib=emu_lib)
    al = api.Aligner(ix, max_pairs=4)
    res = al.align(np.zeros((2, 0, 150), np.uint8), np.zeros((2, 0, 150), np.uint8), np.zeros((2, 0), np.int32), [])
    assert res.n_pairs == 0 and res.n_survivors == 0
    al.close()
    ix.close()


OPTION_VARIANTS = [
    ("entry_limit", dict(max_entries=600)),
    ("nonstop", dict(mode=1 | 2 | 0x10, max_top2=0x7fffffff)),
    ("fixed_maxdiff", dict(fnr=-1.0, max_diff=4, max_gapo=2, max_gape=3, mode=2)),
    ("shared_buckets", dict(s_mm=4, s_gapo=4, s_gape=4)),
    ("loggap_multi", dict(mode=1 | 2 | 4, n_multi=8, N_multi=20, max_occ=50, is_sw=0)),
    ("il13_trim", dict(mode=1 | 2 | 0x200, trim_qual=15)),      # Phred+64 input (`--I`) with quality trimming
]


@pytest.mark.parametrize("name,okw", OPTION_VARIANTS, ids=[v[0] for v in OPTION_VARIANTS])
def test_emulated_pipeline_matches_oracle_with_option_variants(name, okw, emu_lib, tmp_path):
    """Non-default options: the host pipeline + kernel bodies against the oracle run with the same options."""
    from fastquick_amd import synth
    ref = synth.make_reference(n_markers=60, n_long=6, seed=35, repeat_every=2, tandem_every=7)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre, lib=emu_lib)
    rb = synth.make_reads(ref, 700, on_target=0.95, seed=45, sub_rate=0.03, del_frac=0.08, ins_frac=0.07, n_rate=0.004, indel_len_max=3, chimera_frac=0.06, qual_decay=name == "il13_trim")
    if okw.get("mode", 0) & 0x200:
        rb.qual[rb.qual > 0] += 31
    ix = api.Index(pre, lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib, **okw), max_pairs=400, debug=True)
    api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, 400, str(tmp_path / "emu.stages"), str(tmp_path / "emu.sam"))
    oa = ob.OracleAligner(pre, ob.default_opts(**okw))
    oa.align(rb.names, rb.seq, rb.qual, rb.lens, str(tmp_path / "orc.stages"), str(tmp_path / "orc.sam"), batch=400)
    diffs = [d for d in ob.diff_stage_files(str(tmp_path / "orc.stages"), str(tmp_path / "emu.stages"))]
    assert not diffs, "\n".join(diffs[:20])
    assert filecmp.cmp(str(tmp_path / "orc.sam"), str(tmp_path / "emu.sam"), shallow=False)
    al.close(); ix.close(); oa.close()


def trimmed_max_len_case(lib, tmp_path, device=None):
    """infer_isize's max_len is the longest TRIMMED read of the reference batch, filtered reads included (bwape.c:60-61).  Short
    fragments make the estimate's lower bound equal to it (p25 - 2 IQR < max_len), so it shows in the result.  Every on-target
    read is trimmed (decaying qualities, --q 15); a few off-target (filtered) reads keep their full length.  The packed
    boundary must find that length (a) from the reads' last quality bytes and (b) without them, by uploading the qualities."""
    import ctypes as C
    import numpy as np
    from fastquick_amd import synth
    ref = synth.make_reference(n_markers=60, n_long=6, seed=81)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre, lib=lib)
    n, L = 3000, 100
    rb = synth.make_reads(ref, n, read_len=L, on_target=0.1, seed=82, qual_decay=True, frag_mean=115, frag_sd=10)
    ix = api.Index(pre, lib=lib) if device is None else api.Index(pre, device=device, lib=lib)
    opts = lambda: api.default_opts(lib, trim_qual=15, batch_pairs=1000)
    al = api.Aligner(ix, opts(), max_pairs=n)
    res = al.align(rb.seq, rb.qual, rb.lens, rb.names)
    surv = set(np.ctypeslib.as_array(res.pair_idx, shape=(res.n_survivors,)).tolist())
    trimmed_max = max(res.rec[i].clip_len for i in range(2 * res.n_survivors))
    assert trimmed_max < L and res.n_survivors > 150
    off = [p for p in range(n) if p not in surv]
    for p in (off[3], off[len(off) // 2], off[-2]):       # one whole read in each reference batch of 1000, all in filtered pairs
        rb.qual[0, p, :] = ord("I")
    assert off[3] < 1000 <= off[len(off) // 2] < 2000 <= off[-2]
    res = al.align(rb.seq, rb.qual, rb.lens, rb.names)
    want = [(res.isize_sub[k].low, res.isize_sub[k].avg, res.isize_sub[k].std) for k in range(res.n_sub)]
    want_sam = al.sam_text()
    al.close()
    assert all(w[0] == L for w in want), want          # the filtered whole read sets the bound, not the survivors' maximum
    for drop_qlast in (False, True):
        al = api.Aligner(ix, opts(), max_pairs=n)
        hp = api.HostPacked(rb.seq, rb.qual, rb.lens, rb.names, lib=lib)
        if drop_qlast:
            hp.p.contents.qual_last = None
        res = al.align_packed(hp)
        got = [(res.isize_sub[k].low, res.isize_sub[k].avg, res.isize_sub[k].std) for k in range(res.n_sub)]
        st = al.stats()
        assert got == want and al.sam_text() == want_sam, (drop_qlast, got, want)
        full_quals = 2 * n * rb.qual.shape[2]
        assert (st["h2d_bytes"] > full_quals) == drop_qlast      # the quality rows of every read only travel when they must
        al.close()
        hp.free()
    ix.close()


def test_trimmed_max_len_of_filtered_reads(emu_lib, tmp_path):
    trimmed_max_len_case(emu_lib, tmp_path)


@pytest.mark.parametrize("packed", [False, True], ids=["ascii", "packed"])
@pytest.mark.parametrize("tag", golden_util.se_case_tags())
def test_emulated_single_end_matches_reference_golden(tag, packed, golden_cases, emu_lib):
    """BwtMapper::SingleEndMapper (fq_opts_t::single_end): the first FASTQ of the case alone, against the reference's single-end run;
    ASCII batches and the packed boundary (fq_pack_single_reads_into -> fq_align_packed)."""
    g = golden_cases[tag]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"], single_end=1), max_pairs=max(16, g["batch"]), debug=True)
    st, sam = os.path.join(g["dir"], "emu_se.stages"), os.path.join(g["dir"], "emu_se.sam")
    api.align_stream(al, list(names), seq[:1], qual[:1], lens[:1], g["batch"], st, sam, packed=packed)
    al.close()
    ix.close()
    diffs = [d for d in ob.diff_stage_files(g["se_stages"], st)]
    assert not diffs, "\n".join(diffs)
    assert filecmp.cmp(g["se_sam"], sam, shallow=False)


def test_single_end_edge_batches(golden_cases, emu_lib):
    """Single-end contexts: an empty batch; a batch whose reads are all filtered (random sequence); the packed entry refuses."""
    import numpy as np
    g = golden_cases["basic"]
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib, single_end=1), max_pairs=64)
    res = al.align(np.zeros((1, 0, 160), np.uint8), np.zeros((1, 0, 160), np.uint8), np.zeros((1, 0), np.int32), [])
    assert res.n_pairs == 0 and res.n_survivors == 0
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (1, 40, 160))].copy()
    qual = np.full((1, 40, 160), ord("I"), np.uint8)
    lens = np.full((1, 40), 150, np.int32)
    seq[:, :, 150:] = 0; qual[:, :, 150:] = 0
    res = al.align(seq, qual, lens, [b"r%d" % i for i in range(40)])
    assert res.n_pairs == 40 and res.n_survivors == 0 and res.n_both_filtered == 40
    assert al.sam_text() == b""
    names, s2, q2, l2 = ob.read_fastq_pair(g["fq1"], g["fq2"])
    with pytest.raises(api.FastquickError):
        hp = api.HostPacked(s2[:, :16], q2[:, :16], l2[:, :16], names[:16], lib=emu_lib)
        try:
            al.align_packed(hp)
        finally:
            hp.free()
    al.close()
    ix.close()


def test_a_hook_that_raises_stops_the_call_and_marks_the_stream_broken(golden_cases, emu_lib):
    """an exception in the `before` hook (a receive that timed out) used to be swallowed until the C call returned -- which
    had meanwhile computed from the stale state and exported it as good.  The guard now tells the library (fq_ctx_mark_stream_broken):
    the call stops after the hook, and the state handed on carries the broken mark."""
    g = golden_cases["basic"]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib), max_pairs=max(16, g["batch"]))
    exported = []

    def before():
        raise TimeoutError("recv timed out")

    def after():
        exported.append(al.export_state())

    al.set_serial_hooks(before, after)
    n = min(seq.shape[1], g["batch"])
    with pytest.raises(api.FastquickError, match="serial hook raised"):
        al.align(seq[:, :n], qual[:, :n], lens[:, :n], names[:n])
    assert len(exported) == 1 and exported[0][:6] == b"_FQSTX", "the state handed on must carry the broken mark"
    other = api.Aligner(ix, api.default_opts(emu_lib), max_pairs=max(16, g["batch"]))
    with pytest.raises(api.FastquickError):
        other.import_state(exported[0])
    other.close()
    al.close()
    ix.close()


def test_a_malformed_stream_state_is_refused_whole_and_breaks_the_stream(golden_cases, emu_lib):
    """a (k,l) entry that claims 2^62 positions made `4 * m` wrap, passed the bound check and threw through the C entry; and a
    token refused half-way had already overwritten the drand48 state and the cache.  Now the token is parsed into temporaries: refused
    tokens change nothing but the broken mark (the state the stream should have continued from never arrived)."""
    import struct
    g = golden_cases["basic"]
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib), max_pairs=16)
    good = al.export_state()
    head = _stream_state(12345)[:-8]
    for bad in (head + struct.pack("<Q", 1) + struct.pack("<QQ", 7, 1 << 62),              # 4 * m wraps to 0
                head + struct.pack("<Q", 1) + struct.pack("<QQ", 7, 3) + b"\x00" * 8,       # three positions promised, two present
                head + struct.pack("<Q", 1 << 60),                                          # more entries than bytes
                head + struct.pack("<Q", 2) + struct.pack("<QQI", 7, 1, 9)):                # second entry missing
        with pytest.raises(api.FastquickError):
            al.import_state(bad)
        rest = al.export_state()
        assert rest[:6] == b"_FQSTX" and rest[8:] == good[8:], "a refused token must leave the state untouched, marked broken"
    al.close()
    ix.close()


def test_two_contexts_in_turn_with_the_state_moved_equal_one(golden_cases, emu_lib):
    """fq_ctx_state_move (what the command line's two contexts hand to each other): the reference batches of the `wide` golden -- a 1,292-fold repeat through the (k,l)
    cache -- and of `isize` (the last_ii chain) aligned on two contexts in turn, the state moved before every call, give the golden's text; the context a state was
    moved from is left with a fresh stream's cache (its export is as long as one without entries)."""
    for tag in ("wide", "isize"):
        g = golden_cases[tag]
        names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
        B = g["batch"]
        ix = api.Index(g["prefix"], lib=emu_lib)
        ctx = [api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"]), max_pairs=max(16, B)) for _ in range(2)]
        empty = len(ctx[0].export_state())
        got = b""
        for k, lo in enumerate(range(0, seq.shape[1], B)):
            hi = min(seq.shape[1], lo + B)
            cur, other = ctx[k & 1], ctx[(k & 1) ^ 1]
            if k:
                cur.take_state_of(other)
                assert len(other.export_state()) == empty
            cur.align(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi])
            got += cur.sam_text()
        want = b"".join(l for l in open(g["sam"], "rb").read().splitlines(keepends=True) if not l.startswith(b"@"))
        assert got == want, tag
        with pytest.raises(api.FastquickError):
            ctx[0].take_state_of(ctx[0])
        for a in ctx:
            a.close()
        ix.close()


def _stream_state(rng: int) -> bytes:
    """fq_ctx_state_export's layout for a fresh stream whose drand48 state is `rng`: mark, rng, last_ii (avg = std = -1, rest 0), no (k,l) entries."""
    import struct
    return b"_FQST1\x00\x00" + struct.pack("<Q", rng) + struct.pack("<dddIIII", -1.0, -1.0, 0.0, 0, 0, 0, 0) + struct.pack("<Q", 0)


@pytest.mark.parametrize("steps_before_zero", [1, 2, 3, 7, 40, 65, 100])
def test_the_drand48_stream_through_its_state_zero(steps_before_zero, golden_cases, emu_lib, tmp_path):
    zero_state_case(steps_before_zero, golden_cases["basic"], emu_lib, tmp_path)


def zero_state_case(steps_before_zero, g, lib, tmp_path, device=None):
    """bwa_aln2seq_core takes a read's only best hit "unless the draw is exactly 0" (libbwa/bwase.c:29-41): then it draws once, not twice,
    and every read behind it sees a stream shifted by one.  That happens once in 2^48 draws -- unless the stream is put there: the state
    that is k steps before the state 0 is imported (fq_ctx_state_import / the oracle's set_rng), for k odd and even, inside the first chunk
    of 16 pairs and behind it.  Where the 0 is a read's second draw (the row inside its hit's interval: row k) everything is defined and
    must equal the oracle; where it is the first draw, see below.  The host's replay (which skips over chunks of one-hit reads with one multiply-add, but not across the
    state 0), the device's draws for the pairs before its own and the choice itself must all agree with the oracle."""
    a_inv = pow(0x5DEECE66D, -1, 1 << 48)
    x = 0
    for _ in range(steps_before_zero):
        x = (a_inv * (x - 0xB)) % (1 << 48)
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=lib) if device is None else api.Index(g["prefix"], device=device, lib=lib)
    al = api.Aligner(ix, api.default_opts(lib), max_pairs=seq.shape[1], debug=True)
    al.import_state(_stream_state(x))
    st, sam = str(tmp_path / "emu.stages"), str(tmp_path / "emu.sam")
    if steps_before_zero in (1, 3, 7, 65):
        # the 0 is the FIRST draw of a one-hit read (the case's first 33 reads all have one best hit, two draws each): no hit is taken, the
        # reference's record keeps its slot's previous SA row and it reads the reference wherever that leads -- undefined; refused, loudly
        with pytest.raises(api.FastquickError, match="drew exactly 0"):
            api.align_stream(al, names, seq, qual, lens, seq.shape[1], st, sam)
        al.close(); ix.close()
        return
    api.align_stream(al, names, seq, qual, lens, seq.shape[1], st, sam)
    al.close(); ix.close()
    oa = ob.OracleAligner(g["prefix"])
    oa.set_rng(x)
    oa.align(names, seq, qual, lens, str(tmp_path / "o.stages"), str(tmp_path / "o.sam"), batch=seq.shape[1])
    oa.close()
    diffs = [d for d in ob.diff_stage_files(str(tmp_path / "o.stages"), st)]
    assert not diffs, "\n".join(diffs)
    assert filecmp.cmp(str(tmp_path / "o.sam"), sam, shallow=False)


def test_streams_run_inside_the_library_equal_the_calls_made_one_by_one(golden_cases, emu_lib):
    streams_case(golden_cases, emu_lib)


def streams_case(golden_cases, emu_lib):
    """fq_stream_run: three streams (three contexts) walking their reference batches inside the library -- the records of every call, read in
    the per-call callback, are those of fq_align_packed called by hand in the same order (the stream state: last_ii chain, drand48, (k,l) cache)"""
    per_stream, want = [], []
    ix = None
    for tag in ("basic", "qc", "isize"):
        g = golden_cases[tag]
        names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
        if ix is None or ix.prefix != g["prefix"]:
            ix = api.Index(g["prefix"], lib=emu_lib)
        B, n = g["batch"], seq.shape[1]
        packs = [api.HostPacked(seq[:, a:a + B], qual[:, a:a + B], lens[:, a:a + B], names[a:a + B], lib=emu_lib) for a in range(0, n, B)][:3]
        al = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"]), max_pairs=max(16, B))
        sams = []
        for p in packs:
            al.align_packed(p)
            sams.append(al.sam_text())
        al.close()
        want.append(sams)
        per_stream.append((ix, g, packs))
    aligners = [api.Aligner(ix_, api.default_opts(emu_lib, trim_qual=g_["trim_qual"]), max_pairs=max(16, g_["batch"])) for ix_, g_, _ in per_stream]
    got = [[None] * len(p) for _, _, p in per_stream]

    def on_call(stream, call):
        if call < len(got[stream]):
            got[stream][call] = aligners[stream].sam_text()
    n_calls = min(len(p) for _, _, p in per_stream)
    surv = api.stream_run(aligners, [p for _, _, p in per_stream], n_calls, on_call=on_call)
    for s in range(3):
        assert got[s][:n_calls] == want[s][:n_calls], "stream %d" % s
        assert surv[s] > 0
    # a stream that fails (a batch larger than the context takes) reports its code and leaves the others' results standing
    small = api.Aligner(per_stream[0][0], api.default_opts(emu_lib), max_pairs=16)
    with pytest.raises(api.FastquickError):
        api.stream_run([small], [per_stream[0][2]], 1)
    small.close()
    for al in aligners:
        al.close()
    for _, _, packs in per_stream:
        for p in packs:
            p.free()
