"""GPU tier: the HIP library (through the C ABI) against (a) the REAL reference's golden vectors, (b) the oracle on
fresh seeded inputs, (c) size-independent properties at full batch size."""
import filecmp
import hashlib
import os

import numpy as np
import pytest

import golden_util
import oracle_binding as ob
from fastquick_amd import api, synth

pytestmark = pytest.mark.gpu
_ACGT_BYTES = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def lib():
    L = api.load_library()   # fastquick_amd/libfastquick_amd.so -- raises if missing: no fallback
    return L


# search_mode: "lanes" = the default tiering (one read per lane; the wavefront-per-read kernel only takes over long searches
# once the queue is dry), "wave1"/"wave64" = hand every search over to the wavefront-per-read kernel after 1 / 64 pops, so
# that its parallel rounds, commit rule and run bookkeeping are exercised by every read of the case
SEARCH_MODES = {"lanes": {}, "wave1": {"gap_long_pops": 1, "gap_long_always": 1},
                # wave64 also sends every mate-SW window above 300 bases down the one-task-per-lane kernel (the path of oversize windows)
                "wave64": {"gap_long_pops": 64, "gap_long_always": 1, "sw_wave_max": 300},
                # the packed-batch boundary (fq_pack_reads -> fq_packed_prefetch -> fq_align_packed): survivors' rows gathered on the
                # host / the whole body uploaded and gathered on the device
                "packed": {"packed_bulk_min": 1 << 30, "md_mask_min": 0}, "packed_bulk": {"packed_bulk_min": 0, "md_mask_min": 0},
                # ... with the filter kernel on the device's shared stream of the highest priority (a tuning key: DESIGN section 9, cfg 3)
                "packed_prio": {"packed_bulk_min": 1 << 30, "md_mask_min": 0, "prep_priority": 1},
                # every launch begins with the round that searches without gap children, as device-filling launches do
                "nogap": {"gap_nogap_min": 0},
                # ... with the kernels that read the options from the launch instead of the ones compiled for FASTQuick's own option block
                "generic_opts": {"gap_nogap_min": 0, "gap_generic_opts": 1},
                # the hand-over rule of small launches with a low threshold: a search still running after 8 pops once the work queue
                # is dry (both of its blocks) goes to the wavefront-per-read kernel
                "handover": {"gap_long_pops": 8},
                # the mate-rescue kernel's reverse pass as the serial statement on one lane (the default walks a row with the whole wavefront)
                "sw_serial": {"sw_serial_reverse": 1},
                # the first round in three segments of the queue, each segment's second round on the context's second stream beside
                # the next segment's first (what calls of >= 4 M searched reads do)
                "pipeline": {"gap_nogap_min": 0, "gap_pipeline_min": 0, "gap_pipeline_segs": 3}}


@pytest.fixture(params=list(SEARCH_MODES))
def search_mode(request):
    return request.param


@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_gpu_matches_reference_golden(tag, golden_cases, lib, search_mode):
    g = golden_cases[tag]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], device=0)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"]), max_pairs=max(16, g["batch"]), debug=True, tuning=SEARCH_MODES[search_mode])
    st, sam = os.path.join(g["dir"], "gpu.stages"), os.path.join(g["dir"], "gpu.sam")
    api.align_stream(al, names, seq, qual, lens, g["batch"], st, sam, packed=search_mode.startswith("packed"))
    stats = al.stats()
    al.close()
    ix.close()
    diffs = [d for d in ob.diff_stage_files(g["stages"], st)]
    assert not diffs, "\n".join(diffs)
    assert filecmp.cmp(g["sam"], sam, shallow=False)
    assert stats["kernel_launches"][2] > 0, "the gap-search kernel did not run on the device"
    if tag in ("basic", "repeat", "qc"):
        assert stats["pairs_on_device"] > 0, "k_pair did not run"
    if tag == "wide":      # SA intervals of >= 1000 rows: paired on the host through the (k,l) cache (libbwa/bwape.h:105, src/BwtMapper.cpp:815-843)
        assert stats["host_pairs"] > 300, "the pairs of the 1,292-fold repeat must have gone through the position cache"
    if search_mode.startswith("wave") or (search_mode == "handover" and tag != "cfg0_example"):   # (cfg0_example searches one pair: its reads stop before the 64-pop check)
        assert stats["tier_retries"] > 0, "the wavefront-per-read kernel was not exercised"


CASES = [
    ("easy150", dict(n_markers=200, n_long=20, seed=31), dict(on_target=0.6, seed=41), 6000, 2048, 0),
    ("hard150", dict(n_markers=120, n_long=12, seed=32, repeat_every=2, tandem_every=7),
     dict(on_target=0.95, seed=42, sub_rate=0.03, del_frac=0.08, ins_frac=0.07, n_rate=0.004, indel_len_max=3, chimera_frac=0.06), 4000, 1500, 0),
    ("trim100", dict(n_markers=100, n_long=10, seed=33),
     dict(read_len=100, on_target=0.9, seed=43, qual_decay=True, sub_rate=0.01, del_frac=0.04, ins_frac=0.03, chimera_frac=0.1, frag_mean=260), 3000, 3000, 15),
    ("exome76", dict(n_markers=150, n_long=0, seed=34, repeat_every=5),
     dict(read_len=76, on_target=1.0, seed=44, sub_rate=0.01, del_frac=0.05, ins_frac=0.05, indel_len_max=2, frag_mean=200, frag_sd=20), 4000, 1024, 0),
]


@pytest.mark.parametrize("tag,refkw,readkw,n,batch,q", CASES, ids=[c[0] for c in CASES])
def test_gpu_matches_oracle_on_fresh_inputs(tag, refkw, readkw, n, batch, q, lib, tmp_path, search_mode):
    ref = synth.make_reference(**refkw)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    rb = synth.make_reads(ref, n, **readkw)
    ix = api.Index(pre, device=0)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=q), max_pairs=batch, debug=True, tuning=SEARCH_MODES[search_mode])
    api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, batch, str(tmp_path / "gpu.stages"), str(tmp_path / "gpu.sam"), packed=search_mode.startswith("packed"))
    oa = ob.OracleAligner(pre, ob.default_opts(trim_qual=q))
    oa.align(rb.names, rb.seq, rb.qual, rb.lens, str(tmp_path / "orc.stages"), str(tmp_path / "orc.sam"), batch=batch)
    diffs = [d for d in ob.diff_stage_files(str(tmp_path / "orc.stages"), str(tmp_path / "gpu.stages"))]
    assert not diffs, "\n".join(diffs)
    assert filecmp.cmp(str(tmp_path / "orc.sam"), str(tmp_path / "gpu.sam"), shallow=False)
    # work counters agree with the oracle's instrumented counts (algorithmic-byte model)
    oc, gs = oa.counters(), al.stats()
    assert gs["filter_probes"] == oc["filter_probes"]
    assert 0 < gs["sa_rows"] <= oc["sa_calls"]          # the GPU walks each enumerated row once; the CPU path re-walks rows per use
    # the numerator of the search kernels' roofline (48 B x gap_occ_touches, bench.py): the blocks bwt_match_gap reads, however the
    # search was scheduled (rounds, tiers, hand-overs: a read counts once, in the launch that completed it)
    assert gs["gap_occ_touches"] == oc["occ_gap_touches"]
    if search_mode in ("nogap", "generic_opts", "pipeline"):
        # a read the round without gap children settles ends when its stack runs empty; the reference pops one more entry -- a gap
        # child that round never pushed -- and stops on it: at most one pop per searched read fewer
        assert oc["stack_pops"] - gs["reads_searched"] <= gs["stack_pops"] <= oc["stack_pops"]
    elif gs["tier_retries"] == 0:
        assert gs["stack_pops"] == oc["stack_pops"]
    else:
        # reads searched again by one wavefront per read: its parallel rounds count the pops of a round together (a few pops per
        # such read may be missing where a round ended early); the touches above are exact there too
        assert abs(gs["stack_pops"] - oc["stack_pops"]) <= gs["tier_retries"]
    al.close(); ix.close(); oa.close()


def test_full_batch_properties(lib, tmp_path):
    """262,144 pairs (the reference's READ_BUFFER_SIZE) of a WGS-like mix: determinism, filter/oracle agreement on a
    sample, counters, and invariants that hold at any size."""
    ref = synth.make_reference(n_markers=2000, n_long=200, seed=51)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    n = 262144
    rb = synth.make_reads(ref, n, on_target=0.01, seed=52)
    ix = api.Index(pre, device=0)
    digests = []
    for rep in range(2):
        al = api.Aligner(ix, max_pairs=n)
        res = al.align(rb.seq, rb.qual, rb.lens, rb.names)
        sam = al.sam_text()
        digests.append(hashlib.md5(sam).hexdigest())
        surv = np.ctypeslib.as_array(res.pair_idx, shape=(res.n_survivors,)).copy()
        assert res.n_pairs == n and res.n_both_filtered == n - res.n_survivors
        assert np.all(np.diff(surv) > 0), "survivors must come out in input order (consumer contract)"
        assert res.n_bases == int(rb.lens.sum())
        recs = [res.rec[i] for i in range(2 * res.n_survivors)]
        for r in recs[:2000]:
            assert r.len == 150 and r.full_len == 150
            if r.type != 0:
                assert r.pos + 1 <= ix.l_pac
        al.close()
    assert digests[0] == digests[1], "two fresh contexts must give identical output (srand48(11) per stream)"
    # the first 3,000 pairs alone (fresh stream) must equal the oracle on the same prefix, filter decisions included
    k = 3000
    al = api.Aligner(ix, max_pairs=k, debug=True)
    api.align_stream(al, rb.names[:k], rb.seq[:, :k], rb.qual[:, :k], rb.lens[:, :k], k, str(tmp_path / "g.st"), str(tmp_path / "g.sam"))
    oa = ob.OracleAligner(pre)
    oa.align(rb.names[:k], rb.seq[:, :k], rb.qual[:, :k], rb.lens[:, :k], str(tmp_path / "o.st"), str(tmp_path / "o.sam"), batch=k)
    assert not [d for d in ob.diff_stage_files(str(tmp_path / "o.st"), str(tmp_path / "g.st"))]
    assert filecmp.cmp(str(tmp_path / "o.sam"), str(tmp_path / "g.sam"), shallow=False)
    al.close(); oa.close(); ix.close()


def test_cli_sam_out_matches_reference_golden(golden_cases, tmp_path):
    """`FASTQuick_amd align --sam_out` (C++ front end: gz FASTQ tokenizer + C ABI) on the golden FASTQ files must print the
    reference's SAM text, also when the input spans several chunks (reader threads prefetch the next chunk while the device
    works on the current one)."""
    import gzip
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fastquick_amd", "bin", "FASTQuick_amd")
    assert os.path.exists(exe), "build() must produce the CLI"
    for tag, chunk_batches in (("repeat", 1), ("nref", 1), ("basic", 1), ("isize", 2), ("example151", 1)):   # example151: the
        # reference's own example reads -- mates named differently, lower-case bases, a shorter read, one short batch
        g = golden_cases[tag]   # several chunks per run where the golden batch is smaller than the input: the prefetching reader
        fq = []
        for k in ("fq1", "fq2"):      # exercise the gz path of the tokenizer
            gz = str(tmp_path / (tag + os.path.basename(g[k]) + ".gz"))
            with open(g[k], "rb") as fi, gzip.open(gz, "wb") as fo:
                fo.write(fi.read())
            fq.append(gz)
        prefix = g["prefix"][:-len(".FASTQuick.fa")]
        cmd = [exe, "align", "--index_prefix", prefix, "--fastq_1", fq[0], "--fastq_2", fq[1], "--out_prefix", str(tmp_path / tag), "--sam_out"]
        if g["trim_qual"]:
            cmd += ["--q", str(g["trim_qual"])]
        cmd += ["--batch_pairs", str(g["batch"]), "--chunk_pairs", str(chunk_batches * g["batch"])]
        run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert run.returncode == 0, "%s: %s" % (tag, run.stderr.decode(errors="replace")[-2000:])
        assert run.stdout == open(g["sam"], "rb").read(), tag
        if tag == "example151":   # with full batches the reference's mate-name check fires on the first pair (TestRead_1 / TestRead_2)
            run = subprocess.run(cmd[:-4] + ["--batch_pairs", "128", "--chunk_pairs", "128"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert run.returncode != 0 and b"same order" in run.stderr
    # without --sam_out: <out>.bam in genome coordinates and StatCollector's QC files, against the reference's
    from test_cli_consumers import cli_bam_and_qc, cli_frac_samp
    cli_bam_and_qc(exe, golden_cases["qc"], str(tmp_path / "cli_qc"))
    cli_frac_samp(exe, golden_cases["qc"], str(tmp_path / "cli_frac"))      # --frac_samp 0.8 against the reference's down-sampled run


# Non-default options, each against the oracle run with the same options (the oracle itself is pinned against the reference with
# the default options; these variants pin the GPU path to the oracle where the two could drift apart: the entry limit and the
# exact tier behind it, non-stop mode, a fixed max_diff with more gaps, scores that make child classes share a bucket, XA limits).
OPTION_VARIANTS = [
    ("entry_limit", dict(max_entries=600)),
    ("nonstop", dict(mode=1 | 2 | 0x10, max_top2=0x7fffffff)),
    ("fixed_maxdiff", dict(fnr=-1.0, max_diff=4, max_gapo=2, max_gape=3, mode=2)),
    ("shared_buckets", dict(s_mm=4, s_gapo=4, s_gape=4)),
    ("loggap_multi", dict(mode=1 | 2 | 4, n_multi=8, N_multi=20, max_occ=50, is_sw=0)),
    ("il13_trim", dict(mode=1 | 2 | 0x200, trim_qual=15)),      # Phred+64 input (`--I`) with quality trimming
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,okw", OPTION_VARIANTS, ids=[v[0] for v in OPTION_VARIANTS])
def test_gpu_matches_oracle_with_option_variants(name, okw, lib, tmp_path, search_mode):
    if search_mode in ("wave64", "packed_bulk", "generic_opts", "packed_prio") or (search_mode == "nogap" and name == "nonstop"):
        pytest.skip("covered by lanes, wave1 and packed")
    ref = synth.make_reference(n_markers=120, n_long=12, seed=35, repeat_every=2, tandem_every=7)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    rb = synth.make_reads(ref, 2500, on_target=0.95, seed=45, sub_rate=0.03, del_frac=0.08, ins_frac=0.07, n_rate=0.004, indel_len_max=3, chimera_frac=0.06, qual_decay=name == "il13_trim")
    if okw.get("mode", 0) & 0x200:
        rb.qual[rb.qual > 0] += 31
    ix = api.Index(pre, device=0)
    al = api.Aligner(ix, api.default_opts(lib, **okw), max_pairs=1000, debug=True, tuning=SEARCH_MODES[search_mode])
    api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, 1000, str(tmp_path / "gpu.stages"), str(tmp_path / "gpu.sam"), packed=search_mode.startswith("packed"))
    oa = ob.OracleAligner(pre, ob.default_opts(**okw))
    oa.align(rb.names, rb.seq, rb.qual, rb.lens, str(tmp_path / "orc.stages"), str(tmp_path / "orc.sam"), batch=1000)
    diffs = [d for d in ob.diff_stage_files(str(tmp_path / "orc.stages"), str(tmp_path / "gpu.stages"))]
    assert not diffs, "\n".join(diffs[:20])
    assert filecmp.cmp(str(tmp_path / "orc.sam"), str(tmp_path / "gpu.sam"), shallow=False)
    if name == "entry_limit":
        assert al.stats()["tier_retries"] > 0, "the entry limit was meant to push reads into the exact tier"
    al.close(); ix.close(); oa.close()


@pytest.mark.gpu
def test_large_ontarget_call_matches_oracle_prefix_and_chunked_run(lib, tmp_path):
    """BASELINE cfg 5 shape (2x76 bp, every pair on target, indel-rich) at a size that fills the device: one call of 49,152 pairs =
    12 reference batches of 4,096 (98 k searched reads: sorted queue, whole-wavefront refill, threaded host phases).
    (1) SAM text identical to the same stream fed in three calls of four batches (results do not depend on the chunking);
    (2) the first three reference batches equal the oracle's run over that prefix (the stream is sequential: a prefix of the
        input gives a prefix of the output), stage by stage."""
    B, n = 4096, 12 * 4096
    ref = synth.make_reference(n_markers=400, n_long=0, seed=61, repeat_every=5)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    rb = synth.make_reads(ref, n, read_len=76, on_target=1.0, seed=62, sub_rate=0.01, del_frac=0.05, ins_frac=0.05, indel_len_max=2, frag_mean=200, frag_sd=20)
    ix = api.Index(pre, device=0)
    opts = api.default_opts(lib, batch_pairs=B)
    # (gap_nogap_min lowered so that this launch begins like a device-filling one: a first round without gap children, the reads it
    # cannot settle -- here, with 10 % indel reads, a good part -- searched in full by the next)
    al = api.Aligner(ix, opts, max_pairs=n, debug=True, tuning={"gap_nogap_min": 1000})
    al.align(rb.seq, rb.qual, rb.lens, rb.names)
    sam_one, stages_one = al.sam_text(), al.stage_text()
    assert al.stats()["reads_searched"] > 65536
    assert al.stats()["tier_retries"] > 1000, "the round without gap children was meant to run"
    al.close()
    al = api.Aligner(ix, api.default_opts(lib, batch_pairs=B), max_pairs=4 * B)
    parts = []
    for b0 in range(0, n, 4 * B):
        al.align(rb.seq[:, b0:b0 + 4 * B], rb.qual[:, b0:b0 + 4 * B], rb.lens[:, b0:b0 + 4 * B], rb.names[b0:b0 + 4 * B])
        parts.append(al.sam_text())
    al.close()
    assert b"".join(parts) == sam_one
    k = 3 * B
    oa = ob.OracleAligner(pre)
    oa.align(rb.names[:k], rb.seq[:, :k], rb.qual[:, :k], rb.lens[:, :k], str(tmp_path / "o.st"), str(tmp_path / "o.sam"), batch=B)
    want = open(str(tmp_path / "o.st"), "rb").read()
    # the device's dump numbers the batches of the call 0..11; the oracle's 0..2: the first three sections must be identical
    cut = stages_one.find(b"B 3 ")
    assert cut > 0
    with open(str(tmp_path / "g.st"), "wb") as fh:
        fh.write(stages_one[:cut])
    assert not [d for d in ob.diff_stage_files(str(tmp_path / "o.st"), str(tmp_path / "g.st"))]
    body = open(str(tmp_path / "o.sam"), "rb").read()
    body = body[len(ix.sam_header()):] if body.startswith(ix.sam_header()) else body
    assert sam_one.startswith(body), "SAM text of the first three reference batches"
    oa.close(); ix.close()


def test_packed_boundary_properties(lib, tmp_path):
    """The measured boundary: packed batches in pinned host memory in, records in host memory out, the next batch's
    upload running under this one's kernels.  A WGS-like stream of four calls of 262,144 pairs with N bases, quality trimming
    (--q 15) and decaying qualities must give byte-identical SAM text through fq_align_batch (ASCII rows) and through
    fq_pack_reads / fq_packed_prefetch / fq_align_packed, with and without the prefetch; the H2D volume must be the 48 bytes of
    filter keys per pair plus the survivors' rows (and qualities), not the batch."""
    ref = synth.make_reference(n_markers=1500, n_long=150, seed=71)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    n, calls = 262144, 4
    rb = synth.make_reads(ref, n * calls, on_target=0.004, seed=72, n_rate=0.002, qual_decay=True, sub_rate=0.01)
    good = np.random.default_rng(73).random(rb.qual.shape[:2]) < 0.3      # three reads in ten keep good qualities to their end
    rb.qual[good] = ord("I")
    ix = api.Index(pre, device=0)
    opts = lambda: api.default_opts(lib, trim_qual=15)
    al = api.Aligner(ix, opts(), max_pairs=n)
    want = []
    for k in range(calls):
        sl = slice(k * n, (k + 1) * n)
        al.align(rb.seq[:, sl], rb.qual[:, sl], rb.lens[:, sl], rb.names[sl])
        want.append(al.sam_text())
    al.close()
    assert sum(len(w) for w in want) > 100000
    packs = [api.HostPacked(rb.seq[:, k * n:(k + 1) * n], rb.qual[:, k * n:(k + 1) * n], rb.lens[:, k * n:(k + 1) * n], rb.names[k * n:(k + 1) * n]) for k in range(calls)]
    for prefetch in (True, False):
        al = api.Aligner(ix, opts(), max_pairs=n)
        got = []
        for k in range(calls):
            if prefetch and k + 1 < calls:
                al.prefetch(packs[k + 1])
            res = al.align_packed(packs[k])
            got.append(al.sam_text())
            assert res.n_bases == int(rb.lens[:, k * n:(k + 1) * n].sum())
        st = al.stats()
        al.close()
        assert got == want, "packed boundary (prefetch=%s) differs from the ASCII boundary" % prefetch
        surv = st["reads_searched"]
        assert st["h2d_bytes"] < 50 * n * calls + 1200 * surv + (1 << 20), st["h2d_bytes"]   # 48 B of filter keys + 2 last-quality bytes per pair
        assert st["h2d_bytes"] >= 48 * n * calls
    for p in packs:
        p.free()
    ix.close()


def test_trimmed_max_len_of_filtered_reads_gpu(lib, tmp_path):
    from test_pipeline_emu import trimmed_max_len_case
    trimmed_max_len_case(lib, tmp_path, device=0)


@pytest.mark.parametrize("on_device", [False, True], ids=["host_consumer", "device_consumer"])
@pytest.mark.parametrize("packed", [False, True], ids=["ascii", "packed"])
@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_qc_files_match_reference_golden(tag, packed, on_device, golden_cases, lib):
    """StatCollector's files (.InsertSizeTable .Pileup .DepthDist .GCDist .EmpRepDist .EmpCycleDist .RawInsertSizeDist .SexChromInfo
    .FASTQ.csv .Sequence.csv .Summary) from the QC consumer fed by the HIP path, byte for byte the REAL reference's -- counted on the host from
    the result arrays, and counted by the kernels of fq_emit.h inside the calls (fq_ctx_attach_qc)."""
    from test_qc_consumer import qc_case, explain
    bad = qc_case(golden_cases[tag], lib, device=0, packed=packed, on_device=on_device)
    assert not bad, explain(bad)


@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_bam_records_match_reference_golden(tag, golden_cases, lib):
    """The BAM file written from the HIP path's records (genome coordinates, @SQ from the .fai, RG) decodes to the fields the REAL
    reference's SetSamRecord puts into its SamRecords."""
    from test_bam_writer import bam_case
    bam_case(golden_cases[tag], lib, device=0, packed=(tag in ("qc", "edge")))


def test_100k_marker_index_real_shaped_reads(lib, tmp_path):
    """BASELINE cfg 3 shape: a ~100k-marker reduced reference (l_pac 6.5e7: Occ tables ten times those of the 10k set, beyond every
    cache level but the Infinity Cache) and "real-shaped" reads (N bases, decaying qualities with --q 15 trimming, adapter
    tails, 5 % duplicates, reads over contig ends) in a WGS-like mix.  One reference batch of 262,144 pairs through the packed
    boundary; the first 12,288 pairs (a fresh stream) against the oracle stage by stage; invariants on the whole batch."""
    import time
    t0 = time.time()
    ref = synth.make_reference(n_markers=100000, n_long=10000, seed=91)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    n, k = 262144, 12288
    rb = synth.make_reads(ref, n, on_target=0.021, seed=92, n_rate=0.001, qual_decay=True, sub_rate=0.008, del_frac=0.02, ins_frac=0.01,
                          adapter_frac=0.01, dup_frac=0.05, edge_frac=0.02)
    good = np.random.default_rng(93).random(rb.qual.shape[:2]) < 0.5
    rb.qual[good] = ord("I")
    ix = api.Index(pre, device=0)
    assert ix.l_pac == 10000 * 2001 + 90000 * 501
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=15), max_pairs=n)
    hp = api.HostPacked(rb.seq, rb.qual, rb.lens, rb.names)
    res = al.align_packed(hp)
    sam_all = al.sam_text()
    surv = np.ctypeslib.as_array(res.pair_idx, shape=(res.n_survivors,)).copy()
    assert res.n_pairs == n and 0.015 * n < res.n_survivors < 0.08 * n      # on-target pairs plus chance passes of the filter (more bits set at 100k)
    assert np.all(np.diff(surv) > 0)
    st = al.stats()
    assert st["h2d_bytes"] < 60 * n + 2000 * st["reads_searched"]
    al.close(); hp.free()
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=15), max_pairs=k, debug=True)
    api.align_stream(al, rb.names[:k], rb.seq[:, :k], rb.qual[:, :k], rb.lens[:, :k], k, str(tmp_path / "g.st"), str(tmp_path / "g.sam"), packed=True)
    oa = ob.OracleAligner(pre, ob.default_opts(trim_qual=15))
    oa.align(rb.names[:k], rb.seq[:, :k], rb.qual[:, :k], rb.lens[:, :k], str(tmp_path / "o.st"), str(tmp_path / "o.sam"), batch=k)
    assert not [d for d in ob.diff_stage_files(str(tmp_path / "o.st"), str(tmp_path / "g.st"))]
    assert filecmp.cmp(str(tmp_path / "o.sam"), str(tmp_path / "g.sam"), shallow=False)
    oa.close()
    # ... and the WHOLE batch of 262,144 pairs against the oracle (not a 12,288-pair prefix only): SAM text byte for byte
    oa = ob.OracleAligner(pre, ob.default_opts(trim_qual=15))
    oa.set_threads(max(2, min(64, os.cpu_count() or 2)))
    oa.align(rb.names, rb.seq, rb.qual, rb.lens, None, str(tmp_path / "o_all.sam"), batch=n, header=False)
    want_all = open(str(tmp_path / "o_all.sam"), "rb").read()
    assert len(sam_all) > 100000
    assert sam_all == want_all, "the 262,144-pair batch differs from the oracle's"
    al.close(); oa.close(); ix.close()
    print("cfg3 test: %.1f s" % (time.time() - t0))


def test_bench_call_shape_matches_oracle_exactly(lib, tmp_path):
    """The shape bench.py times: ONE call of 16 reference batches of 262,144 pairs (4,194,304 pairs) of the WGS-like mix against the 10k-
    marker reference, through the packed boundary.  The oracle runs the same 4.2 M pairs as one stream of 16 batches (drand48 stream,
    last_ii chain and (k,l) cache carried from batch to batch, as PairEndMapper does); the SAM text of the call must be the oracle's
    byte for byte, and the work counters that price the kernels must be its counts."""
    ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    B, nb = 262144, 16
    n = B * nb
    on_frac = ref.l_pac / 3.1e9
    rng = np.random.default_rng(77)
    L = 150
    seq = _ACGT_BYTES[rng.integers(0, 4, (2, n, L), dtype=np.uint8)]
    n_on = int(rng.binomial(n, on_frac))
    on = synth.make_reads(ref, n_on, on_target=1.0, seed=78)
    slots = np.sort(rng.choice(n, size=n_on, replace=False))
    seq[:, slots] = on.seq
    qual = np.full((2, n, L), ord("I"), dtype=np.uint8)
    lens = np.full((2, n), L, dtype=np.int32)
    names = [b"r%09d" % i for i in range(n)]
    ix = api.Index(pre, device=0)
    al = api.Aligner(ix, max_pairs=n)
    hp = api.HostPacked(seq, qual, lens, names)
    res = al.align_packed(hp)
    assert res.n_sub == nb and res.n_pairs == n
    got = al.sam_text()
    gs = al.stats()
    al.close(); hp.free(); ix.close()
    oa = ob.OracleAligner(pre)
    oa.align(names, seq, qual, lens, None, str(tmp_path / "o.sam"), batch=B, header=False)
    oc = oa.counters()
    oa.close()
    want = open(str(tmp_path / "o.sam"), "rb").read()
    assert len(want) > 100000
    assert got == want, "the call's SAM text differs from the oracle's stream of 16 batches"
    assert gs["filter_probes"] == oc["filter_probes"] and gs["gap_occ_touches"] == oc["occ_gap_touches"]
    assert abs(gs["stack_pops"] - oc["stack_pops"]) <= gs["tier_retries"]      # (reads handed to the wavefront-per-read kernel: see the fresh-input test)


@pytest.mark.parametrize("shape", ["1m_150", "4m_150", "1m_76"])
def test_ontarget_call_matches_oracle_on_a_prefix_of_two_batches(shape, lib, tmp_path):
    """bench.py's on-target legs, at the shapes it times: one call of 4,194,304 on-target pairs (16 reference batches: the call the
    search-stage roofline is quoted on -- VERDICT r3: the test used to stop at 1,048,576), one of 1,048,576 (the throughput leg), and one
    of 1,048,576 pairs of 2x76 bp indel-rich reads (BASELINE cfg 4).  Each is a device-filling search launch with its round without gap
    children, the hard reads leading the second round, and records that never leave the device between the search and the result
    arrays.  A stream is causal, so the first two reference batches of the call's output must be what the oracle gives for those 524,288
    pairs alone (its search stage sliced over the host's cores as the reference's --t does)."""
    ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    B, k = 262144, 2 * 262144
    n = (1 << 22) if shape == "4m_150" else (1 << 20)
    kw = dict(read_len=76, frag_mean=200, frag_sd=20, del_frac=0.05, ins_frac=0.05, indel_len_max=2) if shape == "1m_76" else {}
    rb = synth.make_reads(ref, n, on_target=1.0, seed=3000, **kw)
    ix = api.Index(pre, device=0)
    al = api.Aligner(ix, max_pairs=n)
    hp = api.HostPacked(rb.seq, rb.qual, rb.lens, rb.names)
    res = al.align_packed(hp)
    assert res.n_sub == n // B
    got = al.sam_text()
    gs = al.stats()
    assert gs["kernel_launches"][8] >= 1, "the round without gap children did not run"
    assert gs["pairs_on_device"] > 0.9 * res.n_survivors * 0.5, "pairing ran on the host"
    al.close(); hp.free(); ix.close()
    oa = ob.OracleAligner(pre)
    oa.set_threads(max(2, min(64, os.cpu_count() or 2)))
    oa.align(rb.names[:k], rb.seq[:, :k], rb.qual[:, :k], rb.lens[:, :k], None, str(tmp_path / "o.sam"), batch=B, header=False)
    oa.close()
    want = open(str(tmp_path / "o.sam"), "rb").read()
    # records of the pairs of the first two batches: names are r<index>, two records per surviving pair, in input order.  (The 4.2 M-pair
    # call's text is 3 GB: the cut is found from the front.)
    at, cut = 0, len(got)
    while at < len(got):
        if int(got[at + 1:got.index(b"\t", at)]) >= k:
            cut = at
            break
        at = got.index(b"\n", at) + 1
        if at > len(want) + 4096:
            cut = at
            break
    assert len(want) > (4000000 if shape == "1m_76" else 10000000) and got[:cut] == want


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [False, True], ids=["ascii", "packed"])
@pytest.mark.parametrize("tag", golden_util.se_case_tags())
def test_gpu_single_end_matches_reference_golden(tag, packed, golden_cases, lib):
    """BwtMapper::SingleEndMapper (fq_opts_t::single_end) on the device: the first FASTQ of the case alone, as ASCII batches and
    through the packed boundary."""
    g = golden_cases[tag]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], device=0)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"], single_end=1), max_pairs=max(16, g["batch"]), debug=True)
    st, sam = os.path.join(g["dir"], "gpu_se.stages"), os.path.join(g["dir"], "gpu_se.sam")
    api.align_stream(al, list(names), seq[:1], qual[:1], lens[:1], g["batch"], st, sam, packed=packed)
    stats = al.stats()
    al.close()
    ix.close()
    diffs = [d for d in ob.diff_stage_files(g["se_stages"], st)]
    assert not diffs, "\n".join(diffs)
    assert filecmp.cmp(g["se_sam"], sam, shallow=False)
    assert stats["kernel_launches"][2] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["basic", "trim76", "edge"])
def test_gpu_single_end_consumers_match_reference(tag, golden_cases, lib):
    """The single-end mapper's StatCollector files (AddAlignment(p, 0)) and BAM records (SetSamRecord(p, 0)) from the device path."""
    from test_qc_consumer import qc_case, explain
    from test_bam_writer import bam_case
    bad = qc_case(golden_cases[tag], lib, device=0, se=True)
    assert not bad, explain(bad)
    bad = qc_case(golden_cases[tag], lib, device=0, se=True, on_device=True)
    assert not bad, explain(bad)
    bam_case(golden_cases[tag], lib, device=0, se=True)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["basic", "wide", "qc"])
def test_filter_tables_entered_by_the_device_are_the_references_tables(tag, golden_cases, lib, tmp_path, monkeypatch):
    """fq_index_load without .rollhash: k_bitmap_kmers over the reference in HBM (and the host's listing) == the reference's own tables"""
    from test_index_build import bitmap_case
    bitmap_case(tag, golden_cases, lib, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_streams_run_inside_the_library_on_the_gpu(golden_cases, lib):
    """fq_stream_run (what bench.py's timed region drives): three streams on three contexts of one device == the calls made one by one"""
    from test_pipeline_emu import streams_case
    streams_case(golden_cases, lib)



@pytest.mark.gpu
@pytest.mark.parametrize("steps_before_zero", [1, 3, 100])
def test_gpu_drand48_stream_through_its_state_zero(steps_before_zero, golden_cases, lib, tmp_path):
    """The once-in-2^48 arm of bwa_aln2seq_core ("taken unless the draw is exactly 0"), reached by importing a stream state that is a
    few steps before the generator's state 0: k_main_hit's draws and the host's replay against the oracle (see test_pipeline_emu.py)."""
    from test_pipeline_emu import zero_state_case
    zero_state_case(steps_before_zero, golden_cases["basic"], lib, tmp_path, device=0)
