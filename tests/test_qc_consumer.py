"""The QC consumer (fq_qc_*: StatCollector's side of the boundary) against the QC files the REAL reference wrote for every golden
case (tests/golden/<case>/ref.qc.*, made by oracle/_ref/fq_ref_driver with the reference's own StatCollector in the loop).
CPU tier: the host pipeline on the host-loop backend feeds the consumer; the GPU tier runs the same comparison through the HIP
library (test_gpu_parity.py::test_qc_files_match_reference_golden)."""
import os
import subprocess

import pytest

import golden_util
import oracle_binding as ob
from fastquick_amd import api

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
QC_FILES = ["InsertSizeTable", "DepthDist", "GCDist", "EmpRepDist", "EmpCycleDist", "RawInsertSizeDist", "SexChromInfo", "Pileup",
            "FASTQ.csv", "Sequence.csv", "Summary", "AdjustedInsertSizeDist", "vcf"]


def qc_bytes(path):
    """A QC file's content; the genotype .vcf without the line that carries the day it was written."""
    data = open(path, "rb").read()
    if path.endswith(".vcf"):
        data = b"\n".join(ln for ln in data.split(b"\n") if not ln.startswith(b"##fileDate="))
    return data


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return api.load_library(os.path.join(EMU_DIR, "libfq_emu.so"))


def qc_case(g, lib, device=None, packed=False, tuning=None, se=False, on_device=False):
    """Runs case g through the pipeline + QC consumer; returns {file: (got, want)} for the files that differ.
    se: the single-end mapper on the first FASTQ alone (AddAlignment(p, 0)), against the ref_se.qc.* goldens.
    on_device: StatCollector's part of every call runs in the kernels of fq_emit.h (fq_ctx_attach_qc); the consumer's host side only appends."""
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=lib) if device is None else api.Index(g["prefix"], device=device, lib=lib)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"], single_end=1 if se else 0), max_pairs=max(16, g["batch"]), tuning=tuning or {})
    out = os.path.join(g["dir"], "got_qc_se" if se else "got_qc")
    qc = api.QC(ix, g["prefix"], out, genome_size=g["genome_size"], read_len=g["qc_read_len"])
    if on_device:
        qc.attach(al)
    qc.begin_file(g["fq1"], g["fq1"] if se else g["fq2"])     # FileStatCollector(fq1) names the one file twice
    if se:
        api.align_stream(al, list(names), seq[:1], qual[:1], lens[:1], g["batch"], None, None, qc=qc, packed=packed)
    else:
        api.align_stream(al, names, seq, qual, lens, g["batch"], None, None, qc=qc, packed=packed)
    qc.end_file()
    qc.write()
    qc.close(); al.close(); ix.close()
    bad = {}
    for f in QC_FILES:
        got = qc_bytes(out + "." + f)
        want = qc_bytes(os.path.join(g["dir"], ("ref_se.qc." if se else "ref.qc.") + f))
        if got != want:
            bad[f] = (got, want)
    return bad


def explain(bad):
    msg = []
    for f, (got, want) in bad.items():
        gl, wl = got.split(b"\n"), want.split(b"\n")
        first = next((i for i, (a, b) in enumerate(zip(gl, wl)) if a != b), min(len(gl), len(wl)))
        msg.append("%s: %d vs %d lines, first difference at line %d:\n  got  %r\n  want %r" % (
            f, len(gl), len(wl), first + 1, gl[first][:200] if first < len(gl) else None, wl[first][:200] if first < len(wl) else None))
    return "\n".join(msg)


SIDES = pytest.mark.parametrize("on_device", [False, True], ids=["host_consumer", "device_consumer"])


@SIDES
@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_qc_files_match_reference(tag, on_device, golden_cases, emu_lib):
    bad = qc_case(golden_cases[tag], emu_lib, on_device=on_device)
    assert not bad, explain(bad)


@pytest.mark.parametrize("tag", ["qc", "trim76", "repeat"])
def test_qc_files_from_packed_batches_on_the_device(tag, golden_cases, emu_lib):
    """the packed boundary: the surviving reads' qualities and names are uploaded for the consumers' kernels"""
    if tag not in golden_cases:
        pytest.skip("no such golden case")
    bad = qc_case(golden_cases[tag], emu_lib, packed=True, on_device=True)
    assert not bad, explain(bad)


SE_CONSUMER_TAGS = [t for t in golden_util.se_case_tags() if os.path.exists(os.path.join(golden_util.GOLD, t, "ref_se.qc.Summary.gz"))]


@SIDES
@pytest.mark.parametrize("packed", [False, True], ids=["ascii", "packed"])
@pytest.mark.parametrize("tag", SE_CONSUMER_TAGS)
def test_single_end_qc_files_match_reference(tag, packed, on_device, golden_cases, emu_lib):
    bad = qc_case(golden_cases[tag], emu_lib, se=True, packed=packed, on_device=on_device)
    assert not bad, explain(bad)


def test_a_consumer_counts_on_one_side_only(golden_cases, emu_lib):
    """a consumer that counted a call on the device refuses a batch of a context it is not attached to (the duplicate set and the sums live on
    one side), and the other way round"""
    g = golden_cases["basic"]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=emu_lib)
    kw = dict(genome_size=g["genome_size"], read_len=g["qc_read_len"])
    a1 = api.Aligner(ix, api.default_opts(emu_lib), max_pairs=max(16, g["batch"]))
    a2 = api.Aligner(ix, api.default_opts(emu_lib), max_pairs=max(16, g["batch"]))
    B = g["batch"]
    dev, host = a1, a2
    host.align(seq[:, :B], qual[:, :B], lens[:, :B], names[:B])
    qc = api.QC(ix, g["prefix"], os.path.join(g["dir"], "one_side_dev"), **kw)
    qc.begin_file(g["fq1"], g["fq2"])
    qc.attach(dev)
    dev.align(seq[:, :B], qual[:, :B], lens[:, :B], names[:B])
    qc.add(dev)
    with pytest.raises(api.FastquickError, match="counts on the device"):
        qc.add(host)
    qc.close()
    qc = api.QC(ix, g["prefix"], os.path.join(g["dir"], "one_side_host"), **kw)
    qc.begin_file(g["fq1"], g["fq2"])
    qc.add(host)
    qc.attach(dev)
    with pytest.raises(api.FastquickError, match="one side only"):
        dev.align(seq[:, :B], qual[:, :B], lens[:, :B], names[:B])
    emu_lib.fq_ctx_attach_qc(dev.h, None)
    qc.close()
    a1.close(); a2.close(); ix.close()


def _diff(out, g, prefix):
    bad = {}
    for f in QC_FILES:
        got, want = qc_bytes(out + "." + f), qc_bytes(os.path.join(g["dir"], prefix + f))
        if got != want:
            bad[f] = (got, want)
    return bad


@SIDES
@pytest.mark.parametrize("tag", ["qc", "basic", "edge"])
def test_segments_of_one_stream_merge_to_the_references_files(tag, on_device, golden_cases, emu_lib):
    """ONE FASTQ pair whose reference batches go to two shard consumers in turn (what two ranks of a sharded stream hold): every
    batch's segment is exported, and the segments merged in batch order into a third consumer give the reference's 13 files."""
    g = golden_cases[tag]
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=emu_lib)
    al = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"]), max_pairs=max(16, g["batch"]))
    kw = dict(genome_size=g["genome_size"], read_len=g["qc_read_len"])
    shards = [api.QC(ix, g["prefix"], os.path.join(g["dir"], "shard%d" % r), **kw) for r in range(2)]
    for q in shards:
        q.begin_file(g["fq1"], g["fq2"])
        q.state_reset()
    root = api.QC(ix, g["prefix"], os.path.join(g["dir"], "merged"), **kw)
    root.begin_file(g["fq1"], g["fq2"])
    n, B = seq.shape[1], g["batch"]
    for b, lo in enumerate(range(0, n, B)):
        hi = min(n, lo + B)
        q = shards[b % 2]
        if on_device:
            q.attach(al)
        al.align(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi])
        q.add(al)
        root.merge(q.state_export())
        q.state_reset()
    root.end_file()
    root.write()
    bad = _diff(os.path.join(g["dir"], "merged"), g, "ref.qc.")
    for q in shards + [root]:
        q.close()
    al.close(); ix.close()
    assert not bad, explain(bad)


@SIDES
def test_two_fastq_pairs_merge_to_the_references_files(on_device, golden_cases, emu_lib):
    """The two lines of a --fq_list on two shard consumers (one FASTQ pair each, closed there); their states merged in list order
    give what the reference writes for the two-pair run (tests/golden/qc/ref_fqlist.*)."""
    g = golden_cases["qc"]
    halves = golden_util.split_halves(g, g["dir"])
    ix = api.Index(g["prefix"], lib=emu_lib)
    kw = dict(genome_size=g["genome_size"], read_len=g["qc_read_len"])
    root = api.QC(ix, g["prefix"], os.path.join(g["dir"], "merged_fl"), **kw)
    sam = b""
    for r, (f1, f2) in enumerate(halves):
        names, seq, qual, lens = ob.read_fastq_pair(f1, f2)
        al = api.Aligner(ix, api.default_opts(emu_lib, trim_qual=g["trim_qual"]), max_pairs=max(16, g["batch"]))
        q = api.QC(ix, g["prefix"], os.path.join(g["dir"], "fl_shard%d" % r), **kw)
        q.state_reset()
        q.begin_file(f1, f2)
        if on_device:
            q.attach(al)
        for lo in range(0, seq.shape[1], g["batch"]):
            hi = min(seq.shape[1], lo + g["batch"])
            al.align(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi])
            q.add(al)
            sam += al.sam_text()
        q.end_file()
        root.merge(q.state_export())
        q.close(); al.close()
    root.write()
    bad = _diff(os.path.join(g["dir"], "merged_fl"), g, "ref_fqlist.qc.")
    root.close(); ix.close()
    assert not bad, explain(bad)
    want = b"".join(l for l in open(os.path.join(g["dir"], "ref_fqlist.sam"), "rb").read().splitlines(keepends=True) if not l.startswith(b"@"))
    assert sam == want


def test_estimator_cells_added_at_once_equal_the_loop(emu_lib):
    """The estimator's cells start at 1e-6 and take 1.0 per table line (src/InsertSizeEstimator.cpp:43-143); the device path hands over counts, and fq_qc_write adds them
    binade by binade (fq_qc.cpp: add_ones) instead of one by one: the same roundings, held to the loop here."""
    import ctypes as C
    import numpy as np
    lib = emu_lib
    lib.fq_qc_add_ones.restype = C.c_double
    lib.fq_qc_add_ones.argtypes = [C.c_double, C.c_uint64]
    rng = np.random.default_rng(77)
    starts = [1e-6, 0.0, 0.3, 0.9999999999, 1.0, 1.000001, 1.5, 2.0 ** 20 - 0.5, 2.0 ** 30 + 0.25, 2.0 ** 52 - 3.5, 2.0 ** 52 - 1.0, 2.0 ** 53 - 2.0] + list(rng.random(6) * 1e-5)
    counts = [0, 1, 2, 3, 7, 8, 9, 1023, 1024, 1025, 65535, 65537, 1000003] + [int(v) for v in rng.integers(1, 3000000, 8)]
    for x0 in starts:
        for t in counts:
            x = np.float64(x0)
            one = np.float64(1.0)
            if t <= 70000:
                for _ in range(t):
                    x = x + one
            else:      # (the loop itself, in blocks the float64 cumsum runs sequentially)
                left = t
                while left:
                    n = min(left, 1 << 20)
                    x = np.cumsum(np.concatenate(([x], np.ones(n))))[-1]
                    left -= n
            got = lib.fq_qc_add_ones(float(x0), t)
            assert got == float(x), (x0, t, got, float(x))
