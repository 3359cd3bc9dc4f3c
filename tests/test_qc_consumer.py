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


def qc_case(g, lib, device=None, packed=False, tuning=None, se=False):
    """Runs case g through the pipeline + QC consumer; returns {file: (got, want)} for the files that differ.
    se: the single-end mapper on the first FASTQ alone (AddAlignment(p, 0)), against the ref_se.qc.* goldens."""
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=lib) if device is None else api.Index(g["prefix"], device=device, lib=lib)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"], single_end=1 if se else 0), max_pairs=max(16, g["batch"]), tuning=tuning or {})
    out = os.path.join(g["dir"], "got_qc_se" if se else "got_qc")
    qc = api.QC(ix, g["prefix"], out, genome_size=g["genome_size"], read_len=g["qc_read_len"])
    qc.begin_file(g["fq1"], g["fq1"] if se else g["fq2"])     # FileStatCollector(fq1) names the one file twice
    if se:
        api.align_stream(al, list(names), seq[:1], qual[:1], lens[:1], g["batch"], None, None, qc=qc)
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


@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_qc_files_match_reference(tag, golden_cases, emu_lib):
    bad = qc_case(golden_cases[tag], emu_lib)
    assert not bad, explain(bad)


SE_CONSUMER_TAGS = [t for t in golden_util.se_case_tags() if os.path.exists(os.path.join(golden_util.GOLD, t, "ref_se.qc.Summary.gz"))]


@pytest.mark.parametrize("tag", SE_CONSUMER_TAGS)
def test_single_end_qc_files_match_reference(tag, golden_cases, emu_lib):
    bad = qc_case(golden_cases[tag], emu_lib, se=True)
    assert not bad, explain(bad)
