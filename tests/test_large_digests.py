"""The HIP path and the command line against the REAL reference at shapes far above the goldens' (VERDICT r5 #5): tests/golden/large_digests.json
holds the SHA-256 of the SAM text and of each of the 13 QC files the reference itself (oracle/_ref/fq_ref_driver) wrote for three seeded inputs
at its own batch size -- tests/golden/make_large_digests.py, run in the build container.  The inputs are regenerated here from the same seeds:

  real_batches  2 x 262,144 + 40,000 pairs, 2,000 markers: three reference batches in one packed call
  cfg3_trim     262,144 pairs of the real-shaped WGS mix against 100,000 markers, --q 15
  ont76         262,144 pairs of 2x76 indel-rich on-target reads

Every shape goes (1) through the library (packed boundary, StatCollector on the device) and (2) through the command line from BGZF FASTQ files
(device front end, consumers in kernels)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_large_digests as mld  # noqa: E402

from fastquick_amd import api, synth  # noqa: E402

DIGESTS = os.path.join(HERE, "golden", "large_digests.json")
pytestmark = pytest.mark.gpu


def _digests():
    if not os.path.exists(DIGESTS):
        pytest.skip("tests/golden/large_digests.json is not there")
    return json.load(open(DIGESTS))


def _check_qc(out, want, label, fix_names=False):
    bad = []
    for f in mld.QC_FILES:
        data = open(out + "." + f, "rb").read()
        if f == "vcf":
            data = b"\n".join(ln for ln in data.split(b"\n") if not ln.startswith(b"##fileDate="))
        if fix_names and f == "FASTQ.csv":
            data = data.replace(b".fq.gz", b".fq")
        if hashlib.sha256(data).hexdigest() != want["qc"][f]["sha256"]:
            bad.append("%s (%d bytes, the reference's has %d)" % (f, len(data), want["qc"][f]["bytes"]))
    assert not bad, "%s: QC files differ from the reference's: %s" % (label, bad)


@pytest.mark.parametrize("key", ["real_batches", "cfg3_trim", "ont76"])
def test_library_and_command_line_give_the_references_digests(key, tmp_path):
    want = _digests().get(key)
    if want is None:
        pytest.skip("no digest for %s" % key)
    lib = api.load_library()
    ref, rb, o = mld.shape_inputs(key, synth, np)
    n, L = rb.seq.shape[1], o["read_len"]
    assert n == want["pairs"] and L == want["read_len"]
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    api.build_index(pre)
    synth.write_qc_inputs(pre, ref)
    # ---- (1) the library: the whole stream as one packed call, StatCollector on the device
    ix = api.Index(pre, device=0)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=o["trim_qual"]), max_pairs=n)
    out = str(tmp_path / "lib")
    qc = api.QC(ix, pre, out, genome_size=want["genome_size"], read_len=want["qc_read_len"])
    qc.attach(al)
    qc.begin_file("reads_1.fq", "reads_2.fq")
    hp = api.HostPacked(rb.seq, rb.qual, rb.lens, rb.names)
    res = al.align_packed(hp)
    assert res.n_sub == (n + mld.B - 1) // mld.B
    sam = ix.sam_header() + al.sam_text()          # (host formatter and device formatter: api.Aligner holds them to each other)
    qc.add(al)
    qc.end_file()
    qc.write()
    qc.close(); al.close(); hp.free(); ix.close()
    assert len(sam) == want["sam_bytes"] and hashlib.sha256(sam).hexdigest() == want["sam_sha256"], "library: SAM text differs from the reference's (%d bytes, %d there)" % (len(sam), want["sam_bytes"])
    del sam
    _check_qc(out, want, "library")
    # ---- (2) the command line from BGZF files: device front end, two contexts in turn, consumers in kernels
    fq = [str(tmp_path / ("reads_%d.fq.gz" % (e + 1))) for e in range(2)]
    for e in range(2):
        synth.write_fastq_uniform(rb.seq[e], rb.qual[e], L, fq[e], threads=8)
    synth.write_param(pre, ref, sum(1 for nm in ref.names if nm.endswith("|L")))
    with open(pre + ".genome.fa.fai", "w") as fh:
        fh.write("1\t%d\t3\t60\t61\n" % len(ref.genome))
    exe = os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd")
    outc = str(tmp_path / "cli")
    cmd = [exe, "align", "--index_prefix", pre[:-len(".FASTQuick.fa")], "--fastq_1", fq[0], "--fastq_2", fq[1], "--out_prefix", outc, "--sam_out", "--read_len", "151"] + \
          (["--q", str(o["trim_qual"])] if o["trim_qual"] else [])
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-2000:]
    assert b"front end on the device" in run.stderr
    assert len(run.stdout) == want["sam_bytes"] and hashlib.sha256(run.stdout).hexdigest() == want["sam_sha256"], "command line: SAM text differs from the reference's"
    _check_qc(outc, want, "command line", fix_names=True)
