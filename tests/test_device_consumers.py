"""The consumers on the device as an interface (fq_ctx_set_emit, fq_sam_device_last, fq_ctx_attach_qc, fq_ctx_attach_bam; csrc/fq_emit.h): what a call leaves where, what
the host-side consumers say to a batch whose arrays stayed in HBM, what happens between calls.  (That the outputs are the reference's bytes is pinned elsewhere: every sam_text()
of the suite formats on both sides, tests/test_qc_consumer.py and test_bam_writer.py run both sides on every golden, tests/test_large_digests.py holds the command line to the
reference's digests.)  CPU tier: the kernel bodies on the host-loop backend; GPU tier: the kernels."""
import os
import subprocess

import numpy as np
import pytest

import golden_util  # noqa: F401
import oracle_binding as ob
from fastquick_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
EMIT_DEVICE_ONLY = 2


def check_interface(lib, g, **dev):
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    B = g["batch"]
    ix = api.Index(g["prefix"], lib=lib, **dev)
    want = [l for l in open(g["sam"], "rb").read().splitlines(keepends=True) if not l.startswith(b"@")]
    # ---- FQ_EMIT_SAM | FQ_EMIT_DEVICE_ONLY: the text is there, the arrays are not
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"]), max_pairs=max(16, B), emit=api.EMIT_SAM | EMIT_DEVICE_ONLY)
    got = b""
    for lo in range(0, seq.shape[1], B):
        hi = min(seq.shape[1], lo + B)
        res = al.align(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi])
        assert res.n_pairs == hi - lo and res.n_survivors > 0
        assert not res.rec and not res.cigar and not res.md and not res.multi, "the result arrays were to stay on the device"
        assert lib.fq_sam_format_last(al.h, None, 0) == -1, "the host formatter has nothing to format from"
        text = al.sam_text_device()
        assert len(text) == lib.fq_sam_device_bytes(al.h)
        assert al.sam_text_device() == text, "the text stays until the next call"
        got += text
    assert got == b"".join(want)
    # a host-side QC consumer refuses such a batch, and says why
    qc = api.QC(ix, g["prefix"], os.path.join(g["dir"], "iface_qc"), genome_size=g["genome_size"], read_len=g["qc_read_len"])
    qc.begin_file(g["fq1"], g["fq2"])
    with pytest.raises(api.FastquickError, match="left on the device"):
        qc.add(al)
    qc.close()
    al.close()
    # ---- without FQ_EMIT_SAM there is no device text to fetch
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"]), max_pairs=max(16, B), emit=0)
    al.align(seq[:, :B], qual[:, :B], lens[:, :B], names[:B])
    with pytest.raises(api.FastquickError):
        al.sam_text_device()
    assert lib.fq_ctx_set_emit(al.h, 1 << 10) == -1, "unknown flags are refused"
    # ---- an empty batch: empty text, a consumer that is attached takes it
    assert lib.fq_ctx_set_emit(al.h, api.EMIT_SAM) == 0
    qc = api.QC(ix, g["prefix"], os.path.join(g["dir"], "iface_qc2"), genome_size=g["genome_size"], read_len=g["qc_read_len"])
    qc.attach(al)
    qc.begin_file(g["fq1"], g["fq2"])
    al.align(seq[:, :0], qual[:, :0], lens[:, :0], names[:0])
    assert al.sam_text_device() == b""
    qc.add(al)
    # ... and a batch of pairs that all fail the filter (random bases)
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(3).integers(0, 4, (2, 64, seq.shape[2]))]
    al.align(rnd, qual[:, :64], lens[:, :64], names[:64])
    assert al.sam_text_device() == b""
    qc.add(al)
    qc.end_file()
    qc.write()
    qc.close(); al.close(); ix.close()


def test_interface_on_the_host_loop_backend(golden_cases):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so"])
    check_interface(api.load_library(os.path.join(emu, "libfq_emu.so")), golden_cases["basic"])


@pytest.mark.gpu
def test_interface_on_the_gpu(golden_cases):
    check_interface(api.load_library(), golden_cases["basic"], device=0)


def test_sixteen_byte_pieces_equal_the_byte_statements():
    """k_sam_body / k_bam_body write a piece that lies wholly inside SEQ or QUAL four bytes at a time (csrc/fq_emit.h: fq_swar_*): tests/emu/swar_check.cpp holds those forms to
    fq_sam_body_char / fq_bam_body_byte -- every byte value in every position, 60,000 random runs of every form piece by piece -- under ASan + UBSan."""
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "swar_check"])
    run = subprocess.run([os.path.join(emu, "swar_check")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0 and run.stdout.strip() == b"ok", run.stderr.decode(errors="replace")[-1500:]
