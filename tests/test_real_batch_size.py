"""A run at the reference's REAL batch size (READ_BUFFER_SIZE = 262,144 pairs, src/BwtMapper.h:36): two full batches and a short one,
2,000 markers, a fifth of the pairs on target.  The goldens' batches hold at most 1,200 pairs; what only shows at this size -- three
insert-size inferences from tens of thousands of samples each (libbwa/bwape.c:49-117: the summation order of the moments is part of
the result), the last_ii chain across full batches (src/BwtMapper.cpp:780-781), the mate-name check at a full batch's boundary
(:2087-2092), a packed call that carries three reference batches at once -- is compared three ways:

    the REAL reference (oracle/_ref/fq_ref_driver)  ==  the oracle (oracle/fq_oracle.c)  ==  the product's kernel bodies and host
    pipeline (host-loop library, tests/emu) fed the whole stream as ONE packed call

on the SAM text (byte for byte) and the three insert-size estimates as hex doubles.  About three minutes here; CPU tier, build
container only (the reference cannot travel)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from fastquick_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
B = 262144


def isize_lines(stages_path):
    return [ob.normalise_stage_line(ln.rstrip("\n")) for ln in open(stages_path) if ln.startswith("I ")]


@pytest.mark.refbuild
@pytest.mark.skipif(not os.path.exists(ob.REF_DRIVER), reason="oracle/_ref is built where /root/reference exists")
def test_two_full_reference_batches_and_a_short_one(tmp_path):
    n = 2 * B + 40000
    ref = synth.make_reference(n_markers=2000, n_long=200, seed=501)
    pre = str(tmp_path / "ref.FASTQuick.fa")
    ref.write_fasta(pre)
    subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    synth.write_qc_inputs(pre, ref)      # (the driver runs StatCollector::AddAlignment on every pair before it prints it, as PairEndMapper does)
    rb = synth.make_reads(ref, n, on_target=0.2, seed=502, sub_rate=0.008, del_frac=0.03, ins_frac=0.02, n_rate=0.001, chimera_frac=0.02)
    fq = [str(tmp_path / ("reads_%d.fq" % (e + 1))) for e in range(2)]
    for e in range(2):
        synth.write_fastq_uniform(rb.seq[e], rb.qual[e], 150, fq[e], bgzf=False)
    # ---- the reference itself: three batches of its own size
    ob.run_reference(pre, fq[0], fq[1], str(tmp_path / "ref"), "--batch", B, "--genome_size", len(ref.genome))
    ref_sam = open(str(tmp_path / "ref.sam"), "rb").read()
    ref_ii = isize_lines(str(tmp_path / "ref.stages"))
    os.remove(str(tmp_path / "ref.stages"))
    assert len(ref_ii) == 3 and all("avg=bff0" not in x for x in ref_ii), "every batch must have inferred its insert sizes"
    assert len(ref_sam) > 50e6
    # ---- the oracle, batch by batch
    oa = ob.OracleAligner(pre)
    oa.set_threads(8)
    oa.align(rb.names, rb.seq, rb.qual, rb.lens, str(tmp_path / "orc.stages"), str(tmp_path / "orc.sam"), batch=B)
    oa.close()
    assert isize_lines(str(tmp_path / "orc.stages")) == ref_ii, "oracle: insert-size estimates of the three batches"
    os.remove(str(tmp_path / "orc.stages"))
    assert open(str(tmp_path / "orc.sam"), "rb").read() == ref_sam, "oracle: SAM text"
    os.remove(str(tmp_path / "orc.sam"))
    # ---- the product's kernel bodies and host pipeline: the whole stream as ONE packed call of three reference batches
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libfq_emu.so"])
    lib = api.load_library(os.path.join(EMU_DIR, "libfq_emu.so"))
    ix = api.Index(pre, lib=lib)
    al = api.Aligner(ix, api.default_opts(lib), max_pairs=n)
    hp = api.HostPacked(rb.seq, rb.qual, rb.lens, rb.names, lib=lib)
    res = al.align_packed(hp)
    assert res.n_sub == 3
    got_ii = ["I avg=%016x std=%016x ap=%016x low=%d high=%d hb=%d" % (np.float64(s.avg).view(np.uint64), np.float64(s.std).view(np.uint64), np.float64(s.ap_prior).view(np.uint64), s.low, s.high, s.high_bayesian)
              for s in (res.isize_sub[k] for k in range(3))]
    assert got_ii == ref_ii, "host-loop library: insert-size estimates of the three reference batches of one call"
    sam = ix.sam_header() + al.sam_text()
    al.close(); hp.free(); ix.close()
    assert sam == ref_sam, "host-loop library: SAM text of one packed call over three reference batches"
