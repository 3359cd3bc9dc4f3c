"""The reference-side binding, compiled and run (missing #4): `oracle/_ref/fq_ref_driver via_lib` loads the library through
include/fastquick_amd.h, copies every record it returns into a bwa_seq_t with the adapter INTEGRATION.md spells out (fill_bwa_seq) and
feeds the REFERENCE'S OWN consumers -- StatCollector::AddAlignment (src/StatCollector.h:151), BwtMapper::SetSamRecord
(src/BwtMapper.cpp:977) and bwa_print_sam1 (libbwa/bwase.c:455), in PairEndMapper's order (src/BwtMapper.cpp:2047-2085).  What they
write must be the committed goldens: SAM text, the SamRecord field dump and header, the 13 QC files -- "StatCollector and the
downstream steps consume it unchanged", shown instead of argued.
CPU tier: the host-loop build of the library (tests/emu); GPU tier: the HIP product, where oracle/_ref travelled with the snapshot."""
import os
import subprocess

import pytest

import golden_util
import oracle_binding as ob
from test_qc_consumer import QC_FILES, qc_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_LIB = os.path.join(HERE, "emu", "libfq_emu.so")
HIP_LIB = os.path.join(ROOT, "fastquick_amd", "libfastquick_amd.so")
needs_ref = pytest.mark.skipif(not os.path.exists(ob.REF_DRIVER), reason="oracle/_ref is built where /root/reference exists")


def via_lib(lib, g, out, *extra, fq2=None):
    args = ["--batch", g["batch"], "--genome_size", g["genome_size"]] + (["--q", g["trim_qual"]] if g["trim_qual"] else []) + \
           ["--read_len", g["qc_read_len"]]
    cmd = [ob.REF_DRIVER, "via_lib", lib, g["prefix"], g["fq1"], fq2 or g["fq2"], out] + [str(a) for a in args + list(extra)]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode == 0, run.stderr.decode(errors="replace")[-3000:]


def check_case(lib, g, tmp, se=False):
    stem = "ref_se" if se else "ref"
    out = os.path.join(str(tmp), "via")
    se_args = ("--se", 1) if se else ()
    via_lib(lib, g, out, *se_args)
    want_sam = g["se_sam"] if se else g["sam"]
    assert open(out + ".sam", "rb").read() == open(want_sam, "rb").read(), "SAM text printed by the reference's bwa_print_sam1"
    bad = [f for f in QC_FILES if qc_bytes(out + "." + f) != qc_bytes(os.path.join(g["dir"], stem + ".qc." + f))]
    assert not bad, "QC files written by the reference's StatCollector differ: %s" % bad
    if os.path.exists(os.path.join(g["dir"], stem + ".bamtxt")):
        outb = os.path.join(str(tmp), "via_bam")
        via_lib(lib, g, outb, "--bam_dump", 1, "--fai", os.path.join(g["dir"], "genome.fai"), *se_args)
        assert open(outb + ".bamhdr").read() == open(os.path.join(g["dir"], stem + ".bamhdr")).read()
        assert open(outb + ".bamtxt").read() == open(os.path.join(g["dir"], stem + ".bamtxt")).read(), "SamRecords filled by the reference's SetSamRecord"


@needs_ref
@pytest.mark.refbuild
@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_reference_consumers_take_the_librarys_records(tag, golden_cases, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])
    check_case(EMU_LIB, golden_cases[tag], tmp_path)


@needs_ref
@pytest.mark.refbuild
@pytest.mark.parametrize("tag", ["edge", "trim76", "qc"])
def test_reference_consumers_take_single_end_records(tag, golden_cases, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])
    g = golden_cases[tag]
    if "se_sam" not in g:
        pytest.skip("no single-end golden for this case")
    check_case(EMU_LIB, g, tmp_path, se=True)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["basic", "qc", "edge", "trim76"])
def test_reference_consumers_take_the_hip_librarys_records(tag, golden_cases, tmp_path):
    check_case(HIP_LIB, golden_cases[tag], tmp_path)
