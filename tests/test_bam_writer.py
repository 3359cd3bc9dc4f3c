"""The BAM consumer (fq_bam_*: BwtMapper::SetSamRecord / SetSamFileHeader behind an own BGZF layer) against what the REAL reference
puts into its SamRecords for every golden case (tests/golden/<case>/ref.bamtxt, ref.bamhdr: oracle/_ref/fq_ref_driver --bam_dump,
the fields read back through the reference's own SamRecord getters).  The file this repository writes is decoded here by an
independent BAM reader (gzip members + the record layout of the SAM specification) and compared field by field; tags as a set (the
reference's SamRecord iterates its tags in hash order)."""
import gzip
import os
import struct
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


def check_bgzf(path):
    """Every block is a gzip member with the BC extra field and its own size; the file ends with the 28-byte empty block."""
    raw = open(path, "rb").read()
    at, n = 0, 0
    while at < len(raw):
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 16] == b"BC\x02\x00", "not a BGZF block at %d" % at
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        at += bsize
        n += 1
    assert at == len(raw) and n >= 2
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), "BGZF end-of-file block missing"


def decode_bam(path, sort_tags=True):
    data = gzip.open(path, "rb").read()      # concatenated gzip members
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", data, at)[0]
    at += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, at)[0]
        refs.append((data[at + 4:at + 4 + l_name - 1].decode(), struct.unpack_from("<i", data, at + 4 + l_name)[0]))
        at += 8 + l_name
    recs = []
    while at < len(data):
        bs = struct.unpack_from("<i", data, at)[0]
        rid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mrid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", data, at + 4)
        p = at + 36
        name = data[p:p + l_name - 1].decode()
        p += l_name
        cig = "".join("%d%s" % (v >> 4, "MIDNSHP=X"[v & 15]) for v in struct.unpack_from("<%dI" % n_cig, data, p)) or "*"
        p += 4 * n_cig
        seq = "".join("=ACMGRSVTWYHKDBN"[(data[p + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = "".join(chr(q + 33) for q in data[p:p + l_seq])
        p += l_seq
        tags = []
        end = at + 4 + bs
        while p < end:
            tag, ty = data[p:p + 2].decode(), chr(data[p + 2])
            p += 3
            if ty == "Z":
                z = data.index(b"\0", p)
                tags.append("%s:Z:%s" % (tag, data[p:z].decode()))
                p = z + 1
            elif ty == "A":
                tags.append("%s:A:%s" % (tag, chr(data[p])))
                p += 1
            else:
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
                tags.append("%s:i:%d" % (tag, struct.unpack_from(fmt, data, p)[0]))
                p += struct.calcsize(fmt)
        rname = refs[rid][0] if rid >= 0 else "*"
        rnext = "*" if mrid < 0 else ("=" if mrid == rid else refs[mrid][0])
        recs.append([name, str(flag), rname, str(pos + 1), str(mapq), cig, rnext, str(mpos + 1), str(tlen), seq, qual] + (sorted(tags) if sort_tags else tags))
        at = end
    return text, refs, recs


def bam_case(g, lib, device=None, packed=False, se=False, on_device=False):
    """se: the single-end mapper on the first FASTQ alone (SetSamRecord(p, 0)), against the ref_se.* goldens.
    on_device: the records are formatted by the kernels of fq_emit.h inside the calls (fq_ctx_attach_bam)."""
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=lib) if device is None else api.Index(g["prefix"], device=device, lib=lib)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"], single_end=1 if se else 0), max_pairs=max(16, g["batch"]))
    path = os.path.join(g["dir"], "got_se.bam" if se else "got.bam")
    bam = api.BamWriter(ix, os.path.join(g["dir"], "genome.fai"), path)
    if on_device:
        bam.attach(al)
    if se:
        api.align_stream(al, list(names), seq[:1], qual[:1], lens[:1], g["batch"], None, None, bam=bam)
    else:
        api.align_stream(al, names, seq, qual, lens, g["batch"], None, None, bam=bam, packed=packed)
    bam.close()
    al.close(); ix.close()
    check_bgzf(path)
    text, refs, recs = decode_bam(path, sort_tags=False)      # tags in file order: the reference writes them in its tag hash's slot order
    stem = "ref_se" if se else "ref"
    want_hdr = open(os.path.join(g["dir"], stem + ".bamhdr")).read()
    assert text == want_hdr, "header:\n%s\nvs the reference's\n%s" % (text, want_hdr)
    assert [r[0] for r in refs] == [l.split("\t")[1][3:] for l in want_hdr.splitlines() if l.startswith("@SQ")]
    want = []
    for line in open(os.path.join(g["dir"], stem + ".bamtxt")):
        want.append(line.rstrip("\n").split("\t"))
    assert len(recs) == len(want), "%d records vs %d" % (len(recs), len(want))
    for i, (a, b) in enumerate(zip(recs, want)):
        assert a == b, "record %d:\n got  %s\n want %s" % (i, "\t".join(a)[:400], "\t".join(b)[:400])


SIDES = pytest.mark.parametrize("on_device", [False, True], ids=["host_formatter", "device_formatter"])


@SIDES
@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_bam_records_match_reference(tag, on_device, golden_cases, emu_lib):
    bam_case(golden_cases[tag], emu_lib, on_device=on_device, packed=on_device and tag in ("qc", "trim76"))


SE_CONSUMER_TAGS = [t for t in golden_util.se_case_tags() if os.path.exists(os.path.join(golden_util.GOLD, t, "ref_se.bamtxt.gz"))]


@SIDES
@pytest.mark.parametrize("tag", SE_CONSUMER_TAGS)
def test_single_end_bam_records_match_reference(tag, on_device, golden_cases, emu_lib):
    bam_case(golden_cases[tag], emu_lib, se=True, on_device=on_device)
