"""The FASTQ front end (fq_fastq.cpp): the same records whatever the container (plain text, one gzip member, several gzip members,
BGZF), the block size and the number of threads; kseq_read3_fpc's tokens on odd input (the byte-wise path); the read-slot model.
The comparison with the reference's own reader on quirky files is tests/test_cli_fastq_quirks.py (through the command line)."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libfq_emu.so")


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libfq_emu.so"])
    from fastquick_amd import api
    return api.load_library(EMU)


def bgzf_bytes(data: bytes, member: int = 3000) -> bytes:
    """BGZF as bgzip writes it (SAM spec 4.1): members of at most 64 KiB with a BC extra field, then the empty end-of-file member"""
    out = bytearray()
    chunks = [data[i:i + member] for i in range(0, len(data), member)] + [b""]
    for ch in chunks:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        bsize = 12 + 6 + len(comp) + 8 - 1
        out += struct.pack("<4BI2BH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += comp + struct.pack("<II", zlib.crc32(ch) & 0xffffffff, len(ch))
    return bytes(out)


def make_fastq(rng, n, ragged=False, names_vary=True):
    recs = []
    for i in range(n):
        L = int(rng.integers(20, 151)) if ragged else 150
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        qual = bytes(rng.integers(33, 74, L).astype(np.uint8))     # may start with '@' or '+'
        nm = ("r%d%s" % (i, "x" * int(rng.integers(0, 9)) if names_vary else "")).encode()
        recs.append((nm, seq, qual))
    text = b"".join(b"@" + nm + b" comment +@>\n" + s + b"\n+" + (nm if i % 3 == 0 else b"") + b"\n" + q + b"\n" for i, (nm, s, q) in enumerate(recs))
    return recs, text


def read_all(api, lib, path, threads, block, chunk=777, **kw):
    f = api.FastqFile(path, threads=threads, block_bytes=block, lib=lib, **kw)
    out = []
    while True:
        seq, qual, lens, names = f.read(chunk)
        if len(lens) == 0:
            break
        for r in range(len(lens)):
            L = lens[r]
            out.append((bytes(names[r]).split(b"\0")[0], bytes(seq[r, :L]), bytes(qual[r, :L]), bytes(seq[r, L:]), bytes(qual[r, L:])))
        if len(lens) < chunk:
            break
    bg = f.is_bgzf
    f.close()
    return out, bg


def test_same_records_from_every_container(lib, tmp_path):
    from fastquick_amd import api
    rng = np.random.default_rng(5)
    recs, text = make_fastq(rng, 5000)
    paths = {}
    paths["plain"] = str(tmp_path / "a.fq"); open(paths["plain"], "wb").write(text)
    paths["gz"] = str(tmp_path / "a.fq.gz"); open(paths["gz"], "wb").write(gzip.compress(text, 6))
    cut = [0, 100000, 100001, 700000, len(text)]
    paths["multi"] = str(tmp_path / "m.fq.gz"); open(paths["multi"], "wb").write(b"".join(gzip.compress(text[a:b], 1) for a, b in zip(cut[:-1], cut[1:])))
    paths["bgzf"] = str(tmp_path / "b.fq.gz"); open(paths["bgzf"], "wb").write(bgzf_bytes(text))
    want = [(nm, s, q) for nm, s, q in recs]
    for kind, path in paths.items():
        for threads, block in ((1, 0), (4, 0), (3, 5000), (2, 1000), (4, 300)):
            got, bg = read_all(api, lib, path, threads, block, slot_mode=api.FastqFile.SLOTS_FRESH)
            assert bg == (kind == "bgzf")
            assert [(g[0], g[1], g[2]) for g in got] == want, (kind, threads, block)
            assert all(set(g[3]) <= {0} and set(g[4]) <= {0} for g in got)      # rows are cleared behind the read


def test_odd_files_take_the_bytewise_path(lib, tmp_path):
    """wrapped base lines, blank lines between records, a record without line end at the end: kseq_read3_fpc's tokens"""
    from fastquick_amd import api
    text = (b"@a 1\nACGT\nAC\n+\nIIIIII\n"            # bases over two lines, quality on one (kseq reads len(seq) quality bytes)
            b"\n\n@b\nGGGG\n+\n@@@@\n"                 # blank lines in front; quality of '@'
            b"@c\tx\nTT TT\n+c\nJJJJ\n"                # a blank inside the base line is dropped: 4 bases
            b"@d\nAAAA\n+\nKKKK")                       # no line end: not returned
    for kind in ("plain", "bgzf"):
        path = str(tmp_path / ("odd." + kind))
        open(path, "wb").write(text if kind == "plain" else bgzf_bytes(text, 17))
        for threads, block in ((1, 0), (2, 300)):
            f = api.FastqFile(path, threads=threads, block_bytes=block, slot_mode=api.FastqFile.SLOTS_FRESH, lib=lib)
            seq, qual, lens, names = f.read(100)
            got = [(bytes(names[r]).split(b"\0")[0], bytes(seq[r, :lens[r]]), bytes(qual[r, :lens[r]])) for r in range(len(lens))]
            assert got == [(b"a", b"ACGTAC", b"IIIIII"), (b"b", b"GGGG", b"@@@@"), (b"c", b"TTTT", b"JJJJ")]
            assert f.dropped_record() == "d"
            f.close()


def test_refusals_carry_the_references_messages(lib, tmp_path):
    from fastquick_amd import api
    cases = {b"@a\nACGT\n+\nIII\n@b\nAC\n+\nII\n": "Error:a this fastq file contains reads with different length",
             b">a\nACGT\n>b\nAC\n": "FASTA input is not supported"}
    for text, msg in cases.items():
        path = str(tmp_path / "bad.fq")
        open(path, "wb").write(text)
        f = api.FastqFile(path, threads=2, lib=lib)
        with pytest.raises(api.FastquickError) as e:
            f.read(10)
        assert msg in str(e.value)
        f.close()
    path = str(tmp_path / "long.fq")
    open(path, "wb").write(b"@long\n" + b"A" * 200 + b"\n+\n" + b"I" * 200 + b"\n")
    f = api.FastqFile(path, threads=1, lib=lib)
    with pytest.raises(api.FastquickError) as e:
        f.read(10)
    assert "longer than the batch rows (200 > 160)" in str(e.value)
    f.close()


def slot_model(recs, batch_pairs, stride, reused_names):
    """ReadSlots of round 2's command line: slot = record % batch_pairs of set (record / batch_pairs) & 1"""
    names = [dict(), dict()]
    bases = [dict(), dict()]
    out = []
    for g, (nm, seq, qual) in enumerate(recs):
        st, slot = (g // batch_pairs) & 1, g % batch_pairs
        h = bases[st].get(slot, bytes(96))
        row = bytearray(seq) + bytearray(stride - len(seq))
        for i in range(len(seq), min(96, stride)):
            row[i] = h[i]
        nh = bytearray(h); nh[:min(len(seq), 96)] = seq[:96]
        bases[st][slot] = bytes(nh)
        if reused_names:
            b = bytearray(names[st].get(slot, b""))
            if len(b) < len(nm):
                b += bytes(len(nm) - len(b))
            b[:len(nm)] = nm
            names[st][slot] = bytes(b)
            nm = bytes(b).split(b"\0")[0]
        out.append((nm, bytes(row)))
    return out


@pytest.mark.parametrize("reused", [True, False])
def test_slot_history(lib, tmp_path, reused):
    from fastquick_amd import api
    rng = np.random.default_rng(11)
    recs, text = make_fastq(rng, 3000, ragged=True)
    path = str(tmp_path / "r.fq.gz")
    open(path, "wb").write(bgzf_bytes(text))
    want = slot_model(recs, 37, 160, reused)
    for threads, block in ((1, 0), (4, 2000)):
        got, _ = read_all(api, lib, path, threads, block, chunk=500, batch_pairs=37,
                          slot_mode=api.FastqFile.SLOTS_REUSED if reused else api.FastqFile.SLOTS_CLEAN_NAMES)
        assert [(g[0], g[1] + g[3]) for g in got] == want
