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
EMU = os.environ.get("FQ_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libfq_emu.so")      # (FQ_EMU_LIB: a sanitizer build of the same library)


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


def _outcome(api, lib, path, threads, block):
    """every record the reader returns, then how it ended: ("end",) or ("error", message)"""
    f = api.FastqFile(path, threads=threads, block_bytes=block, lib=lib, slot_mode=api.FastqFile.SLOTS_FRESH)
    recs, end = [], ("end",)
    try:
        while True:
            seq, qual, lens, names = f.read(5000)
            if len(lens) == 0:
                break
            recs += [(bytes(names[r]).split(b"\0")[0], bytes(seq[r, :lens[r]]), bytes(qual[r, :lens[r]])) for r in range(len(lens))]
            if len(lens) < 5000:
                break
    except api.FastquickError as e:
        end = ("error", str(e))
    f.close()
    return recs, end


def _gz_member(data: bytes, level: int, flags: int = 0, flush_every: int = 0) -> bytes:
    """a gzip member written by hand: any header field (FEXTRA 4, FNAME 8, FCOMMENT 16, FHCRC 2), any level, flush points"""
    head = struct.pack("<4BI2B", 0x1f, 0x8b, 8, flags, 0, 0, 3)
    if flags & 4:
        head += struct.pack("<H", 7) + b"XY\x03\x00abc"
    if flags & 8:
        head += b"reads.fq\0"
    if flags & 16:
        head += b"a comment\0"
    if flags & 2:
        head += struct.pack("<H", zlib.crc32(head) & 0xffff)
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = b""
    if flush_every:
        for a in range(0, len(data), flush_every):
            body += co.compress(data[a:a + flush_every]) + co.flush(zlib.Z_FULL_FLUSH if (a // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
        body += co.flush()
    else:
        body = co.compress(data) + co.flush()
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def test_gzip_streams_through_the_fast_decoder_are_gzreads_records(lib, tmp_path, monkeypatch):
    """VERDICT r4 item 6 -- a gzip file that is not BGZF (what sequencers and `gzip` write) is decoded by fq_inflate.h's decoder as a stream, block
    after block with the 32 KiB window carried along, instead of gzread: the same records for every level, flush points, stored blocks, header
    fields, members back to back (an empty one among them) -- and on truncated and damaged files whatever gzread makes of them (the records before
    the damage, then its message): FASTQUICK_ZLIB_INFLATE=1 is that reader."""
    from fastquick_amd import api
    rng = np.random.default_rng(11)
    recs, text = make_fastq(rng, 30000, names_vary=True)          # ~10 MB: dozens of 192 KiB blocks, matches across their boundaries
    want = [(nm, s, q) for nm, s, q in recs]
    files = {
        "level6": _gz_member(text, 6), "level1": _gz_member(text, 1), "level9_fields": _gz_member(text, 9, flags=4 | 8 | 16 | 2),
        "stored": _gz_member(text, 0), "flushed": _gz_member(text, 6, flush_every=70001),
        "members": _gz_member(text[:3000000], 6, flags=8) + _gz_member(b"", 6) + _gz_member(text[3000000:3000001], 1) + _gz_member(text[3000001:], 4, flush_every=500000),
        "python_gzip": gzip.compress(text, 6),
    }
    for name, blob in files.items():
        path = str(tmp_path / (name + ".fq.gz"))
        open(path, "wb").write(blob)
        for threads, block in ((4, 0), (2, 1 << 18)):
            got, end = _outcome(api, lib, path, threads, block)
            assert end == ("end",) and got == want, (name, threads, block)
    # ---- damage: the same outcome as gzread's, whatever it is
    base = files["level6"]
    damaged = {"truncated_mid": base[:len(base) // 2], "truncated_trailer": base[:-5], "truncated_1": base[:-1], "truncated_head": base[:7],
               "wrong_crc": base[:-8] + bytes([base[-8] ^ 1]) + base[-7:], "wrong_size": base[:-1] + bytes([base[-1] ^ 0x40]),
               "garbage_behind": base + b"not a gzip member at all", "zeros_behind": base + b"\0" * 100,
               "second_member_cut": base + _gz_member(text[:50000], 6)[:3000],
               "reserved_flag": base[:3] + bytes([0x20]) + base[4:]}
    for k in range(6):
        at = int(rng.integers(20, len(base) - 20))
        damaged["bitflip_%d" % k] = base[:at] + bytes([base[at] ^ (1 << int(rng.integers(0, 8)))]) + base[at + 1:]
    for name, blob in damaged.items():
        path = str(tmp_path / (name + ".fq.gz"))
        open(path, "wb").write(blob)
        monkeypatch.setenv("FASTQUICK_ZLIB_INFLATE", "1")
        ref, ref_end = _outcome(api, lib, path, 2, 0)
        monkeypatch.delenv("FASTQUICK_ZLIB_INFLATE")
        got, end = _outcome(api, lib, path, 2, 0)
        assert end == ref_end, (name, end, ref_end)
        assert got == ref, (name, len(got), len(ref))
    # (with blocks smaller than the file the two readers may differ, on a DAMAGED file, in how many of the same records they return before the
    #  error -- gzread reads up to 128 KiB ahead behind a member's header: tests/fuzz_gzip_stream.py states and soaks that property)
