"""The FASTQ front end on the device (fastquick_amd/csrc/fq_frontend.h / fq_frontend.cpp): the BGZF member decoder (`k_inflate_bgzf`, one
wavefront per member) against zlib -- the reference reads its input through zlib's gzread (libbwa/bwaseqio.c:41-52).  Same bytes for every
stream zlib produces, refusal (never wrong bytes) for damaged ones: a member the device refuses goes to the host's decoder and then to zlib,
whose verdict stands.
CPU tier: the same kernel body as a wavefront of one lane through the host-loop library (tests/emu, test infrastructure), and a fuzz of it
under AddressSanitizer / UBSan; GPU tier: the HIP kernel."""
import os
import random
import subprocess
import zlib

import numpy as np
import pytest

from fastquick_amd import api, synth
from test_inflate import raw_deflate, samples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libfq_emu.so"])
    return api.load_library(os.path.join(EMU_DIR, "libfq_emu.so"))


TIERS = [pytest.param("emu", id="host-loop"), pytest.param("hip", id="gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=TIERS)
def lib(request):
    return emu() if request.param == "emu" else api.load_library()


def test_every_level_and_strategy_decodes_to_zlibs_bytes(lib):
    streams, datas = [], []
    for data in samples():
        for level in (0, 1, 2, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                streams.append((raw_deflate(data, level, strategy), len(data), zlib.crc32(data)))
                datas.append(data)
    assert len(streams) > 400
    res, _ = api.inflate_device(streams, lib=lib)
    for k, (st, out) in enumerate(res):
        assert st == 0 and out == datas[k], (k, st, len(datas[k]))


def test_flushed_streams_of_many_blocks(lib):
    rng = random.Random(11)
    data = bytes(rng.choice(b"ACGT\nF:#@+") for _ in range(200000))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for i in range(0, len(data), 7001):                              # sync / full flushes: empty stored blocks between Huffman blocks
        parts.append(c.compress(data[i:i + 7001]))
        parts.append(c.flush(zlib.Z_SYNC_FLUSH if (i // 7001) % 2 else zlib.Z_FULL_FLUSH))
    parts.append(c.flush())
    (st, out), = api.inflate_device([(b"".join(parts), len(data), zlib.crc32(data))], lib=lib)[0]
    assert st == 0 and out == data


def test_wrong_size_truncation_and_damage_are_refused_or_decode_to_what_zlib_says(lib):
    rng = random.Random(3)
    data = b"".join(b"@r%d\n" % i + bytes(rng.choice(b"ACGT") for _ in range(100)) + b"\n+\n" + b"F" * 100 + b"\n" for i in range(300))
    comp = raw_deflate(data, 1)
    crc = zlib.crc32(data)
    cases = [(comp, len(data), crc), (comp, len(data) - 1, zlib.crc32(data[:-1])), (comp, len(data) + 1, crc), (comp, len(data), crc ^ 0x10)]
    cases += [(comp[:cut], len(data), crc) for cut in (1, 2, 5, len(comp) // 2, len(comp) - 1)]
    cases.append((comp + b"\x00\x00garbage", len(data), crc))          # bytes behind the end of the stream are the caller's business (as with inflate())
    res, _ = api.inflate_device(cases, lib=lib)
    assert [st for st, _ in res[:4]] == [0, 1, 1, 2] and res[0][1] == data
    assert all(st == 1 for st, _ in res[4:9])
    assert res[9][0] == 0 and res[9][1] == data
    # flipped bits: the decoder either refuses, or returns exactly what zlib's inflate() returns for the same bytes
    flips = []
    for _ in range(400):
        b = bytearray(comp)
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        flips.append(bytes(b))
    res, _ = api.inflate_device([(f, len(data), crc) for f in flips], lib=lib)
    agree = refused = 0
    for f, (st, out) in zip(flips, res):
        d = zlib.decompressobj(-15)
        try:
            ref = d.decompress(f) + d.flush()
            ok = d.eof and len(ref) == len(data)
        except zlib.error:
            ref, ok = None, False
        if st == 0:
            assert ok and ref == out and zlib.crc32(out) == crc
            agree += 1
        else:
            assert st in (1, 2) and not (ok and zlib.crc32(ref) == crc), "refused a stream zlib decodes to the promised size and CRC"
            refused += 1
    assert agree + refused == 400 and refused > 300


def test_random_mixtures_of_literals_and_repeats(lib):
    """seeded streams built from random literals and copies at every distance class (inside one wavefront's width, inside the LDS ring, beyond it)"""
    rng = random.Random(2024)
    streams, datas = [], []
    for it in range(300):
        buf = bytearray()
        target = rng.choice((50, 700, 5000, 40000, 66000))
        alpha = rng.choice((b"ACGT", b"ACGTN\n@+F:,#", bytes(range(256))))
        while len(buf) < target:
            if buf and rng.random() < 0.6:
                d = min(len(buf), rng.choice((1, 2, 3, 5, 7, 8, 9, 16, 63, 64, 65, 150, 302, 3900, 3968, 3969, 4000, 4096, 4097, 32768)))
                n = rng.choice((3, 4, 8, 17, 63, 64, 65, 150, 258, 259, 1000))
                start = len(buf) - d
                for i in range(n):
                    buf.append(buf[start + i])
            else:
                buf += bytes(rng.choice(alpha) for _ in range(rng.choice((1, 2, 10, 100))))
        data = bytes(buf)
        streams.append((raw_deflate(data, rng.choice((1, 1, 6, 9)), rng.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED)), mem=rng.choice((1, 8, 9))), len(data), zlib.crc32(data)))
        datas.append(data)
    res, _ = api.inflate_device(streams, lib=lib)
    for k, (st, out) in enumerate(res):
        assert st == 0 and out == datas[k], (k, len(datas[k]))


def test_a_bgzf_file_image_comes_out_as_its_text(lib):
    rng = np.random.default_rng(5)
    n, L = 3000, 150
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, L))]
    qual = np.frombuffer(b"F:,#", dtype=np.uint8)[rng.choice(4, size=(n, L), p=[0.7, 0.15, 0.1, 0.05])]
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for level in (1, 6):
            path = os.path.join(tmp, "r.fq")
            nbytes = synth.write_fastq_uniform(seq, qual, L, path, bgzf=False)
            text = open(path, "rb").read()
            blob = synth.bgzf_compress(text, level=level)
            out, status, _ = api.bgzf_inflate_device(blob, nbytes + 16, lib=lib)
            assert status and not any(status) and out.tobytes() == text


def test_device_decoder_under_address_and_ub_sanitizers(tmp_path):
    """the kernel body as a wavefront of one lane, exact-size heap buffers: valid and damaged members back to back at every alignment"""
    exe = str(tmp_path / "devinflate_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "fastquick_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(EMU_DIR, "devinflate_fuzz.cpp"), "-lz"])
    out = subprocess.run([exe, "1500"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    assert "mismatches 0" in out.stdout


# ---- the whole front end: BGZF files -> batches resident on the device -> fq_align_text ------------------------------------------------
import golden_util  # noqa: E402
import oracle_binding as ob  # noqa: E402


def bgzf_pair(g, tmp, level=6, member=3000):
    """the case's FASTQ pair as BGZF files of small members (many members, records cut anywhere by their borders)"""
    out = []
    for k in ("fq1", "fq2"):
        path = os.path.join(str(tmp), os.path.basename(g[k]) + ".gz")
        with open(path, "wb") as fh:
            fh.write(synth.bgzf_compress(open(g[k], "rb").read(), threads=2, level=level, member=member))
        out.append(path)
    return out


def run_front_end(lib, g, fq, stages, sam, chunk_batches=2, slot_mode=0, max_read_len=None, se=False, device=0):
    max_len = max_read_len or max(160, (g["qc_read_len"] + 15) // 16 * 16)
    fe = api.DeviceFrontEnd(fq[0], None if se else fq[1], batch_pairs=g["batch"], chunk_pairs=chunk_batches * g["batch"], slot_mode=slot_mode, max_read_len=max_len, device=device, lib=lib)
    ix = api.Index(g["prefix"], lib=lib) if lib is not None and getattr(lib, "_name", "").endswith("libfq_emu.so") else api.Index(g["prefix"], device=device, lib=lib)
    al = api.Aligner(ix, api.default_opts(lib, trim_qual=g["trim_qual"], batch_pairs=g["batch"], single_end=1 if se else 0), max_pairs=chunk_batches * g["batch"], debug=True)
    n_total, calls = 0, 0
    with open(stages, "wb") as st, open(sam, "wb") as sm:
        sm.write(ix.sam_header())
        while True:
            n, b = fe.next()
            if n <= 0:
                break
            al.align_text(b)
            st.write(al.stage_text())
            sm.write(al.sam_text())
            fe.release(b)
            n_total += n
            calls += 1
    stats = fe.stats()
    al.close(); ix.close(); fe.close()
    return n, n_total, calls, stats


@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_front_end_batches_align_to_the_references_output(tag, golden_cases, lib, tmp_path):
    """FASTQ.gz in, the reference's stage dumps and SAM text out, with nothing of the reads' text touched by the host: inflate, lines,
    records, filter keys, read slots and names on the device (the same kernel bodies in the host-loop tier)."""
    g = golden_cases[tag]
    fq = bgzf_pair(g, tmp_path)
    st, sam = str(tmp_path / "fe.stages"), str(tmp_path / "fe.sam")
    end, n_total, calls, stats = run_front_end(lib, g, fq, st, sam)
    assert end == 0, "a well-formed file must be the device's from its first record to its last"
    assert n_total == g["n_pairs"] and calls == -(-g["n_pairs"] // (2 * g["batch"]))
    assert stats["members"] > 10 and stats["refused"] == 0
    diffs = [d for d in ob.diff_stage_files(g["stages"], st)]
    assert not diffs, "\n".join(diffs)
    assert open(sam, "rb").read() == open(g["sam"], "rb").read()


def host_arrays(lib, fq, batch_pairs, slot_mode, stride=256, se=False):
    """the host path on the same files: fq_fastq_read (kseq_read3_fpc's tokens, read-slot history) + fq_pack_reads_into -> head, len, names"""
    import ctypes as C
    files = [api.FastqFile(p, threads=2, batch_pairs=batch_pairs, slot_mode=slot_mode, stride=stride, name_stride=304, lib=lib) for p in (fq[:1] if se else fq)]
    rows = [f.read(1 << 20) for f in files]
    for f in files:
        f.close()
    n = min(len(r[2]) for r in rows)
    seq = np.stack([r[0][:n] for r in rows]); qual = np.stack([r[1][:n] for r in rows]); lens = np.stack([r[2][:n] for r in rows])
    hp = api.HostPacked(seq, qual, lens, None, lib=lib)
    pb = hp.p.contents
    rows_n = n * len(files)
    head = np.ctypeslib.as_array(C.cast(pb.head, C.POINTER(C.c_uint64)), shape=(3 * rows_n,)).reshape(3, rows_n).copy()
    hp.free()
    names = np.concatenate([r[3][:n] for r in rows])
    return n, head, lens.reshape(-1).astype(np.uint16), names


def front_end_arrays(lib, fq, batch_pairs, chunk_pairs, slot_mode, max_read_len=256, se=False):
    fe = api.DeviceFrontEnd(fq[0], None if se else fq[1], batch_pairs=batch_pairs, chunk_pairs=chunk_pairs, slot_mode=slot_mode, max_read_len=max_read_len, lib=lib)
    heads, lens, names, total = [[], []], [[], []], [[], []], 0
    while True:
        n, b = fe.next()
        if n <= 0:
            break
        h, l, nm = fe.fetch(b, n, single_end=se)
        for e in range(1 if se else 2):
            heads[e].append(h[:, e * n:(e + 1) * n]); lens[e].append(l[e * n:(e + 1) * n])
            full = np.zeros((n, 304), dtype=np.uint8); full[:, :nm.shape[1]] = nm[e * n:(e + 1) * n]
            names[e].append(full)
        fe.release(b)
        total += n
    return n, total, heads, lens, names, fe


@pytest.mark.parametrize("slot_mode", [0, 1, 2], ids=["reused_slots", "clean_names", "fresh"])
@pytest.mark.parametrize("tag", golden_util.case_tags())
def test_front_end_arrays_are_the_host_readers_and_packers(tag, slot_mode, golden_cases, lib, tmp_path, monkeypatch):
    """the filter's keys, lengths and names the device forms from the text == fq_fastq_read + fq_pack_reads_into, bit for bit.
    (A chunk's new text is inflated before the length of the text carried over from the chunk before is known, a fixed distance into the
    buffer: with FASTQUICK_FE_HEADROOM that distance is the default -- the carried text fits in front -- , tiny -- it fits some times --, or
    none -- the new text is always moved back.)"""
    if slot_mode == 1:
        monkeypatch.setenv("FASTQUICK_FE_HEADROOM", "0")
    elif slot_mode == 2:
        monkeypatch.setenv("FASTQUICK_FE_HEADROOM", "4096")
    g = golden_cases[tag]
    fq = bgzf_pair(g, tmp_path, level=1, member=5000)
    B = 64
    n, head, lens, names = host_arrays(lib, fq, B, slot_mode, stride=256)
    end, total, dh, dl, dn, fe = front_end_arrays(lib, fq, B, 3 * B, slot_mode)
    fe.close()
    assert end == 0 and total == n == g["n_pairs"]
    for e in range(2):
        assert np.array_equal(np.concatenate(dh[e], axis=1), head[:, e * n:(e + 1) * n]), "filter keys of end %d" % e
        assert np.array_equal(np.concatenate(dl[e]), lens[e * n:(e + 1) * n])
        assert np.array_equal(np.concatenate(dn[e]), names[e * n:(e + 1) * n]), "names of end %d" % e


@pytest.mark.parametrize("overlap", ["1", "0"], ids=["inflating_ahead", "one_after_the_other"])
def test_front_end_hands_over_at_a_reference_batch_boundary(overlap, golden_cases, lib, tmp_path, monkeypatch):
    """an odd record in the middle of a file: the device's part ends at the boundary of the reference batch that holds it; the host readers
    standing there return the rest -- together exactly the records (names, read-slot history included) the host path alone returns"""
    monkeypatch.setenv("FASTQUICK_FE_OVERLAP", overlap)
    monkeypatch.setenv("FASTQUICK_FE_HEADROOM", "8192")
    g = golden_cases["qc"]
    recs = [open(g[k], "rb").read().split(b"\n") for k in ("fq1", "fq2")]
    B, odd = 256, 1500
    fq = []
    for e in range(2):
        lines = recs[e]
        out = []
        for i in range(len(lines) // 4):
            nm, s, p, q = lines[4 * i:4 * i + 4]
            if i % 3 == 1:
                nm = nm + b":" + b"y" * (i % 40)               # names of many lengths: a slot's earlier, longer name shows behind a shorter one
            if i % 5 == 2:
                s, q = s[:60 + i % 80], q[:60 + i % 80]        # short reads: the slot's earlier bases stand behind them in the filter's window
            if i == odd and e == 0:
                s = s[:70] + b"\n" + s[70:]                    # a wrapped base line
            out += [nm, s, p, q]
        path = str(tmp_path / ("odd_%d.fq.gz" % (e + 1)))
        with open(path, "wb") as fh:
            fh.write(synth.bgzf_compress(b"\n".join(out) + b"\n", threads=2, level=6, member=4000))
        fq.append(path)
    n, head, lens, names = host_arrays(lib, fq, B, 0)
    end, total, dh, dl, dn, fe = front_end_arrays(lib, fq, B, 2 * B, 0)
    assert end == api.FQ_EFALLBACK and total == odd // B * B
    readers = fe.handover(threads=2, stride=256, name_stride=304)
    rest = [r.read(1 << 20) for r in readers]
    for r in readers:
        r.close()
    assert fe.unequal_lengths()
    fe.close()
    assert min(len(r[2]) for r in rest) == n - total
    for e in range(2):
        assert np.array_equal(np.concatenate(dl[e]), lens[e * n:e * n + total])
        assert np.array_equal(np.concatenate(dh[e], axis=1), head[:, e * n:e * n + total])
        assert np.array_equal(np.concatenate(dn[e]), names[e * n:e * n + total])
        assert np.array_equal(rest[e][2][:n - total], lens[e * n + total:(e + 1) * n])
        assert np.array_equal(rest[e][3][:n - total], names[e * n + total:(e + 1) * n]), "names behind the hand-over (slot history travelled)"
        hp = api.HostPacked(np.stack([rest[0][0][:n - total], rest[1][0][:n - total]]), np.stack([rest[0][1][:n - total], rest[1][1][:n - total]]),
                            np.stack([rest[0][2][:n - total], rest[1][2][:n - total]]), None, lib=lib)
        import ctypes as C
        hh = np.ctypeslib.as_array(C.cast(hp.p.contents.head, C.POINTER(C.c_uint64)), shape=(3 * 2 * (n - total),)).reshape(3, -1).copy()
        hp.free()
        assert np.array_equal(hh[:, e * (n - total):(e + 1) * (n - total)], head[:, e * n + total:(e + 1) * n]), "filter keys behind the hand-over (bases of the slots travelled)"


@pytest.mark.parametrize("slot_mode", [0, 2], ids=["reused_slots", "fresh"])
def test_any_printable_byte_in_the_bases_and_qualities(slot_mode, lib, tmp_path):
    """IUPAC codes, lower case, '.', '-', '*', digits in the base line (kseq_read3_fpc takes every isgraph() byte, libbwa/kseq.h:340; nst_nt4_table
    makes 4 of all but ACGTacgt) and every byte 33..126 in the quality line, '@', '+' and '>' at its start included: the device's filter keys,
    lengths and names == the host reader's and packer's"""
    rng = np.random.default_rng(11 + slot_mode)
    alpha = np.frombuffer(b"ACGTNacgtnRYKMSWBDHVryu.-*=~!0Z", dtype=np.uint8)
    fq = []
    for e in range(2):
        out = []
        for i in range(1500):
            L = int(rng.integers(30, 200))
            q = rng.integers(33, 127, L).astype(np.uint8)
            q[0] = (64, 43, 62, q[0])[i % 4]                   # '@', '+', '>' first
            out.append(b"@r%d/%d\n" % (i, e + 1) + bytes(rng.choice(alpha, L)) + b"\n+\n" + bytes(q) + b"\n")
        path = str(tmp_path / ("any_%d.fq.gz" % (e + 1)))
        with open(path, "wb") as fh:
            fh.write(synth.bgzf_compress(b"".join(out), threads=2, level=6, member=4000))
        fq.append(path)
    n, head, lens, names = host_arrays(lib, fq, 64, slot_mode, stride=256)
    end, total, dh, dl, dn, fe = front_end_arrays(lib, fq, 64, 192, slot_mode)
    fe.close()
    assert end == 0 and total == n == 1500
    for e in range(2):
        assert np.array_equal(np.concatenate(dh[e], axis=1), head[:, e * n:(e + 1) * n]), "filter keys of end %d" % e
        assert np.array_equal(np.concatenate(dl[e]), lens[e * n:(e + 1) * n])
        assert np.array_equal(np.concatenate(dn[e]), names[e * n:(e + 1) * n])


def test_carriage_returns_are_the_references_refusal(golden_cases, lib, tmp_path):
    """CR LF line ends from some record on: kseq_read3_fpc (libbwa/kseq.h:361-365) takes len(seq) quality bytes (isgraph() dropped the CR from
    the bases) and wants a line feed next -- it finds the CR, prints "this fastq file contains reads with different length" and exits. The
    device's part ends at the reference batch in front of that record; the host reader standing there refuses with the reference's words."""
    g = golden_cases["qc"]
    B, odd = 256, 700
    fq = []
    for e, k in enumerate(("fq1", "fq2")):
        lines = open(g[k], "rb").read().split(b"\n")
        head, tail = lines[:4 * odd], lines[4 * odd:]
        text = b"\n".join(head) + b"\n" + (b"\r\n".join(tail) if e == 0 else b"\n".join(tail))
        path = str(tmp_path / ("crlf_%d.fq.gz" % (e + 1)))
        with open(path, "wb") as fh:
            fh.write(synth.bgzf_compress(text, threads=2, level=6, member=4000))
        fq.append(path)
    end, total, dh, dl, dn, fe = front_end_arrays(lib, fq, B, 2 * B, 0)
    assert end == api.FQ_EFALLBACK and total == odd // B * B
    readers = fe.handover(threads=2, stride=256, name_stride=304)
    with pytest.raises(api.FastquickError, match="this fastq file contains reads with different length"):
        readers[0].read(1 << 20)
    for r in readers:
        r.close()
    fe.close()


@pytest.mark.gpu
def test_a_million_pair_bgzf_file_through_the_front_end(tmp_path):
    """1,048,576 pairs of BGZF FASTQ: every batch's keys equal the host packer's on the same rows"""
    import ctypes as C
    n, L = 1 << 20, 150
    rng = np.random.default_rng(17)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (2, n, L), dtype=np.uint8)]
    seq[0, ::1000, 17] = ord("N")
    qual = np.frombuffer(b"F:,#", dtype=np.uint8)[rng.choice(4, size=(2, n, L), p=[0.7, 0.15, 0.1, 0.05])]
    fq = [str(tmp_path / ("big_%d.fq.gz" % (e + 1))) for e in range(2)]
    for e in range(2):
        synth.write_fastq_uniform(seq[e], qual[e], L, fq[e], threads=16)
    hp = api.HostPacked(seq, qual, np.full((2, n), L, dtype=np.int32), None, threads=16)
    head = np.ctypeslib.as_array(C.cast(hp.p.contents.head, C.POINTER(C.c_uint64)), shape=(3 * 2 * n,)).reshape(3, 2 * n).copy()
    hp.free()
    fe = api.DeviceFrontEnd(fq[0], fq[1], batch_pairs=262144, chunk_pairs=2 * 262144, slot_mode=0, max_read_len=160)
    at = 0
    while True:
        m, b = fe.next()
        if m <= 0:
            break
        h, l, nm = fe.fetch(b, m)
        for e in range(2):
            assert np.array_equal(h[:, e * m:(e + 1) * m], head[:, e * n + at:e * n + at + m])
        assert (l == L).all() and bytes(nm[0][:10]) == b"r%09d" % at and bytes(nm[m][:10]) == b"r%09d" % at
        fe.release(b)
        at += m
    st = fe.stats()
    fe.close()
    assert m == 0 and at == n and st["refused"] == 0 and st["members"] > 9000


@pytest.mark.timeout(120)
def test_closing_in_the_middle_of_a_stream_does_not_hang(golden_cases, lib, tmp_path):
    """a front end closed after its first batch -- and one closed before any batch was taken -- while its reader and producer threads are at
    work or waiting for each other: fq_frontend_close stops and joins them"""
    g = golden_cases["qc"]
    fq = bgzf_pair(g, tmp_path, level=1, member=2000)
    for take in (1, 0):
        fe = api.DeviceFrontEnd(fq[0], fq[1], batch_pairs=64, chunk_pairs=64, slot_mode=0, max_read_len=256, lib=lib)
        for _ in range(take):
            n, b = fe.next()
            assert n == 64
        fe.close()
