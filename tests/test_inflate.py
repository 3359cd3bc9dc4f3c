"""The front end's DEFLATE decoder and CRC-32 (fastquick_amd/csrc/fq_inflate.h) against zlib: same bytes for every stream zlib
produces, refusal (never wrong bytes) for damaged ones.  CPU tier: through the host-loop library's C ABI (fq_inflate_raw, fq_crc32)."""
import ctypes as C
import os
import random
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libfq_emu.so")


@pytest.fixture(scope="module", params=["host-loop library (g++)", "product library (hipcc's clang)"])
def lib(request):
    """both builds of the decoder: the two compilers disagree about what its loops should look like"""
    from fastquick_amd import api
    if request.param.startswith("host-loop"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libfq_emu.so"])
        return api.load_library(EMU)
    if not os.path.exists(api.DEFAULT_LIB):
        pytest.skip("product library not built")
    return api.load_library(api.DEFAULT_LIB)     # (the reader's entry points need no device)


def raw_deflate(data: bytes, level: int, strategy: int = zlib.Z_DEFAULT_STRATEGY, mem: int = 8) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def inflate(lib, comp: bytes, n: int):
    out = C.create_string_buffer(max(n, 1))
    rc = lib.fq_inflate_raw(comp, len(comp), out, n)
    return rc, out.raw[:n]


def samples():
    rng = random.Random(7)
    nrng = np.random.default_rng(7)
    yield b""
    yield b"A"
    yield b"ACGT" * 5000                                            # long matches, short distances
    yield b"F" * 70000                                              # distance 1, longer than one BGZF member would be
    yield bytes(nrng.integers(0, 256, 40000, dtype=np.uint8))       # incompressible: stored blocks at any level
    yield bytes(nrng.integers(0, 4, 65280, dtype=np.uint8) + 65)    # 2 bits of entropy per byte
    yield bytes(rng.choice(b"ACGTN") for _ in range(30000)) + b"\n" + bytes(rng.choice(b"F:,#") for _ in range(30000))
    fq = []
    for i in range(400):                                            # what the files hold
        L = rng.choice((76, 100, 150, 151))
        fq.append(b"@read%d/1 extra words\n" % i + bytes(rng.choice(b"ACGT") for _ in range(L)) + b"\n+\n" + bytes(rng.choice(b"F:,#FFFF") for _ in range(L)) + b"\n")
    yield b"".join(fq)
    yield bytes(range(256)) * 300                                   # every literal, distance 256
    yield bytes(nrng.integers(0, 256, 300, dtype=np.uint8)) * 200   # distance 300 repeats
    for n in (1, 2, 7, 8, 9, 257, 258, 259, 300, 4095, 4096, 65535):
        yield bytes(rng.choice(b"ab") for _ in range(n))


def test_same_bytes_as_zlib_for_every_level_and_strategy(lib):
    n_streams = 0
    for data in samples():
        for level in (0, 1, 2, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                comp = raw_deflate(data, level, strategy)
                rc, out = inflate(lib, comp, len(data))
                assert rc == 0, (len(data), level, strategy)
                assert out == data, (len(data), level, strategy)
                n_streams += 1
                assert lib.fq_crc32(data, len(data)) == zlib.crc32(data)
    assert n_streams > 400


def test_flushed_streams_of_many_blocks(lib):
    rng = random.Random(11)
    data = bytes(rng.choice(b"ACGT\nF:#@+") for _ in range(200000))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for i in range(0, len(data), 7001):                              # sync / full flushes: empty stored blocks between Huffman blocks
        parts.append(c.compress(data[i:i + 7001]))
        parts.append(c.flush(zlib.Z_SYNC_FLUSH if (i // 7001) % 2 else zlib.Z_FULL_FLUSH))
    parts.append(c.flush())
    comp = b"".join(parts)
    rc, out = inflate(lib, comp, len(data))
    assert rc == 0 and out == data


def test_wrong_size_truncation_and_damage_are_refused_or_decode_to_what_zlib_says(lib):
    rng = random.Random(3)
    data = b"".join(b"@r%d\n" % i + bytes(rng.choice(b"ACGT") for _ in range(100)) + b"\n+\n" + b"F" * 100 + b"\n" for i in range(300))
    comp = raw_deflate(data, 1)
    assert inflate(lib, comp, len(data))[0] == 0
    assert inflate(lib, comp, len(data) - 1)[0] != 0                 # the trailer promises fewer bytes than the stream holds
    assert inflate(lib, comp, len(data) + 1)[0] != 0                 # ... or more
    for cut in (1, 2, 5, len(comp) // 2, len(comp) - 1):
        assert inflate(lib, comp[:cut], len(data))[0] != 0           # truncated
    assert inflate(lib, comp + b"\x00\x00garbage", len(data))[0] == 0   # bytes behind the end of the stream are the caller's business (as with inflate())
    # flipped bits: the decoder either refuses, or returns exactly what zlib's inflate() returns for the same bytes
    agree = refused = 0
    for _ in range(400):
        b = bytearray(comp)
        pos = rng.randrange(len(b))
        b[pos] ^= 1 << rng.randrange(8)
        rc, out = inflate(lib, bytes(b), len(data))
        d = zlib.decompressobj(-15)
        try:
            ref = d.decompress(bytes(b)) + d.flush()
            ok = d.eof
        except zlib.error:
            ref, ok = None, False
        if rc == 0:
            assert ok and ref == out
            agree += 1
        else:
            assert not (ok and len(ref) == len(data)), "refused a stream zlib decodes to the promised size"
            refused += 1
    assert agree + refused == 400 and refused > 100


def test_crc32_of_odd_sizes_and_alignments(lib):
    buf = bytes(np.random.default_rng(5).integers(0, 256, 5000, dtype=np.uint8))
    for off in range(0, 9):
        for n in (0, 1, 7, 15, 16, 17, 31, 33, 255, 4096, 4991):
            piece = buf[off:off + n]
            arr = (C.c_ubyte * (len(buf))).from_buffer_copy(buf)
            assert lib.fq_crc32(C.byref(arr, off), len(piece)) == zlib.crc32(piece)


def test_random_mixtures_of_literals_and_repeats(lib):
    """seeded streams built from random literals and copies at every distance class (1, 2-7, 8+, up to the 32 KiB window)"""
    rng = random.Random(2024)
    for it in range(300):
        buf = bytearray()
        target = rng.choice((50, 700, 5000, 40000, 66000))
        alpha = rng.choice((b"ACGT", b"ACGTN\n@+F:,#", bytes(range(256))))
        while len(buf) < target:
            if buf and rng.random() < 0.6:
                d = min(len(buf), rng.choice((1, 2, 3, 5, 7, 8, 9, 16, 150, 302, 4000, 32768)))
                n = rng.choice((3, 4, 8, 17, 150, 258, 259, 1000))
                start = len(buf) - d
                for i in range(n):
                    buf.append(buf[start + i])
            else:
                buf += bytes(rng.choice(alpha) for _ in range(rng.choice((1, 2, 10, 100))))
        data = bytes(buf)
        level = rng.choice((1, 1, 6, 9))
        comp = raw_deflate(data, level, rng.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED)), mem=rng.choice((1, 8, 9)))
        rc, out = inflate(lib, comp, len(data))
        assert rc == 0 and out == data, (it, len(data), level)


def test_decoder_under_address_and_ub_sanitizers(tmp_path):
    """exact-size heap buffers: an overrun on any of 4,000 valid and damaged streams stops the run (CPU build only: no sanitizer on the device)"""
    exe = str(tmp_path / "inflate_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "fastquick_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "emu", "inflate_fuzz.cpp"), "-lz"])
    out = subprocess.run([exe, "4000"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "mismatches 0" in out.stdout
