"""The device's BGZF compressor (csrc/fq_deflate.h: a wavefront per block of the BAM record stream -- greedy LZ77 over a hash table in LDS, one
fixed-Huffman block, CRC-32, the BC extra field) against zlib: every member must be a well-formed BGZF block that zlib inflates to exactly the
bytes that went in, with the right CRC-32 and ISIZE.  CPU tier: the kernel body on the host-loop backend; GPU tier: the kernel."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from fastquick_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))


def members(blob):
    out, at, n = [], 0, 0
    while at < len(blob):
        assert blob[at:at + 4] == b"\x1f\x8b\x08\x04" and blob[at + 12:at + 16] == b"BC\x02\x00", "not a BGZF member at %d" % at
        bsize = struct.unpack_from("<H", blob, at + 16)[0] + 1
        assert bsize <= 65536
        m = blob[at:at + bsize]
        data = zlib.decompress(m[18:-8], -15)
        crc, isz = struct.unpack("<II", m[-8:])
        assert zlib.crc32(data) == crc and len(data) == isz
        out.append(data)
        at += bsize
        n += 1
    return b"".join(out), n


def cases():
    rng = np.random.default_rng(1)
    rec = lambda i: struct.pack("<iiBBHHHiiii", 3, 1000 + i, 11, 60, 4681, 1, 99, 150, 3, 1300 + i, 450) + b"r%09d\0" % i + \
        rng.integers(0, 256, 75, dtype=np.uint8).tobytes() + bytes([37] * 150) + b"XTAUNMC\0SMC%AMC%X0C\1X1C\0XMC\0XOC\0XGC\0MDZ150\0"
    return {
        "empty": b"", "one_byte": b"A", "short": b"hello hello hello hello",
        "zeros": bytes(200000),
        "random": rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(),
        "nine_bit_literals": rng.integers(144, 256, 3 * 0xd000, dtype=np.uint8).tobytes(),      # the worst case of the fixed code: a member must still fit 64 KiB
        "fastq_text": b"@r000000001\nACGTACGTTTGACCA\n+\nFFFFFFFFFFFFFF:\n" * 5000,
        "bam_like": b"".join(rec(i) for i in range(3000)),
        "exact_block": rng.integers(0, 4, 0xd000, dtype=np.uint8).tobytes(),
        "block_plus_one": rng.integers(0, 4, 0xd000 + 1, dtype=np.uint8).tobytes(),
        "long_runs": b"ab" * 70000 + b"c" * 300 + b"xyz",
        "far_repeats": (rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()) * 3,             # repeats beyond DEFLATE's 32 KiB window inside one 52 KiB block
    }


def check(lib, device=0):
    for name, data in cases().items():
        z, _ms = api.bgzf_deflate_device(data, device=device, lib=lib)
        back, n = members(z)
        assert back == data, name
        assert n == (len(data) + 0xd000 - 1) // 0xd000, name


def test_device_compressor_body_on_the_host_loop_backend():
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so"])
    check(api.load_library(os.path.join(emu, "libfq_emu.so")))


@pytest.mark.gpu
def test_device_compressor_kernel():
    lib = api.load_library()
    check(lib)
    # ... and a few hundred MB of BAM-like records: every member sound, and the kernel's rate on the record
    rng = np.random.default_rng(7)
    one = cases()["bam_like"]
    data = one * 200
    z, ms = api.bgzf_deflate_device(data, lib=lib)
    back, n = members(z)
    assert back == data
    print("device BGZF: %.1f MB -> %.1f MB in %.2f ms = %.1f GB/s" % (len(data) / 1e6, len(z) / 1e6, ms, len(data) / ms / 1e6))
