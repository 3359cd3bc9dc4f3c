"""The vectorised packer (fq_pack.cpp: 32 bases per step) against a plain restatement of the packed-batch definition
(include/fastquick_amd.h: head = the filter's three 32-mers, src/BwtIndexer.cpp:441-456; body 2 bits per base; exception list).
Runs on the host-loop library: the packer is host code, the same source in both builds."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libfq_emu.so")


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libfq_emu.so"])
    from fastquick_amd import api
    return api.load_library(EMU)


def nt4(ch):
    return {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3, 45: 5}.get(int(ch), 4)


def restate(seq, lens, stride):
    """seq [2n][stride] uint8 -> head [3][2n] uint64, body rows, exceptions"""
    n2 = seq.shape[0]
    max_len = int(lens.max()) if n2 else 0
    bstride = max(8, (((max_len + 3) // 4) + 7) & ~7)
    head = np.zeros((3, n2), dtype=np.uint64)
    body = np.zeros((n2, bstride), dtype=np.uint8)
    exc = []
    for r in range(n2):
        L = int(lens[r])
        row = seq[r]
        for ch in range(3):
            k = 0
            for j in range(32):
                p = 32 * ch + j
                c = row[p] if (p < L or (p < stride and row[p])) else 65
                k = ((k << 2) | nt4(c)) & 0xFFFFFFFFFFFFFFFF
            head[ch, r] = k
        for i in range(L):
            c = nt4(row[i])
            if c < 4:
                body[r, i >> 2] |= c << (2 * (i & 3))
            else:
                exc.append((r << 32) | (i << 8) | c)
    return head, body, np.array(exc, dtype=np.uint64), bstride


def make_rows(rng, n_pairs, stride, lens, weird):
    seq = np.zeros((2, n_pairs, stride), dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    for e in range(2):
        for r in range(n_pairs):
            L = lens[e, r]
            seq[e, r, :L] = letters[rng.integers(0, 4, L)]
            if weird:
                for _ in range(rng.integers(0, 4)):
                    seq[e, r, rng.integers(0, L)] = rng.choice(np.frombuffer(b"NnacgtRY-.*", dtype=np.uint8))
                if L < min(stride, 96) and rng.random() < 0.5:    # what an earlier, longer read of the slot left behind a short one
                    seq[e, r, L:min(stride, 96)] = letters[rng.integers(0, 4, min(stride, 96) - L)]
    return seq


@pytest.mark.parametrize("case", ["uniform150", "uniform76", "ragged", "weird", "short", "len250"])
def test_packer_matches_definition(lib, case):
    from fastquick_amd import api
    rng = np.random.default_rng(sum(case.encode()))
    n_pairs = 300
    if case == "uniform150":
        stride, lens, weird = 160, np.full((2, n_pairs), 150, dtype=np.int32), False
    elif case == "uniform76":
        stride, lens, weird = 160, np.full((2, n_pairs), 76, dtype=np.int32), False
    elif case == "len250":
        stride, lens, weird = 256, np.full((2, n_pairs), 250, dtype=np.int32), True
    elif case == "ragged":
        stride, lens, weird = 160, rng.integers(15, 152, (2, n_pairs)).astype(np.int32), False
    elif case == "short":
        stride, lens, weird = 48, rng.integers(15, 49, (2, n_pairs)).astype(np.int32), True
    else:
        stride, lens, weird = 176, rng.integers(15, 177, (2, n_pairs)).astype(np.int32), True
    seq = make_rows(rng, n_pairs, stride, lens, weird)
    qual = rng.integers(33, 74, seq.shape).astype(np.uint8)
    hp = api.HostPacked(seq, qual, lens, None, threads=1, lib=lib)
    for round_ in range(2):     # the second round packs into the same storage (fq_pack_reads_into)
        b = hp.p.contents
        n2 = 2 * n_pairs
        head, body, exc, bstride = restate(seq.reshape(n2, stride), lens.reshape(n2), stride)
        assert b.n_pairs == n_pairs and b.body_stride == bstride
        uni = int(lens.max()) if lens.max() == lens.min() else 0
        assert b.uniform_len == uni
        got_head = np.ctypeslib.as_array(C.cast(b.head, C.POINTER(C.c_uint64)), (3, n2))
        got_body = np.ctypeslib.as_array(C.cast(b.body, C.POINTER(C.c_uint8)), (n2, bstride))
        assert np.array_equal(got_head, head)
        assert np.array_equal(got_body, body)
        assert b.n_exc == len(exc)
        if len(exc):
            assert np.array_equal(np.ctypeslib.as_array(C.cast(b.exc, C.POINTER(C.c_uint64)), (len(exc),)), exc)
        if not uni:
            assert np.array_equal(np.ctypeslib.as_array(C.cast(b.len, C.POINTER(C.c_uint16)), (n2,)), lens.reshape(n2).astype(np.uint16))
        ql = np.ctypeslib.as_array(C.cast(b.qual_last, C.POINTER(C.c_uint8)), (n2,))
        assert np.array_equal(ql, qual.reshape(n2, stride)[np.arange(n2), lens.reshape(n2) - 1])
        serial = b.serial
        assert serial != 0
        if round_ == 0:
            hp.repack(threads=2)
            assert hp.p.contents.serial != serial
    hp.free()
