#!/usr/bin/env python3
"""Soak of the FASTQ front end on the device against the host path, on random FASTQ pairs:

  * random read lengths (uniform or ragged, short reads among long ones), names of random lengths with and without comments and /1 /2 suffixes,
    qualities over the whole printable range ('@' and '+' first included), BGZF members of random sizes at random zlib levels;
  * now and then an odd record at a random place (a wrapped base line, a blank line, a quality string of another length, a missing last line end,
    a name of 400 characters): the device's part must end at the boundary of the reference batch that holds it, and the host readers handed over
    there must return the rest;
  * filter keys, lengths and names of every read from fq_frontend_* (+ fq_frontend_handover) == fq_fastq_read + fq_pack_reads_into on the same files,
    for the three read-slot modes;
  * raw DEFLATE streams of random data kinds, levels, strategies and flush points through the member decoder == zlib (fq_inflate_device).

    python tests/fuzz_frontend.py --seeds 200 [--start N] [--emu]      (GPU by default; --emu: the host-loop library, CPU)"""
import argparse
import os
import sys
import tempfile
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from fastquick_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=50)
ap.add_argument("--start", type=int, default=0)
ap.add_argument("--emu", action="store_true")
a = ap.parse_args()
if a.emu:
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu"), "libfq_emu.so"])
    lib = api.load_library(os.path.join(HERE, "emu", "libfq_emu.so"))
else:
    lib = api.load_library()
import ctypes as C  # noqa: E402


def make_pair(rng, n, odd_at):
    ragged = rng.random() < 0.5
    base_len = int(rng.choice([36, 76, 100, 150, 151, 250]))
    texts = [[], []]
    for i in range(n):
        L = int(rng.integers(max(20, base_len - 60), base_len + 1)) if ragged and rng.random() < 0.4 else base_len
        nm = b"r%d" % i + (b":" + bytes(rng.integers(65, 91, int(rng.integers(0, 30))).astype(np.uint8)) if rng.random() < 0.5 else b"")
        for e in range(2):
            seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn", dtype=np.uint8), L, p=[0.24, 0.24, 0.24, 0.24, 0.02, 0.005, 0.005, 0.005, 0.004, 0.001]))
            qual = bytes(rng.integers(33, 127, L).astype(np.uint8))
            name = nm + (b"/%d" % (e + 1) if i % 7 == 0 else b"") + (b" comment x" if i % 5 == 0 else b"")
            rec = [b"@" + name, seq, b"+" + (nm if i % 11 == 0 else b""), qual]
            if i == odd_at and e == 0:
                kind = int(rng.integers(0, 4))
                if kind == 0:
                    rec[1] = seq[:L // 2] + b"\n" + seq[L // 2:]            # a wrapped base line
                elif kind == 1:
                    rec = [b""] + rec                                       # a blank line in front
                elif kind == 2:
                    rec[0] = b"@" + b"x" * 400                              # a name longer than the reference's buffer
                else:
                    rec[3] = qual[:-1] if L > 21 else qual + b"I"           # a quality string of another length: the reference refuses the file
            texts[e].append(b"\n".join(rec))
    return [b"\n".join(t) + b"\n" for t in texts]


def host_path(fq, B, slot_mode):
    files = [api.FastqFile(p, threads=2, batch_pairs=B, slot_mode=slot_mode, stride=256, name_stride=304, lib=lib) for p in fq]
    rows, err = [], None
    try:
        rows = [f.read(1 << 20) for f in files]
    except api.FastquickError as e:
        err = str(e)
    for f in files:
        f.close()
    if err:
        return None, err
    n = min(len(r[2]) for r in rows)
    seq = np.stack([r[0][:n] for r in rows]); qual = np.stack([r[1][:n] for r in rows]); lens = np.stack([r[2][:n] for r in rows])
    hp = api.HostPacked(seq, qual, lens, None, lib=lib)
    head = np.ctypeslib.as_array(C.cast(hp.p.contents.head, C.POINTER(C.c_uint64)), shape=(3 * 2 * n,)).reshape(3, 2 * n).copy()
    hp.free()
    return (n, head, lens.reshape(-1).astype(np.uint16), np.concatenate([r[3][:n] for r in rows])), None


def device_path(fq, B, chunk, slot_mode):
    fe = api.DeviceFrontEnd(fq[0], fq[1], batch_pairs=B, chunk_pairs=chunk, slot_mode=slot_mode, max_read_len=256, lib=lib)
    heads, lens, names, total = [[], []], [[], []], [[], []], 0
    err = None
    try:
        while True:
            n, b = fe.next()
            if n <= 0:
                break
            h, l, nm = fe.fetch(b, n)
            for e in range(2):
                heads[e].append(h[:, e * n:(e + 1) * n]); lens[e].append(l[e * n:(e + 1) * n])
                full = np.zeros((n, 304), dtype=np.uint8); full[:, :nm.shape[1]] = nm[e * n:(e + 1) * n]
                names[e].append(full)
            fe.release(b)
            total += n
        if n == api.FQ_EFALLBACK:
            readers = fe.handover(threads=2, stride=256, name_stride=304)
            try:
                rest = [r.read(1 << 20) for r in readers]
            finally:
                for r in readers:
                    r.close()
            m = min(len(r[2]) for r in rest)
            if m:
                hp = api.HostPacked(np.stack([r[0][:m] for r in rest]), np.stack([r[1][:m] for r in rest]), np.stack([r[2][:m] for r in rest]), None, lib=lib)
                hh = np.ctypeslib.as_array(C.cast(hp.p.contents.head, C.POINTER(C.c_uint64)), shape=(3 * 2 * m,)).reshape(3, -1).copy()
                hp.free()
                for e in range(2):
                    heads[e].append(hh[:, e * m:(e + 1) * m]); lens[e].append(rest[e][2][:m].astype(np.uint16)); names[e].append(rest[e][3][:m])
                total += m
    except api.FastquickError as e:
        err = str(e)
    fe.close()
    if err:
        return None, err
    if total == 0:
        return (0, np.zeros((3, 0), np.uint64), np.zeros(0, np.uint16), np.zeros((0, 304), np.uint8)), None
    head = np.concatenate([np.concatenate(heads[e], axis=1) for e in range(2)], axis=1)
    return (total, head, np.concatenate([np.concatenate(lens[e]) for e in range(2)]), np.concatenate([np.concatenate(names[e]) for e in range(2)])), None


def decoder_case(rng):
    n = int(rng.integers(1, 60000))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        data = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
    elif kind == 1:
        data = bytes(rng.integers(0, 256, n).astype(np.uint8))
    elif kind == 2:
        data = (b"@r0001\nACGTTGCA\n+\nFFFF:,#F\n" * (n // 26 + 1))[:n]
    elif kind == 3:
        data = bytes(rng.choice(np.frombuffer(b"AB", dtype=np.uint8), n, p=[0.95, 0.05]))
    else:
        data = bytes(rng.integers(97, 104, n).astype(np.uint8))
    level = int(rng.integers(0, 10))
    strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
    comp = b""
    if rng.random() < 0.3 and n > 500:
        step = int(rng.integers(100, 5000))
        for p in range(0, n, step):
            comp += co.compress(data[p:p + step]) + co.flush(zlib.Z_SYNC_FLUSH if rng.random() < 0.5 else zlib.Z_FULL_FLUSH)
        comp += co.flush()
    else:
        comp = co.compress(data) + co.flush()
    return data, comp


ok = fail = 0
t_all = time.time()
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(a.start, a.start + a.seeds):
        rng = np.random.default_rng(seed)
        t0 = time.time()
        # ---- the member decoder on its own
        cases = [decoder_case(rng) for _ in range(int(rng.integers(1, 12)))]
        res, _ = api.inflate_device([(c, len(d), zlib.crc32(d)) for d, c in cases], lib=lib)
        bad = [k for k, (d, _) in enumerate(cases) if res[k][0] != 0 or res[k][1] != d]
        # ---- the front end against the host path
        B = int(rng.choice([32, 64, 128]))
        n = int(rng.integers(1, 9 * B))
        odd_at = int(rng.integers(0, n)) if rng.random() < 0.35 else -1
        texts = make_pair(rng, n, odd_at)
        if rng.random() < 0.15:
            texts[0] = texts[0][:-1]                                        # no line end behind the last record of file 1
        fq = []
        for e in range(2):
            path = os.path.join(tmp, "f%d_%d.fq.gz" % (seed, e))
            with open(path, "wb") as fh:
                fh.write(synth.bgzf_compress(texts[e], threads=2, level=int(rng.integers(1, 10)), member=int(rng.integers(300, 60000))))
            fq.append(path)
        slot_mode = int(rng.integers(0, 3))
        chunk = int(rng.integers(1, 4)) * B
        hres, herr = host_path(fq, B, slot_mode)
        dres, derr = device_path(fq, B, chunk, slot_mode)
        why = None
        if bad:
            why = "decoder: streams %s differ from zlib" % bad
        elif (herr is None) != (derr is None):
            why = "host %r / device %r" % (herr, derr)
        elif herr is None:
            if hres[0] != dres[0]:
                why = "records: host %d device %d" % (hres[0], dres[0])
            elif not (np.array_equal(hres[1], dres[1]) and np.array_equal(hres[2], dres[2]) and np.array_equal(hres[3], dres[3])):
                why = "keys / lengths / names differ"
        elif herr.split("(")[-1] != derr.split("(")[-1]:
            why = "messages differ: host %r device %r" % (herr, derr)
        for p in fq:
            os.remove(p)
        tag = "seed %6d B %3d n %5d chunk %4d slots %d odd %5d %s" % (seed, B, n, chunk, slot_mode, odd_at, "refused" if herr else "       ")
        if why:
            fail += 1
            print(tag, "FAIL", why, flush=True)
        else:
            ok += 1
            print(tag, " OK  %.1fs" % (time.time() - t0), flush=True)
print("%d cases, %d mismatches, %.0f s" % (ok + fail, fail, time.time() - t_all))
sys.exit(1 if fail else 0)
