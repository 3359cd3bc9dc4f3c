#!/usr/bin/env python3
"""Soak of the FASTQ reader's gzip stream decoder (fq_fastq.cpp: fill_gz_stream, fq_inflate.h's decoder block after block) against zlib's gzread
(FASTQUICK_ZLIB_INFLATE=1) on random files: FASTQ text of random shapes at random levels / flush points / member cuts / header fields, intact or
damaged (bit flips, truncation anywhere, bytes appended, a wrong trailer) -- how the read ended (end of file, or the error message) must be the
same, and so must the records; where the read ends in an error one reader may have returned up to a block's worth of records more than the
other, all of them the same records (gzread reads up to 128 KiB ahead behind a member's header, so it meets a bad trailer that much
earlier than the block it lies in).  Host code only: runs on the CPU.
    python tests/fuzz_gzip_stream.py --seeds 500 [--start N]"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from fastquick_amd import api  # noqa: E402
from test_fastq_reader import _gz_member, _outcome, make_fastq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=100)
ap.add_argument("--start", type=int, default=0)
a = ap.parse_args()
subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu"), "libfq_emu.so"])
lib = api.load_library(os.environ.get("FQ_EMU_LIB") or os.path.join(HERE, "emu", "libfq_emu.so"))
ok = fail = 0
t_all = time.time()
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(a.start, a.start + a.seeds):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 6000))
        _, text = make_fastq(rng, n, ragged=bool(rng.random() < 0.5))
        cuts = sorted({0, len(text)} | {int(x) for x in rng.integers(0, len(text) + 1, int(rng.integers(0, 4)))})
        blob = b""
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            blob += _gz_member(text[lo:hi], int(rng.integers(0, 10)), flags=int(rng.choice([0, 0, 8, 4 | 8, 2 | 16, 4 | 8 | 16 | 2])),
                               flush_every=int(rng.integers(200, 90000)) if rng.random() < 0.3 else 0)
        kind = int(rng.integers(0, 6))
        if kind == 1 and len(blob) > 30:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, len(blob)))
                blob = blob[:at] + bytes([blob[at] ^ (1 << int(rng.integers(0, 8)))]) + blob[at + 1:]
        elif kind == 2:
            blob = blob[:int(rng.integers(0, len(blob) + 1))]
        elif kind == 3:
            blob += bytes(rng.integers(0, 256, int(rng.integers(1, 200))).astype(np.uint8))
        elif kind == 4 and len(blob) > 8:
            at = len(blob) - int(rng.integers(1, 9))
            blob = blob[:at] + bytes([blob[at] ^ 0x10]) + blob[at + 1:]
        path = os.path.join(tmp, "f%d.fq.gz" % seed)
        with open(path, "wb") as fh:
            fh.write(blob)
        block = int(rng.choice([0, 1 << 18, 1 << 20]))
        threads = int(rng.integers(1, 5))
        os.environ["FASTQUICK_ZLIB_INFLATE"] = "1"
        try:
            ref = _outcome(api, lib, path, threads, block)
        except api.FastquickError as e:
            ref = ("open failed", str(e))
        os.environ.pop("FASTQUICK_ZLIB_INFLATE")
        try:
            got = _outcome(api, lib, path, threads, block)
        except api.FastquickError as e:
            got = ("open failed", str(e))
        os.remove(path)
        tag = "seed %6d n %5d members %d kind %d bytes %8d" % (seed, n, len(cuts) - 1, kind, len(blob))
        same = got == ref
        # (... and the text one reader alone still returns may hold the record the damage made of a good one: the tokeniser's refusal then stands where
        #  the other reader reports the check sum)
        if not same and isinstance(got[0], list) and isinstance(ref[0], list) and got[1][0] == "error" and ref[1][0] == "error":
            a_, b_ = (got[0], ref[0]) if len(got[0]) <= len(ref[0]) else (ref[0], got[0])
            same = b_[:len(a_)] == a_ and sum(len(x[1]) * 2 + len(x[0]) + 6 for x in b_[len(a_):]) <= max(block, 192 << 10) + (5000 * 400)
        if not same:
            fail += 1
            print(tag, "FAIL", got[1] if isinstance(got[0], list) else got, "|", ref[1] if isinstance(ref[0], list) else ref, flush=True)
        else:
            ok += 1
            print(tag, " OK ", ("%d records, %s" % (len(got[0]), got[1][0])) if isinstance(got[0], list) else got[0], flush=True)
print("%d cases, %d mismatches, %.0f s" % (ok + fail, fail, time.time() - t_all))
sys.exit(1 if fail else 0)
