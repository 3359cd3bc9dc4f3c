#!/usr/bin/env python3
"""The device front end on input that is NOT the bench's uniform synthetic text: reads trimmed to 60..150 bases, Illumina-style names of varying
length with a comment, qualities over 40 values -- a million pairs, concatenated `--copies` times: pairs/s and device ms per kernel group (the
read-slot kernels take their general paths: bases behind short reads, names of unequal lengths), and the same records from the host reader.
    python tools/frontend_ragged_check.py [--pairs N] [--copies C]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastquick_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1 << 20)
ap.add_argument("--copies", type=int, default=8)
ap.add_argument("--workdir", default="/tmp/fq_ragged")
ap.add_argument("--uniform", action="store_true", help="every read 150 bases (names still vary)")
ap.add_argument("--no-host", action="store_true")
a = ap.parse_args()
api.load_library().fq_runtime_configure(20, 1)
os.makedirs(a.workdir, exist_ok=True)
rng = np.random.default_rng(31)
n = a.pairs
t0 = time.time()
lens = [np.where(rng.random(n) < (0.0 if a.uniform else 0.3), rng.integers(60, 151, n), 150) for _ in range(2)]
xs, ys = rng.integers(1000, 40000, n), rng.integers(1000, 200000, n)
paths = []
for e in range(2):
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150), dtype=np.uint8)]
    qual = rng.integers(35, 75, (n, 150), dtype=np.uint8)
    recs = []
    for i in range(n):
        L = int(lens[e][i])
        recs.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d %d:N:0:ACGTACGT\n" % (1101 + i % 1000, xs[i], ys[i], e + 1) + seq[i, :L].tobytes() + b"\n+\n" + qual[i, :L].tobytes() + b"\n")
    text = b"".join(recs)
    path = os.path.join(a.workdir, "ragged_%d.fq.gz" % (e + 1))
    body = synth.bgzf_compress(text, threads=16, level=6)
    eof = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"
    if body.endswith(eof):
        body = body[:-28]
    with open(path, "wb") as fo:
        for _ in range(a.copies):
            fo.write(body)
    paths.append(path)
print("files written in %.0f s: %d pairs x %d, %.2f GB each" % (time.time() - t0, n, a.copies, os.path.getsize(paths[0]) / 1e9), flush=True)
for rep in range(2):
    dfe = api.DeviceFrontEnd(paths[0], paths[1], batch_pairs=262144, chunk_pairs=16 * 262144, slot_mode=0, max_read_len=160)
    t0 = time.perf_counter()
    got = 0
    while True:
        m, b = dfe.next()
        if m <= 0:
            break
        got += m
        dfe.release(b)
    dt = time.perf_counter() - t0
    st = dfe.stats()
    dfe.close()
    print(json.dumps({"device_front_end": {"pairs": got, "rc": m, "s": round(dt, 3), "pairs_per_s": round(got / dt, 1), "ms": {k: round(st[k], 1) for k in ("ms_inflate", "ms_lines", "ms_records", "ms_slots")},
                                            "refused": st["refused"]}}), flush=True)
if a.no_host:
    sys.exit(0)
# the host reader on the same files (8 threads per file)
import threading
rows = [(np.zeros((n * a.copies, 160), np.uint8), np.zeros((n * a.copies, 160), np.uint8), np.zeros(n * a.copies, np.int32), np.zeros((n * a.copies, 64), np.uint8)) for _ in range(2)]
res = [0, 0]


def rd(e):
    f = api.FastqFile(paths[e], threads=8, stride=160, name_stride=64, slot_mode=0)
    res[e] = f.read_into(*rows[e])
    f.close()
t0 = time.perf_counter()
th = [threading.Thread(target=rd, args=(e,)) for e in range(2)]
[t.start() for t in th]
[t.join() for t in th]
dt = time.perf_counter() - t0
print(json.dumps({"host_reader": {"pairs": res, "s": round(dt, 3), "pairs_per_s": round(min(res) / dt, 1)}}))
