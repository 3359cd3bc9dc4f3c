#!/usr/bin/env python3
"""Measures the device front end's kernels on their own (one MI355X): BGZF members inflated in HBM (`k_inflate_bgzf`), at the compression
levels FASTQ files come in (bgzip's default is zlib level 6; bench.py's synthetic files are level 1).
    python tools/frontend_bench.py [--records N] [--levels 1,6] [--repeats R]
Prints one JSON line per level: text GB/s, members, kernel ms."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastquick_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=2_000_000)
    ap.add_argument("--levels", default="1,6")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--names", choices=("plain", "illumina"), default="plain", help="illumina: names with varying coordinates and a comment (the matches of such text reach further back)")
    a = ap.parse_args()
    rng = np.random.default_rng(99)
    n, L = a.records, a.read_len
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, L), dtype=np.uint8)]
    qual = np.frombuffer(b"F:,#", dtype=np.uint8)[rng.choice(4, size=(n, L), p=[0.7, 0.15, 0.1, 0.05])]
    if a.names == "illumina":
        xs, ys = rng.integers(1000, 40000, n), rng.integers(1000, 200000, n)
        text = b"".join(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT\n" % (1101 + i % 1000, xs[i], ys[i]) + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(n))
        nbytes = len(text)
    else:
        path = "/tmp/frontend_bench.fq"
        nbytes = synth.write_fastq_uniform(seq, qual, L, path, bgzf=False)
        text = open(path, "rb").read()
        os.remove(path)
    for level in [int(x) for x in a.levels.split(",")]:
        t0 = time.perf_counter()
        blob = synth.bgzf_compress(text, threads=16, level=level)
        tc = time.perf_counter() - t0
        out, status, ms = api.bgzf_inflate_device(blob, nbytes + 64, repeats=a.repeats)
        ok = out.tobytes() == text
        print(json.dumps({"kernel": "k_inflate_bgzf", "zlib_level": level, "records": n, "text_bytes": nbytes, "file_bytes": len(blob), "members": len(status),
                          "refused": int(sum(1 for s in status if s)), "identical_to_text": ok, "kernel_ms": round(ms, 3),
                          "text_GBps": round(nbytes / ms / 1e6, 2), "bytes_in_plus_out_GBps": round((nbytes + len(blob)) / ms / 1e6, 2), "compress_s": round(tc, 1)}), flush=True)


if __name__ == "__main__":
    main()
