#!/usr/bin/env python3
"""The device front end alone over a pair of BGZF FASTQ files (fq_frontend_*; batches released as they come): pairs/s and where the time went.
    python tools/frontend_stream.py <reads_1.fq.gz> <reads_2.fq.gz> [--chunk-pairs N] [--repeats R]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastquick_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fq1")
    ap.add_argument("fq2")
    ap.add_argument("--chunk-pairs", type=int, default=16 * 262144)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--max-read-len", type=int, default=160)
    a = ap.parse_args()
    api.load_library().fq_runtime_configure(20, 1)     # (as the command line does: the producer's two streams, the readers' and the copy streams on hardware queues of their own)
    for _ in range(a.repeats):
        t_open = time.perf_counter()
        dfe = api.DeviceFrontEnd(a.fq1, a.fq2, batch_pairs=262144, chunk_pairs=a.chunk_pairs, slot_mode=0, max_read_len=a.max_read_len)
        t0 = time.perf_counter()
        pairs = 0
        t_first = None
        while True:
            m, b = dfe.next()
            if m <= 0:
                break
            if t_first is None:
                t_first = time.perf_counter() - t0
            pairs += m
            dfe.release(b)
        dt = time.perf_counter() - t0
        st = dfe.stats()
        dfe.close()
        print(json.dumps({"pairs": pairs, "rc": m, "s": round(dt, 4), "open_s": round(t0 - t_open, 4), "first_batch_s": round(t_first or 0, 4), "pairs_per_s": round(pairs / dt, 1),
                          "stats": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}}), flush=True)


if __name__ == "__main__":
    main()
