#!/usr/bin/env python3
"""What this box's PCIe link delivers host -> device from pinned memory (tools; not on the product path):
one stream, several concurrent streams, and with a memory-bound kernel running beside the copies."""
import time
import torch

dev = torch.device("cuda:0")
MB = 201
src = [torch.empty(MB << 20, dtype=torch.uint8).pin_memory() for _ in range(4)]
dst = [torch.empty(MB << 20, dtype=torch.uint8, device=dev) for _ in range(16)]
big = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
idx = torch.randint(0, 3 << 30, (1 << 26,), device=dev)


def run(n_streams, reps, with_kernel):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    ks = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for s in range(n_streams):
            with torch.cuda.stream(streams[s]):
                dst[s].copy_(src[(s + r) % 4], non_blocking=True)
        if with_kernel:
            with torch.cuda.stream(ks):
                for _ in range(4):
                    big[idx].sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n_streams * reps * MB * (1 << 20) / dt / 1e9


for n in (1, 2, 4, 16):
    for wk in (False, True):
        run(n, 1, wk)
        print("streams %2d kernel_beside %d : %.1f GB/s" % (n, wk, run(n, 8 if n < 16 else 3, wk)), flush=True)
