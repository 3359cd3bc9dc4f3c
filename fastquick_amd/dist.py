"""Multi-GPU plumbing: one process per GPU, each rank aligns its own FASTQ stream(s); no data-path collective.

The reference treats every `--fq_list` line as an independent stream (own srand48(11), own last_ii chain,
src/BwtMapper.cpp:1817), so streams shard across ranks without any exchange.  What is exchanged is bookkeeping only:
a barrier, the MAX of the elapsed time (bench contract), the SUM of the FileStatCollector-style counters, and -- for a
caller that wants one output file -- a gather of per-rank SAM buffers to rank 0 in rank order.
Works with backend "nccl" (RCCL, GPU tensors) and "gloo" (CPU tensors; used by the world_size-2 CPU tests)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def streams_for_rank(n_streams: int, rank: int, world: int) -> list:
    """Contiguous block partition of stream indices (rank order == output order when gathered)."""
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return list(range(lo, hi))


def init(backend: str | None = None) -> tuple:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def _dev():
    return "cuda" if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu"


def world_size() -> int:
    """Ranks the collective backend (RCCL / gloo) actually sees; 1 without a process group."""
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_counters(counters: dict) -> dict:
    """Element-wise SUM of integer counters over ranks (NumBase, TotalFiltered, BwaUnmapped, pairs, ...)."""
    if not dist.is_initialized():
        return dict(counters)
    keys = sorted(counters)
    t = torch.tensor([int(counters[k]) for k in keys], dtype=torch.int64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def gather_bytes_to_rank0(buf: bytes):
    """Variable-length gather of per-rank byte buffers; rank 0 gets them in rank order, others None."""
    if not dist.is_initialized():
        return [buf]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _dev()
    n = torch.tensor([len(buf)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])
    mine = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if buf:
        mine[:len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
    parts = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return None
    return [bytes(p[:s].cpu().numpy().tobytes()) for p, s in zip(parts, sizes)]
