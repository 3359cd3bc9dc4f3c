"""Multi-GPU plumbing: one process per GPU; no data-path collective.

Two partitionings: (1) independent FASTQ streams per rank -- below; (2) ONE stream sharded by reference
batch (StreamShard): rank r aligns batches r, r+W, ...; the order-dependent state of the stream (drand48, last_ii, (k,l) cache: a
few dozen bytes) travels from the owner of batch b to the owner of batch b+1 by point-to-point send/recv around the short serial
part of each call, and rank 0 gathers the per-batch outputs in batch order.

(1): each rank aligns its own FASTQ stream(s).

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


_token_group = None


def token_group():
    """The group the stream tokens of StreamShard travel over: host buffers of a few hundred bytes, sent from inside a library
    call's hook.  Under RCCL they go over a gloo group of the same ranks -- plain sockets, no device tensors, nothing enqueued on
    a GPU stream that a kernel of the sending call could be ordered behind -- and under gloo over the default group."""
    global _token_group
    if dist.is_initialized() and dist.get_backend() == "nccl" and _token_group is None:
        import datetime
        # a lost token raises instead of waiting for ever (and the stream is then marked broken: api.Aligner.set_serial_hooks); the wait
        # must outlast a predecessor that is legitimately slow -- its first call stages the index -- so it is generous and configurable
        secs = float(os.environ.get("FASTQUICK_TOKEN_TIMEOUT_S", "1800"))
        _token_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=secs))
    return _token_group


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


def _send_bytes(blob: bytes, dst: int) -> None:
    g = token_group()
    dist.send(torch.tensor([len(blob)], dtype=torch.int64), dst, group=g)
    if blob:
        dist.send(torch.frombuffer(bytearray(blob), dtype=torch.uint8), dst, group=g)


def _recv_bytes(src: int) -> bytes:
    g = token_group()
    n = torch.zeros(1, dtype=torch.int64)
    dist.recv(n, src, group=g)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8)
    if buf.numel():
        dist.recv(buf, src, group=g)
    return bytes(buf.numpy().tobytes())


def gather_keyed_to_rank0(items):
    """items: this rank's [(key, bytes)]; rank 0 returns all ranks' items sorted by key, the others None."""
    blob = b"".join(len(p).to_bytes(8, "little") + int(k).to_bytes(8, "little") + p for k, p in items)
    got = gather_bytes_to_rank0(blob)
    if got is None:
        return None
    out = []
    for g in got:
        at = 0
        while at < len(g):
            ln, k = int.from_bytes(g[at:at + 8], "little"), int.from_bytes(g[at + 8:at + 16], "little")
            out.append((k, g[at + 16:at + 16 + ln]))
            at += 16 + ln
    return sorted(out, key=lambda t: t[0])


def merge_qc_on_rank0(segments, root_qc) -> None:
    """segments: this rank's [(order key, exported consumer state)] (QC.state_export of a shard consumer, per reference batch or per
    FASTQ pair).  Rank 0 merges every rank's segments into root_qc in key order -- the order of the input -- so that root_qc.write()
    gives the files of the single-process run (fq_qc_merge).  A trivial gather: the states are sums and logs, no rank needs another's."""
    got = gather_keyed_to_rank0(segments)
    if got is None:
        return
    for _k, blob in got:
        root_qc.merge(blob)


class StreamShard:
    """ONE FASTQ stream over `world` ranks, sharded by reference batch.  Rank r owns batches r, r + world, ...; every
    rank drives its own Aligner.  The hooks the library calls around the order-dependent part of a call receive the stream state
    from the owner of the previous batch and pass it on to the owner of the next (fq_ctx_set_serial_hooks /
    fq_ctx_state_export / _import), so that the outputs are those of the single sequential stream -- the reference's."""

    def __init__(self, aligner, rank: int, world: int):
        self.al, self.rank, self.world = aligner, rank, world
        self.batch_index = 0
        self.n_batches = 0
        self.qc_segments = []
        token_group()            # (collective: created by every rank at the same point)
        aligner.set_serial_hooks(self._before, self._after)

    def _before(self):
        if self.world > 1 and self.batch_index > 0:
            self.al.import_state(_recv_bytes((self.batch_index - 1) % self.world))

    def _after(self):
        if self.world > 1 and self.batch_index + 1 < self.n_batches:
            _send_bytes(self.al.export_state(), (self.batch_index + 1) % self.world)

    def owns(self, b: int) -> bool:
        return b % self.world == self.rank

    def align_stream(self, names, seq, qual, lens, batch: int, packed: bool = False, want_sam: bool = True, qc=None):
        """Aligns this rank's batches of the stream; returns [(batch index, SAM text of the batch)] (header not included)."""
        from . import api
        n = seq.shape[1]
        self.n_batches = (n + batch - 1) // batch
        out = []
        for b in range(self.n_batches):
            if not self.owns(b):
                continue
            self.batch_index = b
            lo, hi = b * batch, min(n, (b + 1) * batch)
            if packed:
                hp = api.HostPacked(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi], lib=self.al.L)
                self.al.align_packed(hp)
            else:
                self.al.align(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi])
            if qc is not None:      # a shard consumer (QC.state_reset called): one exported segment per batch, merged in batch order later
                qc.add(self.al)
                self.qc_segments.append((b, qc.state_export()))
                qc.state_reset()
            out.append((b, self.al.sam_text() if want_sam else b""))
            if packed:
                self.al._keep_packed = None
                hp.free()
        return out

    def gather_in_batch_order(self, parts):
        """parts: this rank's [(batch index, bytes)]; rank 0 returns the stream's bytes in batch order, others None."""
        items = gather_keyed_to_rank0(parts)
        return None if items is None else b"".join(p for _b, p in items)
