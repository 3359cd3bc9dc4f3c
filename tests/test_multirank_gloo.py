"""world_size-2 CPU test (gloo) of the N>1 path: each rank aligns its own stream (host-loop backend), counters are
summed, SAM buffers gathered to rank 0 in rank order, elapsed time MAX-reduced -- the same helpers bench.py uses with RCCL."""
import gzip
import os
import subprocess
import sys

import pytest

import golden_util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import golden_util, oracle_binding as ob
from fastquick_amd import api, dist as fqd
rank, local_rank, world = fqd.init("gloo")
assert world == 2
tags = ["basic", "repeat", "isize"]
mine = fqd.streams_for_rank(len(tags), rank, world)
L = api.load_library(os.path.join(%(here)r, "emu", "libfq_emu.so"))
out = b""
cnt = {"pairs": 0, "records": 0, "filtered": 0}
t0 = time.perf_counter()
for si in mine:
    g = golden_util.materialise(tags[si], os.path.join(%(tmp)r, "r%%d_%%s" %% (rank, tags[si])))
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=L)
    al = api.Aligner(ix, api.default_opts(L, trim_qual=g["trim_qual"], batch_pairs=g["batch"]), max_pairs=seq.shape[1])
    res = al.align(seq, qual, lens, names)          # one call, several reference batches
    out += al.sam_text()
    cnt["pairs"] += res.n_pairs; cnt["records"] += res.n_survivors - res.n_both_unmapped; cnt["filtered"] += res.n_both_filtered
    al.close(); ix.close()
elapsed = fqd.max_over_ranks(time.perf_counter() - t0)
tot = fqd.sum_counters(cnt)
parts = fqd.gather_bytes_to_rank0(out)
fqd.barrier()
if rank == 0:
    open(os.path.join(%(tmp)r, "gathered.sam"), "wb").write(b"".join(parts))
    open(os.path.join(%(tmp)r, "totals.txt"), "w").write("%%d %%d %%d %%.3f" %% (tot["pairs"], tot["records"], tot["filtered"], elapsed))
'''


def test_two_ranks_gloo(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, here=HERE, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)], env=env, timeout=600)
    want = b""
    n_pairs = 0
    for tag in ["basic", "repeat", "isize"]:
        with gzip.open(os.path.join(golden_util.GOLD, tag, "ref.sam.gz"), "rb") as fh:
            want += b"".join(l for l in fh.read().splitlines(keepends=True) if not l.startswith(b"@"))
        n_pairs += golden_util.case_params(tag)["n_pairs"]
    got = (tmp_path / "gathered.sam").read_bytes()
    assert got == want, "rank-ordered gather of per-rank SAM must equal the reference's records for the same streams"
    tot = (tmp_path / "totals.txt").read_text().split()
    assert int(tot[0]) == n_pairs and int(tot[1]) == want.count(b"\n") // 2


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import golden_util, oracle_binding as ob
from fastquick_amd import api, dist as fqd
rank, local_rank, world = fqd.init("gloo")
assert world == 2
L = api.load_library(os.path.join(%(here)r, "emu", "libfq_emu.so"))
for tag in ["isize", "basic", "qc", "edge"]:
    g = golden_util.materialise(tag, os.path.join(%(tmp)r, "s%%d_%%s" %% (rank, tag)))
    names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
    ix = api.Index(g["prefix"], lib=L)
    al = api.Aligner(ix, api.default_opts(L, trim_qual=g["trim_qual"]), max_pairs=max(16, g["batch"]))
    sh = fqd.StreamShard(al, rank, world)
    parts = sh.align_stream(names, seq, qual, lens, g["batch"], packed=(tag == "qc"))
    assert len(parts) == len([b for b in range(sh.n_batches) if b %% world == rank])
    sam = sh.gather_in_batch_order(parts)
    if rank == 0:
        open(os.path.join(%(tmp)r, "shard_%%s.sam" %% tag), "wb").write(sam)
    fqd.barrier()
    al.close(); ix.close()
'''


def test_one_stream_sharded_over_two_ranks(tmp_path):
    """ONE FASTQ stream split by reference batch over two ranks (batches 0, 2, ... on rank 0; 1, 3, ... on rank 1), the drand48
    state, the last_ii fallback and the (k,l) cache handed from batch to batch by send/recv: the SAM text gathered in batch order
    must be the reference's for the sequential stream.  `isize` has a failed insert-size inference that falls back on the previous
    batch's (which lives on the other rank); `basic` and `qc` have three batches (drand48 crosses ranks twice)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % dict(root=ROOT, here=HERE, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env, timeout=900)
    for tag in ["isize", "basic", "qc", "edge"]:
        with gzip.open(os.path.join(golden_util.GOLD, tag, "ref.sam.gz"), "rb") as fh:
            want = b"".join(l for l in fh.read().splitlines(keepends=True) if not l.startswith(b"@"))
        got = (tmp_path / ("shard_%s.sam" % tag)).read_bytes()
        assert got == want, tag
