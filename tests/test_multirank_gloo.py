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


QC_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import golden_util, oracle_binding as ob
from fastquick_amd import api, dist as fqd
rank, local_rank, world = fqd.init("gloo")
assert world == 2
L = api.load_library(os.path.join(%(here)r, "emu", "libfq_emu.so"))
g = golden_util.materialise("qc", os.path.join(%(tmp)r, "q%%d" %% rank))
kw = dict(genome_size=g["genome_size"], read_len=g["qc_read_len"])
ix = api.Index(g["prefix"], lib=L)
# ---- (1) the two lines of a --fq_list, one per rank: each rank's consumer holds one whole FASTQ pair
halves = golden_util.split_halves(g, g["dir"])
f1, f2 = halves[rank]
names, seq, qual, lens = ob.read_fastq_pair(f1, f2)
al = api.Aligner(ix, api.default_opts(L, trim_qual=g["trim_qual"]), max_pairs=max(16, g["batch"]))
q = api.QC(ix, g["prefix"], os.path.join(g["dir"], "mine"), **kw)
q.state_reset()
q.begin_file(f1, f2)
sam = b""
for lo in range(0, seq.shape[1], g["batch"]):
    hi = min(seq.shape[1], lo + g["batch"])
    al.align(seq[:, lo:hi], qual[:, lo:hi], lens[:, lo:hi], names[lo:hi])
    q.add(al)
    sam += al.sam_text()
q.end_file()
root = api.QC(ix, g["prefix"], os.path.join(%(tmp)r, "fqlist"), **kw) if rank == 0 else None
fqd.merge_qc_on_rank0([(rank, q.state_export())], root)
parts = fqd.gather_bytes_to_rank0(sam)
if rank == 0:
    root.write(); root.close()
    open(os.path.join(%(tmp)r, "fqlist.sam"), "wb").write(b"".join(parts))
q.close(); al.close()
fqd.barrier()
# ---- (2) ONE FASTQ pair sharded by reference batch: a segment per batch, merged in batch order
names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
al = api.Aligner(ix, api.default_opts(L, trim_qual=g["trim_qual"]), max_pairs=max(16, g["batch"]))
q = api.QC(ix, g["prefix"], os.path.join(g["dir"], "mine2"), **kw)
q.begin_file(g["fq1"], g["fq2"])
q.state_reset()
sh = fqd.StreamShard(al, rank, world)
sh.align_stream(names, seq, qual, lens, g["batch"], want_sam=False, qc=q)
root = None
if rank == 0:
    root = api.QC(ix, g["prefix"], os.path.join(%(tmp)r, "sharded"), **kw)
    root.begin_file(g["fq1"], g["fq2"])
fqd.merge_qc_on_rank0(sh.qc_segments, root)
if rank == 0:
    root.end_file(); root.write(); root.close()
q.close(); al.close(); ix.close()
fqd.barrier()
'''


def test_qc_consumers_of_two_ranks_merge_to_the_references_files(tmp_path):
    """StatCollector over two ranks: (1) the two FASTQ pairs of a --fq_list, one per rank; (2) one pair sharded by reference batch.
    The per-rank consumer states gathered to rank 0 and merged in input order must give the reference's 13 QC files -- for (1) those
    of its two-pair run (tests/golden/qc/ref_fqlist.*), for (2) those of the case itself."""
    from test_qc_consumer import QC_FILES, qc_bytes, explain
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "qc_worker.py"
    script.write_text(QC_WORKER % dict(root=ROOT, here=HERE, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)], env=env, timeout=900)
    gold = os.path.join(golden_util.GOLD, "qc")
    for out, pre in (("fqlist", "ref_fqlist.qc."), ("sharded", "ref.qc.")):
        bad = {}
        for f in QC_FILES:
            got = qc_bytes(str(tmp_path / (out + "." + f)))
            with gzip.open(os.path.join(gold, pre + f + ".gz"), "rb") as fh:
                want = fh.read()
            if f == "vcf":
                want = b"\n".join(ln for ln in want.split(b"\n") if not ln.startswith(b"##fileDate="))
            if got != want:
                bad[f] = (got, want)
        assert not bad, out + "\n" + explain(bad)
    with gzip.open(os.path.join(gold, "ref_fqlist.sam.gz"), "rb") as fh:
        want = b"".join(l for l in fh.read().splitlines(keepends=True) if not l.startswith(b"@"))
    assert (tmp_path / "fqlist.sam").read_bytes() == want


BROKEN_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import numpy as np
import golden_util, oracle_binding as ob
from fastquick_amd import api, dist as fqd
rank, local_rank, world = fqd.init("gloo")
L = api.load_library(os.path.join(%(here)r, "emu", "libfq_emu.so"))
g = golden_util.materialise("basic", os.path.join(%(tmp)r, "b%%d" %% rank))
names, seq, qual, lens = ob.read_fastq_pair(g["fq1"], g["fq2"])
lens = lens.copy()
lens[0, g["batch"] + 3] = 5          # a read of 5 bases in batch 1 (rank 1's): that call fails with FQ_ELIMIT
ix = api.Index(g["prefix"], lib=L)
al = api.Aligner(ix, api.default_opts(L), max_pairs=max(16, g["batch"]))
sh = fqd.StreamShard(al, rank, world)
failed = ""
try:
    sh.align_stream(names, seq, qual, lens, g["batch"])
except api.FastquickError as e:
    failed = str(e)
open(os.path.join(%(tmp)r, "broken_%%d.txt" %% rank), "w").write(failed)
al.close(); ix.close()
'''


def test_a_failed_call_of_a_sharded_stream_fails_the_ranks_behind_it(tmp_path):
    """A call that fails on one rank must not leave the owner of the next batch waiting for the stream state, nor let it go on with a
    stale one: the state that reaches it is marked, and its call fails too (both ranks return, both report)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "broken_worker.py"
    script.write_text(BROKEN_WORKER % dict(root=ROOT, here=HERE, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29551", str(script)], env=env, timeout=300)
    r0, r1 = (tmp_path / "broken_0.txt").read_text(), (tmp_path / "broken_1.txt").read_text()
    assert "read length outside" in r1, r1
    assert "call that failed" in r0, r0
