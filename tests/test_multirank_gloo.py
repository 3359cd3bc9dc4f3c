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
