"""Helpers that unpack the committed golden fixtures (tests/golden/<case>/) into a scratch dir."""
from __future__ import annotations

import gzip
import os
import shutil

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX_EXT = [".bwt", ".rbwt", ".sa", ".rsa", ".pac", ".ann", ".amb"]


def case_tags():
    return sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)) and not d.startswith("_"))


def case_params(tag: str) -> dict:
    out = {}
    with open(os.path.join(GOLD, tag, "case.txt")) as fh:
        for line in fh:
            k, v = line.rstrip("\n").split("=", 1)
            out[k] = int(v) if k in ("n_pairs", "batch", "trim_qual", "genome_size", "qc_read_len") else v
    return out


def write_sparse(npz_path: str, out_path: str) -> None:
    z = np.load(npz_path)
    with open(out_path, "wb") as fh:
        for t in range(6):
            bits = np.cumsum(z["t%d" % t].astype(np.int64)).astype(np.uint32)
            fh.write(np.uint64(len(bits)).tobytes())
            fh.write(bits.tobytes())


def materialise(tag: str, dst: str) -> dict:
    src = os.path.join(GOLD, tag)
    os.makedirs(dst, exist_ok=True)
    pre = os.path.join(dst, "ref.FASTQuick.fa")
    shutil.copy(os.path.join(src, "ref.FASTQuick.fa"), pre)
    for ext in INDEX_EXT:
        shutil.copy(os.path.join(src, "ref.FASTQuick.fa" + ext), pre + ext)
    write_sparse(os.path.join(src, "rollhash_bits.npz"), pre + ".rollhash.sparse")
    paths = {}
    for name in ("reads_1.fq", "reads_2.fq", "ref.stages", "ref.sam"):
        with gzip.open(os.path.join(src, name + ".gz"), "rb") as fi, open(os.path.join(dst, name), "wb") as fo:
            fo.write(fi.read())
        paths[name] = os.path.join(dst, name)
    for name in sorted(os.listdir(src)):      # StatCollector inputs next to the reference, and the QC files the reference wrote
        if name.startswith("ref.FASTQuick.fa.") and name.endswith(".gz") or name.startswith(("ref.qc.", "ref.bam", "genome.fai", "ref_se.", "ref_fqlist.", "ref_frac.")):
            with gzip.open(os.path.join(src, name), "rb") as fi, open(os.path.join(dst, name[:-3]), "wb") as fo:
                fo.write(fi.read())
    p = case_params(tag)
    p.update(prefix=pre, fq1=paths["reads_1.fq"], fq2=paths["reads_2.fq"], stages=paths["ref.stages"], sam=paths["ref.sam"], dir=dst)
    if os.path.exists(os.path.join(dst, "ref_se.sam")):      # the reference's single-end mapper on reads_1.fq alone
        p.update(se_stages=os.path.join(dst, "ref_se.stages"), se_sam=os.path.join(dst, "ref_se.sam"))
    return p


def split_halves(g: dict, dst: str):
    """The case's FASTQ pair cut into two pairs of files (first half of the records, the rest): the two lines of a --fq_list."""
    out = []
    texts = [open(g[k], "rb").read().split(b"\n") for k in ("fq1", "fq2")]
    n = (len(texts[0]) - 1) // 4
    cut = n // 2
    for half, (lo, hi) in (("a", (0, cut)), ("b", (cut, n))):
        paths = []
        for e in range(2):
            path = os.path.join(dst, "half_%s_%d.fq" % (half, e + 1))
            with open(path, "wb") as fh:
                fh.write(b"\n".join(texts[e][4 * lo:4 * hi]) + b"\n")
            paths.append(path)
        out.append(tuple(paths))
    return out


def se_case_tags() -> list:
    return [t for t in case_tags() if os.path.exists(os.path.join(GOLD, t, "ref_se.sam.gz"))]
