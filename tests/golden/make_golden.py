#!/usr/bin/env python3
"""Regenerate tests/golden/* with the REAL reference (oracle/_ref/fq_ref_driver).

Only runs in the build container (needs /root/reference to have been compiled by
`make -C oracle ref`).  Everything written is DATA: synthetic inputs (seeded, from
fastquick_amd/synth.py), the index files the reference built for them, and the reference's
per-stage dumps / SAM text for those inputs.  No reference source is stored.

    python tests/golden/make_golden.py
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fastquick_amd import synth  # noqa: E402
import oracle_binding as ob  # noqa: E402

CASES = {
    # tag: (reference kwargs, read kwargs, n_pairs, batch, trim_qual)
    "basic": (dict(n_markers=16, n_long=2, seed=101),
              dict(on_target=0.85, seed=201, sub_rate=0.01, del_frac=0.06, ins_frac=0.05, n_rate=0.003,
                   indel_len_max=2, chimera_frac=0.08), 400, 150, 0),
    "repeat": (dict(n_markers=24, n_long=2, seed=102, repeat_every=2, tandem_every=6),
               dict(on_target=0.95, seed=202, sub_rate=0.03, del_frac=0.08, ins_frac=0.07, n_rate=0.003,
                    indel_len_max=3, chimera_frac=0.05), 400, 400, 0),
    "trim76": (dict(n_markers=16, n_long=2, seed=103, repeat_every=3),
               dict(read_len=76, on_target=0.9, seed=203, sub_rate=0.01, del_frac=0.1, ins_frac=0.1,
                    indel_len_max=2, frag_mean=200, frag_sd=20, qual_decay=True), 300, 128, 15),
    "isize": (dict(n_markers=12, n_long=2, seed=104),
              dict(on_target=0.7, seed=204, chimera_frac=0.3), 130, 100, 0),
    "nref": (dict(n_markers=12, n_long=1, seed=105, n_frac=0.002),
             dict(on_target=0.9, seed=205, n_rate=0.01, chimera_frac=0.1, qual_decay=True), 200, 200, 15),
    # 250 bp reads: the reference needs --read_len (its buffers are sized once from it); longer DP windows on the device
    # reads that hang over the ends of their contigs (fragments start up to 60 bp before a flank): StatCollector::AddAlignment
    # turns such hits into NO_MATCH before the records are printed
    "edge": (dict(n_markers=14, n_long=2, seed=108),
             dict(on_target=0.9, seed=208, sub_rate=0.01, del_frac=0.03, ins_frac=0.03, edge_frac=0.35), 300, 300, 0),
    # the QC consumer's case: more pairs per marker (depth, pileups, duplicates), markers on X and Y, trimming
    "qc": (dict(n_markers=40, n_long=4, seed=109, sex_every=7),
           dict(on_target=0.95, seed=209, sub_rate=0.01, del_frac=0.03, ins_frac=0.03, n_rate=0.002, chimera_frac=0.05, qual_decay=True,
                dup_frac=0.08, edge_frac=0.1), 3000, 1024, 15),
    # SA intervals of >= 1000 rows (MIN_HASH_WIDTH, libbwa/bwape.h:105): 1,292 markers with one and the same window, so that a read from any of
    # them has 1,292 equally good places -- the (k,l) -> positions cache of src/BwtMapper.cpp:815-843, keyed by interval, holding the positions
    # computed with the length of the FIRST read that asked; --q 15 on decaying qualities gives the reads ragged trimmed lengths,
    # so later requesters of an interval differ in length from the first.  A few unique windows in front keep insert-size inference alive.
    # The reads are drawn with the eight unique windows weighted 50-fold (`front_weight`: about a quarter of the on-target pairs), so that
    # insert-size inference succeeds and the pairing sweep scores the 1,292 x 1,292 candidate places of a repeat pair with a real estimate.
    "wide": (dict(n_markers=1300, n_long=2, seed=110, identical_from=8),
             dict(on_target=0.9, seed=210, sub_rate=0.01, del_frac=0.03, ins_frac=0.03, qual_decay=True, front_weight=(8, 50)), 600, 256, 15),
    "long250": (dict(n_markers=12, n_long=3, seed=106),
                dict(read_len=250, on_target=0.9, seed=206, sub_rate=0.01, del_frac=0.06, ins_frac=0.05, indel_len_max=3,
                     chimera_frac=0.08, frag_mean=430, frag_sd=30), 160, 160, 0),
}
# Real reads: the 251 pairs of 151 bp (one of 137 bp; lower-case bases, Illumina names with comments, real quality strings) that
# the reference ships as its own example input (example/fq.test.list).  The reference FASTA its example.sh names is not in the
# example can build here (see make_cfg0_case), so the reduced reference is synthetic with 118 of the pairs planted in marker flanks (lightly mutated copies).
EXAMPLE_FQ = ("/root/reference/example/ERR013170_1.filt.fastq.gz.1000.fastq.gz",
              "/root/reference/example/ERR013170_2.filt.fastq.gz.1000.fastq.gz")
# One batch: with full batches the reference's mate-name check (src/BwtMapper.cpp:2087-2092) would abort on the first pair
# (TestRead_1 / TestRead_2); its own example only runs because 251 pairs never fill a batch.
EXAMPLE_CASE = ("example151", dict(n_markers=100, n_long=6, seed=107), 256, 15)


def read_fastq_raw(path):
    with gzip.open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    return [(lines[i], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


def plant_pairs(r1, r2):
    code = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
        code[c + 32] = i

    def codes(seq, rng):
        c = code[np.frombuffer(seq, dtype=np.uint8)].copy()
        bad = c == 255
        c[bad] = rng.integers(0, 4, int(bad.sum()), dtype=np.uint8)
        return c

    def patch(genome, pos, flank, rng):
        nxt = 0
        for k in range(len(pos)):
            lo, hi = int(pos[k]) - 1 - int(flank[k]), int(pos[k]) + int(flank[k])
            at = lo + int(rng.integers(0, 30))
            while nxt < len(r1):
                a = codes(r1[nxt][1], rng)
                b = (3 - codes(r2[nxt][1], rng))[::-1]
                frag = np.concatenate([a, rng.integers(0, 4, int(rng.integers(0, 140)), dtype=np.uint8), b])
                if rng.random() < 0.5:
                    frag = (3 - frag)[::-1]
                m = rng.random(len(frag)) < 0.008                      # the planted copy differs a little from the reads
                frag[m] = (frag[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
                if rng.random() < 0.25:
                    q = int(rng.integers(20, len(frag) - 20))
                    frag = np.delete(frag, [q, q + 1][:int(rng.integers(1, 3))]) if rng.random() < 0.5 else np.insert(frag, q, rng.integers(0, 4, 1, dtype=np.uint8))
                if at + len(frag) > hi + 20:
                    break
                genome[at:at + len(frag)] = frag
                at += len(frag) + int(rng.integers(0, 40))
                nxt += 1
    return patch


def run_bam_dump(tmp, pre, f1, f2, ref, *args):
    """The reference's BAM branch (SetSamRecord / SetSamFileHeader) on the same input: one text line per record, and the header."""
    with open(os.path.join(tmp, "genome.fai"), "w") as fh:      # the original reference's .fai: one contig per chromosome name in use
        for chrom in sorted({nm.split(":")[0] for nm in ref.names}):
            fh.write("%s\t%d\t%d\t60\t61\n" % (chrom, len(ref.genome), len(chrom) + 2))
    ob.run_reference(pre, f1, f2, os.path.join(tmp, "bam_out"), "--bam_dump", 1, "--fai", os.path.join(tmp, "genome.fai"), *args)


def write_case(out, tmp, pre, f1, f2, n, batch, q, refkw, readkw, genome_size, qc_read_len=151):
    shutil.copy(pre, os.path.join(out, "ref.FASTQuick.fa"))
    for ext in INDEX_EXT:
        shutil.copy(pre + ext, os.path.join(out, "ref.FASTQuick.fa" + ext))
    sparse_to_npz(pre + ".rollhash.sparse", os.path.join(out, "rollhash_bits.npz"))
    qc = [(pre + ext, "ref.FASTQuick.fa" + ext + ".gz") for ext in QC_IN_EXT] + [(os.path.join(tmp, "ref_out" + ext), "ref.qc" + ext + ".gz") for ext in QC_OUT_EXT]
    qc += [(os.path.join(tmp, "bam_out.bamtxt"), "ref.bamtxt.gz"), (os.path.join(tmp, "bam_out.bamhdr"), "ref.bamhdr.gz"), (os.path.join(tmp, "genome.fai"), "genome.fai.gz")]
    for src, dst in [(f1, "reads_1.fq.gz"), (f2, "reads_2.fq.gz"),
                     (os.path.join(tmp, "ref_out.stages"), "ref.stages.gz"),
                     (os.path.join(tmp, "ref_out.sam"), "ref.sam.gz")] + qc:
        with open(src, "rb") as fi, gzip.GzipFile(os.path.join(out, dst), "wb", mtime=0) as fo:
            fo.write(fi.read())
    with open(os.path.join(out, "case.txt"), "w") as fh:
        fh.write("n_pairs=%d\nbatch=%d\ntrim_qual=%d\ngenome_size=%d\nqc_read_len=%d\nref=%r\nreads=%r\n" % (n, batch, q, genome_size, qc_read_len, refkw, readkw))


def make_example_case():
    tag, refkw, batch, q = EXAMPLE_CASE
    out = os.path.join(HERE, tag)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    r1, r2 = read_fastq_raw(EXAMPLE_FQ[0]), read_fastq_raw(EXAMPLE_FQ[1])
    with tempfile.TemporaryDirectory() as tmp:
        ref = synth.make_reference(patch=plant_pairs(r1, r2), **refkw)
        pre = os.path.join(tmp, "ref.FASTQuick.fa")
        ref.write_fasta(pre)
        subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=tmp)
        synth.write_qc_inputs(pre, ref)
        fq = []
        for e, src in enumerate(EXAMPLE_FQ):
            dst = os.path.join(tmp, "reads_%d.fq" % (e + 1))
            with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
                fo.write(fi.read())
            fq.append(dst)
        ob.run_reference(pre, fq[0], fq[1], os.path.join(tmp, "ref_out"), "--batch", batch, "--q", q, "--read_len", 152, "--genome_size", len(ref.genome))
        run_bam_dump(tmp, pre, fq[0], fq[1], ref, "--batch", batch, "--q", q, "--read_len", 152)
        write_case(out, tmp, pre, fq[0], fq[1], len(r1), batch, q, refkw, "reference example/fq.test.list", len(ref.genome), 152)
    print(tag, "->", sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")


# configs[0] on the repository's own data: example/ref.test.fa (one 50 kb contig "22"), example/hapmap.test.vcf.gz (candidate sites),
# example/fq.test.list (251 pairs).  `FASTQuick index` itself cannot run here (RefBuilder shells out to bcftools, absent:
# src/RefBuilder.cpp:451-459), so the marker windows are cut by this script -- candidate SNPs whose REF agrees with the FASTA,
# left to right, windows of 2 x 250 + 1 bases that do not overlap, named as RefBuilder names them -- and everything from there on
# (index build, align, StatCollector, SetSamRecord) is the reference's own code on the reference's own reads.
CFG0_FA = "/root/reference/example/ref.test.fa"
CFG0_VCF = "/root/reference/example/hapmap.test.vcf.gz"


def make_cfg0_case():
    tag, batch, q = "cfg0_example", 256, 15
    out = os.path.join(HERE, tag)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    lines = open(CFG0_FA, "rb").read().split(b"\n")
    chrom = lines[0][1:].split()[0].decode()
    text = np.frombuffer(b"".join(lines[1:]), dtype=np.uint8)
    code = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = code[c + 32] = i
    genome = code[text]
    flank = 250
    pos, names, seqs, last_end = [], [], [], 0
    with gzip.open(CFG0_VCF, "rt") as fh:
        recs = sorted((int(f[1]), f[3], f[4]) for f in (ln.split("\t") for ln in fh if not ln.startswith("#")) if len(f[3]) == 1 and len(f[4]) == 1)
    for p1, r, a in recs:
        if p1 - flank <= last_end or p1 + flank > len(text) or chr(text[p1 - 1]).upper() != r:
            continue
        pos.append(p1)
        names.append("%s:%d@%s/%s" % (chrom, p1, r, a))
        seqs.append(text[p1 - 1 - flank:p1 + flank].copy())
        last_end = p1 + flank
    ref = synth.SynthRef(names, seqs, genome, np.array(pos, dtype=np.int64), np.full(len(pos), flank, dtype=np.int64))
    r1 = read_fastq_raw(EXAMPLE_FQ[0])
    with tempfile.TemporaryDirectory() as tmp:
        pre = os.path.join(tmp, "ref.FASTQuick.fa")
        ref.write_fasta(pre)
        subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=tmp)
        synth.write_qc_inputs(pre, ref)
        fq = []
        for e, src in enumerate(EXAMPLE_FQ):
            dst = os.path.join(tmp, "reads_%d.fq" % (e + 1))
            with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
                fo.write(fi.read())
            fq.append(dst)
        ob.run_reference(pre, fq[0], fq[1], os.path.join(tmp, "ref_out"), "--batch", batch, "--q", q, "--read_len", 152, "--genome_size", len(genome))
        run_bam_dump(tmp, pre, fq[0], fq[1], ref, "--batch", batch, "--q", q, "--read_len", 152)
        write_case(out, tmp, pre, fq[0], fq[1], len(r1), batch, q, "reference example/ref.test.fa, %d marker windows cut at example/hapmap.test.vcf.gz sites" % len(pos),
                   "reference example/fq.test.list", len(genome), 152)
    print(tag, "->", len(pos), "markers,", sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")


INDEX_EXT = [".bwt", ".rbwt", ".sa", ".rsa", ".pac", ".ann", ".amb"]
# what StatCollector reads beside the reduced reference (inputs), and the files it writes (expected outputs)
QC_IN_EXT = [".SelectedSite.vcf", ".dbSNP.subset.vcf", ".gc"]
QC_OUT_EXT = [".InsertSizeTable", ".DepthDist", ".GCDist", ".EmpRepDist", ".EmpCycleDist", ".RawInsertSizeDist", ".AdjustedInsertSizeDist",
              ".SexChromInfo", ".Pileup", ".FASTQ.csv", ".Sequence.csv", ".Summary", ".vcf"]


def sparse_to_npz(path_sparse: str, path_npz: str) -> None:
    arrs = {}
    with open(path_sparse, "rb") as fh:
        for t in range(6):
            n = int(np.frombuffer(fh.read(8), dtype=np.uint64)[0])
            bits = np.frombuffer(fh.read(4 * n), dtype=np.uint32)
            arrs["t%d" % t] = np.diff(bits.astype(np.int64), prepend=0).astype(np.uint32)   # delta coded
    np.savez_compressed(path_npz, **arrs)


SE_CASES = ("basic", "repeat", "trim76", "edge", "long250", "nref", "example151")
SE_CONSUMER_CASES = ("basic", "trim76", "edge")


def add_se_outputs(tag):
    """The reference's single-end mapper (BwtMapper::SingleEndMapper, driver option --se 1) on the first FASTQ of an existing case:
    ref_se.stages / ref_se.sam next to the paired-end goldens (the case's inputs stay as they are)."""
    import golden_util
    out = os.path.join(HERE, tag)
    with tempfile.TemporaryDirectory() as tmp:
        g = golden_util.materialise(tag, tmp)
        args = ["--se", 1, "--batch", g["batch"], "--genome_size", g["genome_size"]] + (["--q", g["trim_qual"]] if g["trim_qual"] else []) + \
               (["--read_len", g["qc_read_len"]] if g["qc_read_len"] != 151 else [])
        ob.run_reference(g["prefix"], g["fq1"], g["fq2"], os.path.join(tmp, "se_out"), *args)
        files = [("se_out.stages", "ref_se.stages.gz"), ("se_out.sam", "ref_se.sam.gz")]
        if tag in SE_CONSUMER_CASES:     # StatCollector's files and the BAM branch (SetSamRecord(p, 0)) of the same single-end run
            ob.run_reference(g["prefix"], g["fq1"], g["fq2"], os.path.join(tmp, "se_bam"), "--bam_dump", 1, "--fai", os.path.join(tmp, "genome.fai"), *args)
            files += [("se_out" + ext, "ref_se.qc" + ext + ".gz") for ext in QC_OUT_EXT] + [("se_bam.bamtxt", "ref_se.bamtxt.gz"), ("se_bam.bamhdr", "ref_se.bamhdr.gz")]
        for src, dst in files:
            with open(os.path.join(tmp, src), "rb") as fi, gzip.GzipFile(os.path.join(out, dst), "wb", mtime=0) as fo:
                fo.write(fi.read())
    print(tag, "-> single-end goldens")


FQLIST_CASES = ("qc",)


def add_fqlist_outputs(tag):
    """The reference on TWO FASTQ pairs in one run (a --fq_list of two lines: one StatCollector, one FileStatCollector per pair; driver
    option --more): the case's reads cut in two halves, half_a_[12].fq and half_b_[12].fq (golden_util.split_halves).  Stored:
    ref_fqlist.sam and the 13 QC files ref_fqlist.qc.* -- what a run sharded over two ranks by FASTQ pair must put together."""
    import golden_util
    out = os.path.join(HERE, tag)
    with tempfile.TemporaryDirectory() as tmp:
        g = golden_util.materialise(tag, tmp)
        (a1, a2), (b1, b2) = golden_util.split_halves(g, tmp)
        args = ["--batch", g["batch"], "--genome_size", g["genome_size"]] + (["--q", g["trim_qual"]] if g["trim_qual"] else []) + \
               (["--read_len", g["qc_read_len"]] if g["qc_read_len"] != 151 else [])
        ob.run_reference(g["prefix"], a1, a2, os.path.join(tmp, "fl_out"), "--more", b1 + "," + b2, *args)
        files = [("fl_out.sam", "ref_fqlist.sam.gz")] + [("fl_out" + ext, "ref_fqlist.qc" + ext + ".gz") for ext in QC_OUT_EXT]
        for src, dst in files:
            with open(os.path.join(tmp, src), "rb") as fi, gzip.GzipFile(os.path.join(out, dst), "wb", mtime=0) as fo:
                fo.write(fi.read())
    print(tag, "-> two-pair (--fq_list) goldens")


FRAC_CASES = {"qc": 0.8}


def add_frac_outputs(tag):
    """The reference with --frac_samp (gap_opt_t::frac): its reader draws a number per record from Random(round) and drops the records
    above the fraction (src/BwtMapper.cpp:500-507).  Stored: ref_frac.sam and the 13 QC files ref_frac.qc.* of the down-sampled run."""
    import golden_util
    out = os.path.join(HERE, tag)
    with tempfile.TemporaryDirectory() as tmp:
        g = golden_util.materialise(tag, tmp)
        args = ["--batch", g["batch"], "--genome_size", g["genome_size"], "--frac_samp", FRAC_CASES[tag]] + (["--q", g["trim_qual"]] if g["trim_qual"] else [])
        ob.run_reference(g["prefix"], g["fq1"], g["fq2"], os.path.join(tmp, "fr_out"), *args)
        for src, dst in [("fr_out.sam", "ref_frac.sam.gz")] + [("fr_out" + ext, "ref_frac.qc" + ext + ".gz") for ext in QC_OUT_EXT]:
            with open(os.path.join(tmp, src), "rb") as fi, gzip.GzipFile(os.path.join(out, dst), "wb", mtime=0) as fo:
                fo.write(fi.read())
    print(tag, "-> --frac_samp %g goldens" % FRAC_CASES[tag])


def main() -> None:
    if not os.path.exists(ob.REF_DRIVER):
        sys.exit("oracle/_ref/fq_ref_driver missing: run `make -C oracle ref` in the build container")
    only = set(sys.argv[1:])   # optional: regenerate just these cases (inputs of the others are committed and stay as they are)
    for tag, (refkw, readkw, n, batch, q) in CASES.items():
        if only and tag not in only:
            continue
        out = os.path.join(HERE, tag)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        with tempfile.TemporaryDirectory() as tmp:
            ref = synth.make_reference(**refkw)
            pre = os.path.join(tmp, "ref.FASTQuick.fa")
            ref.write_fasta(pre)
            subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=tmp)
            synth.write_qc_inputs(pre, ref)
            rkw = dict(readkw)
            read_ref = ref
            if "front_weight" in rkw:      # the first n_front markers drawn `times` as often as the others
                n_front, times = rkw.pop("front_weight")
                sel = np.concatenate([np.tile(np.arange(n_front), times), np.arange(n_front, len(ref.marker_pos))])
                read_ref = synth.SynthRef(ref.names, ref.seqs, ref.genome, ref.marker_pos[sel], ref.flank[sel])
            rb = synth.make_reads(read_ref, n, **rkw)
            f1, f2 = rb.write_fastq(os.path.join(tmp, "reads"))
            args = ["--batch", batch] + (["--q", q] if q else []) + (["--read_len", readkw["read_len"] + 1] if readkw.get("read_len", 150) > 150 else [])
            ob.run_reference(pre, f1, f2, os.path.join(tmp, "ref_out"), "--genome_size", len(ref.genome), *args)
            run_bam_dump(tmp, pre, f1, f2, ref, *args)
            write_case(out, tmp, pre, f1, f2, n, batch, q, refkw, readkw, len(ref.genome), readkw["read_len"] + 1 if readkw.get("read_len", 150) > 150 else 151)
        print(tag, "->", sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")
    if not only or EXAMPLE_CASE[0] in only:
        make_example_case()
    if not only or "cfg0_example" in only:
        make_cfg0_case()
    for tag in SE_CASES:
        if not only or tag in only or "se" in only:
            add_se_outputs(tag)
    for tag in FRAC_CASES:
        if not only or tag in only or "frac" in only:
            add_frac_outputs(tag)
    for tag in FQLIST_CASES:
        if not only or tag in only or "fqlist" in only:
            add_fqlist_outputs(tag)


if __name__ == "__main__":
    main()
