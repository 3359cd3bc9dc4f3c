"""Seeded synthetic inputs for the FASTQuick `align` hot path (tests + bench.py).

Geometry follows SURVEY.md section 8(d): a uniform-random genome with one marker every
`spacing` bp; the first `n_long` markers get the long flank (contig name suffix ``|L``), the
rest the short flank, exactly like the reduced reference RefBuilder writes
(reference: src/RefBuilder.cpp:585-613 -- header ``>CHR:POS@REF/ALT[|L]``, sequence
``flank + REF + flank``).  Read pairs are fragments N(frag_mean, frag_sd) cut from the genome
(on-target: overlapping a marker's flank region) or i.i.d. random sequence (off-target).

Nothing here is on the product path; it only fabricates inputs.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Optional

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


@dataclasses.dataclass
class SynthRef:
    """A reduced reference: contig names, sequences (ASCII uint8 arrays) and the source genome."""
    names: list
    seqs: list
    genome: np.ndarray          # codes 0..3
    marker_pos: np.ndarray      # 1-based marker coordinates in the genome
    flank: np.ndarray           # flank length per marker

    @property
    def l_pac(self) -> int:
        return int(sum(len(s) for s in self.seqs))

    def write_fasta(self, path: str) -> None:
        with open(path, "wb") as fh:
            for n, s in zip(self.names, self.seqs):
                fh.write(b">" + n.encode() + b"\n" + s.tobytes() + b"\n")


def make_reference(n_markers: int = 40, n_long: int = 4, flank_short: int = 250, flank_long: int = 1000,
                   spacing: int = 3000, seed: int = 12345, repeat_every: int = 0, repeat_div: float = 0.005,
                   n_frac: float = 0.0, tandem_every: int = 0, patch=None, sex_every: int = 0, identical_from: int = 0) -> SynthRef:
    """Build a synthetic genome + reduced reference.

    repeat_every>0 makes every `repeat_every`-th window a diverged copy of its predecessor
    (repeat-rich variant, exercises c1>1 / drand48 / XA paths); n_frac sprinkles N into flanks.
    identical_from>0 makes the windows of the markers behind that one exact copies of its window: a read from any of them hits an SA
    interval as wide as their number (the (k,l) position cache of src/BwtMapper.cpp:815-843 takes intervals of >= 1000 rows).
    """
    rng = np.random.default_rng(seed)
    glen = 2000 + spacing * n_markers + 2000
    genome = rng.integers(0, 4, glen, dtype=np.uint8)
    if repeat_every:
        for k in range(1, n_markers):
            if k % repeat_every == 0:
                a = 2000 + spacing * (k - 1) - spacing // 2
                b = a + spacing
                seg = genome[a:a + spacing].copy()
                mut = rng.random(spacing) < repeat_div
                seg[mut] = (seg[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
                genome[b:b + spacing] = seg
    if tandem_every:
        for k in range(0, n_markers, tandem_every):
            p = 2000 + spacing * k + 60
            unit = genome[p:p + 40].copy()
            for r in range(1, 3):
                genome[p + 40 * r:p + 40 * (r + 1)] = unit
    pos = 2000 + spacing * np.arange(n_markers, dtype=np.int64)   # 1-based
    flank = np.where(np.arange(n_markers) < n_long, flank_long, flank_short).astype(np.int64)
    if identical_from:
        p0, f0 = int(pos[identical_from]), int(flank[identical_from])
        for k in range(identical_from + 1, n_markers):
            genome[int(pos[k]) - 1 - f0:int(pos[k]) + f0] = genome[p0 - 1 - f0:p0 + f0]
    if patch is not None:      # caller-supplied edit of the genome (codes 0..3) before the flanks are cut, e.g. to plant given reads
        patch(genome, pos, flank, rng)
    names, seqs = [], []
    for k in range(n_markers):
        p, f = int(pos[k]), int(flank[k])
        codes = genome[p - 1 - f:p + f]
        s = _ACGT[codes].copy()
        if n_frac > 0:
            m = rng.random(len(s)) < n_frac
            m[f] = False
            s[m] = ord("N")
        ref = int(genome[p - 1])
        alt = (ref + 1 + k % 3) % 4
        chrom = "1" if not sex_every or k % sex_every else ("X" if (k // sex_every) % 2 == 0 else "Y")   # a few markers on the sex chromosomes (StatCollector's .SexChromInfo)
        nm = "%s:%d@%s/%s" % (chrom, p, "ACGT"[ref], "ACGT"[alt])
        if f == flank_long:
            nm += "|L"
        names.append(nm)
        seqs.append(s)
    return SynthRef(names, seqs, genome, pos, flank)


@dataclasses.dataclass
class ReadBatch:
    """n read pairs; seq/qual are (2, n, L) ASCII uint8, lens (2, n) int32, names list of bytes."""
    seq: np.ndarray
    qual: np.ndarray
    lens: np.ndarray
    names: list

    @property
    def n(self) -> int:
        return self.seq.shape[1]

    def write_fastq(self, prefix: str) -> tuple:
        paths = []
        for e in range(2):
            path = "%s_%d.fq" % (prefix, e + 1)
            with open(path, "wb") as fh:
                for i in range(self.n):
                    ln = int(self.lens[e, i])
                    fh.write(b"@" + self.names[i] + b"\n" + self.seq[e, i, :ln].tobytes() + b"\n+\n"
                             + self.qual[e, i, :ln].tobytes() + b"\n")
            paths.append(path)
        return tuple(paths)


def bgzf_compress(data: bytes, threads: int = 8, level: int = 1, member: int = 65280) -> bytes:
    """BGZF as bgzip writes it (SAM spec 4.1): gzip members of at most 64 KiB of text with a BC extra field, then the empty
    end-of-file member.  Members are compressed on a thread pool (zlib releases the GIL)."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def one(ch: bytes) -> bytes:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        return (struct.pack("<4BI2BH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
                + comp + struct.pack("<II", zlib.crc32(ch) & 0xffffffff, len(ch)))
    view = memoryview(data)
    chunks = [view[i:i + member] for i in range(0, len(data), member)]
    with ThreadPoolExecutor(max(1, threads)) as ex:
        parts = list(ex.map(lambda c: one(bytes(c)), chunks))
    return b"".join(parts) + one(b"")


def write_fastq_uniform(seq: np.ndarray, qual: np.ndarray, read_len: int, path: str, name_prefix: bytes = b"r", bgzf: bool = True, threads: int = 8) -> int:
    """One FASTQ file of n reads of one length from rows seq / qual [n][>= read_len], names <prefix>%09d: the text is laid out with
    numpy (a record is a row of a byte matrix), then written as BGZF or plain text.  Returns the bytes of text."""
    n = seq.shape[0]
    nw = 1 + len(name_prefix) + 9
    rec = nw + 1 + read_len + 3 + read_len + 1
    m = np.empty((n, rec), dtype=np.uint8)
    m[:, 0] = ord("@")
    m[:, 1:1 + len(name_prefix)] = np.frombuffer(name_prefix, dtype=np.uint8)
    idx = np.arange(n, dtype=np.int64)
    for d in range(9):
        m[:, nw - 1 - d] = (idx // 10 ** d % 10 + 48).astype(np.uint8)
    m[:, nw] = 10
    m[:, nw + 1:nw + 1 + read_len] = seq[:, :read_len]
    m[:, nw + 1 + read_len] = 10
    m[:, nw + 2 + read_len] = ord("+")
    m[:, nw + 3 + read_len] = 10
    m[:, nw + 4 + read_len:nw + 4 + 2 * read_len] = qual[:, :read_len]
    m[:, rec - 1] = 10
    text = m.tobytes()
    with open(path, "wb") as fh:
        fh.write(bgzf_compress(text, threads) if bgzf else text)
    return len(text)


def _revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[..., ::-1]]


def make_reads(ref: SynthRef, n_pairs: int, read_len: int = 150, on_target: float = 1.0, seed: int = 7,
               sub_rate: float = 0.005, del_frac: float = 0.02, ins_frac: float = 0.0, n_rate: float = 0.0,
               frag_mean: float = 350.0, frag_sd: float = 30.0, qual_decay: bool = False,
               name_prefix: str = "r", indel_len_max: int = 1, chimera_frac: float = 0.0,
               name_offset: int = 0, edge_frac: float = 0.0, dup_frac: float = 0.0, adapter_frac: float = 0.0) -> ReadBatch:
    """Vectorised read-pair synthesis.  Mates are randomly swapped (so read 1 is on either strand).

    Off-target pairs are i.i.d. random sequence; on-target pairs are cut from the genome inside a marker
    window and then mutated (substitutions, optional indels / N / chimeric mates).

    "Real-shaped" extras (SURVEY 8d cfg 4; each draws from its own generator, so the defaults leave earlier fixtures as they were):
    edge_frac -- fragments that begin up to 60 bp before / end up to 60 bp behind their marker window, so that a mate hangs over the
    end of its contig; dup_frac -- pairs that are exact copies of the pair before them (PCR duplicates); adapter_frac -- reads whose
    last 5..40 bases are adapter sequence."""
    rng = np.random.default_rng(seed)
    L = read_len
    on = rng.random(n_pairs) < on_target
    on_idx = np.flatnonzero(on)
    n_on = len(on_idx)
    out = np.empty((2, n_pairs, L), dtype=np.uint8)
    # off-target (and default) content: uniform random bases, generated as bytes and mapped 2 bits -> base
    raw = rng.integers(0, 256, (2, n_pairs, (L + 3) // 4), dtype=np.uint8)
    codes = np.empty((2, n_pairs, ((L + 3) // 4) * 4), dtype=np.uint8)
    for k in range(4):
        codes[:, :, k::4] = (raw >> (2 * k)) & 3
    out[:] = _ACGT[codes[:, :, :L]]
    del raw, codes
    if n_on:
        g = ref.genome
        frag = np.clip(np.rint(rng.normal(frag_mean, frag_sd, n_on)).astype(np.int64), L + 10, None)
        k = rng.integers(0, len(ref.marker_pos), n_on)
        lo = ref.marker_pos[k] - 1 - ref.flank[k]
        hi = ref.marker_pos[k] + ref.flank[k]              # exclusive end of window
        span = np.maximum(hi - lo - frag, 1)
        start = lo + (rng.random(n_on) * span).astype(np.int64)
        if edge_frac > 0:
            rng_e = np.random.default_rng(seed + 7919)
            sel = rng_e.random(n_on) < edge_frac
            over = rng_e.integers(1, 61, n_on)
            left = rng_e.random(n_on) < 0.5
            start = np.where(sel & left, lo - over, np.where(sel & ~left, hi - frag + over, start))
        ch = rng.random(n_on) < chimera_frac if chimera_frac > 0 else np.zeros(n_on, dtype=bool)
        idx = np.arange(L, dtype=np.int32)
        extra = indel_len_max + 2
        idxx = np.arange(L + extra, dtype=np.int32)
        r1 = g[np.clip(start.astype(np.int32)[:, None] + idxx[None, :], 0, len(g) - 1)]             # forward strand, left end (+extra)
        end = start + frag
        k2 = rng.integers(0, len(ref.marker_pos), n_on)
        end = np.where(ch, ref.marker_pos[k2] + 50, end)   # chimeric mate from another marker window
        r2f = g[np.clip(end.astype(np.int32)[:, None] - 1 - idxx[None, :], 0, len(g) - 1)]          # reverse order from right end
        r2 = (3 - r2f).astype(np.uint8)                                            # complement -> read 2 as sequenced
        for e, r in enumerate((r1, r2)):
            if del_frac > 0 or ins_frac > 0:
                u = rng.random(n_on)
                dsel = (u < del_frac) if e == 0 else (u < del_frac * 0.5)
                isel = (u >= del_frac) & (u < del_frac + ins_frac)
                ppos = rng.integers(20, L - 20, n_on).astype(np.int32)
                dl = rng.integers(1, indel_len_max + 1, n_on).astype(np.int32)
                col = np.broadcast_to(idx[None, :], (n_on, L))
                src = col.copy()
                m = dsel[:, None] & (col >= ppos[:, None])
                src[m] += np.broadcast_to(dl[:, None], (n_on, L))[m]
                m2 = isel[:, None] & (col >= ppos[:, None] + dl[:, None])
                src[m2] -= np.broadcast_to(dl[:, None], (n_on, L))[m2]
                body = np.take_along_axis(r, src, axis=1)
                m3 = isel[:, None] & (col >= ppos[:, None]) & (col < ppos[:, None] + dl[:, None])
                body[m3] = rng.integers(0, 4, int(m3.sum()), dtype=np.uint8)
            else:
                body = r[:, :L].copy()
            if sub_rate > 0:
                m = rng.random((n_on, L), dtype=np.float32) < sub_rate
                body[m] = (body[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            asc = _ACGT[body]
            if n_rate > 0:
                m = rng.random((n_on, L), dtype=np.float32) < n_rate
                asc[m] = ord("N")
            out[e, on_idx] = asc
    swap = np.flatnonzero(rng.random(n_pairs) < 0.5)
    tmp = out[0, swap].copy()
    out[0, swap] = out[1, swap]
    out[1, swap] = tmp
    if adapter_frac > 0:
        rng_a = np.random.default_rng(seed + 104729)
        adapter = np.frombuffer(b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACGATCTCGTATGCCGTCTTCTGCTTG", dtype=np.uint8)
        for e in range(2):
            for i in np.flatnonzero(rng_a.random(n_pairs) < adapter_frac):
                k = int(rng_a.integers(5, 41))
                out[e, i, L - k:] = adapter[:k]
    if qual_decay:
        base = 40 - (np.arange(L) * 38 // L)
        q = np.clip(base[None, None, :] + rng.integers(-3, 4, (2, n_pairs, L)), 2, 41).astype(np.uint8) + 33
    else:
        q = np.full((2, n_pairs, L), ord("I"), dtype=np.uint8)
    if dup_frac > 0:
        rng_d = np.random.default_rng(seed + 15485863)
        for i in np.flatnonzero(rng_d.random(n_pairs) < dup_frac):
            if i > 0:
                out[:, i] = out[:, i - 1]
                q[:, i] = q[:, i - 1]
    lens = np.full((2, n_pairs), L, dtype=np.int32)
    names = [("%s%09d" % (name_prefix, i + name_offset)).encode() for i in range(n_pairs)]
    return ReadBatch(out, q, lens, names)


def write_qc_inputs(prefix_fa: str, ref: SynthRef, dbsnp_extra: int = 3, seed: int = 5) -> None:
    """The files StatCollector reads next to the reduced reference (src/StatCollector.cpp:1742-1839), as `FASTQuick index` leaves
    them: <fa>.SelectedSite.vcf (one record per marker, ID ending in `L` for long-flank markers, src/RefBuilder.cpp:408),
    <fa>.dbSNP.subset.vcf (known variant sites: the markers plus a few positions inside their flanks) and <fa>.gc (per marker a
    u32 length 2*flank+1 and that many bytes: G/C count of the 100 bp genome window [i-50, i+49] around each flank position,
    src/RefBuilder.cpp:38-54).  Records in position order per chromosome, the order RefBuilder::PrepareRefSeq writes them."""
    rng = np.random.default_rng(seed)
    g = ref.genome
    is_gc = ((g == 1) | (g == 2)).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(is_gc)])
    order = sorted(range(len(ref.names)), key=lambda k: (ref.names[k].split(":")[0], int(ref.marker_pos[k])))
    hdr = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    with open(prefix_fa + ".SelectedSite.vcf", "w") as fv, open(prefix_fa + ".dbSNP.subset.vcf", "w") as fd, open(prefix_fa + ".gc", "wb") as fg:
        fv.write(hdr)
        fd.write(hdr)
        for k in order:
            nm = ref.names[k]
            chrom, rest = nm.split(":")
            pos = int(ref.marker_pos[k])
            alleles = rest.split("@")[1].split("|")[0]
            r, a = alleles.split("/")
            long_ = nm.endswith("|L")
            fv.write("%s\t%d\trs%d%s\t%s\t%s\t.\tPASS\tAF=%.4f\n" % (chrom, pos, 1000 + k, "|L" if long_ else "", r, a, 0.05 + 0.9 * ((k * 37) % 100) / 100.0))
            f = int(ref.flank[k])
            sites = sorted(set([pos] + [int(x) for x in rng.integers(pos - f, pos + f + 1, dbsnp_extra)]))
            for sp in sites:
                fd.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\n" % (chrom, sp))
            gc = np.zeros(2 * f + 1, dtype=np.uint8)
            for t, i in enumerate(range(pos - f, pos + f + 1)):          # window [i-50, i+49], 1-based, clipped to the genome
                lo, hi = max(i - 50, 1), min(i + 49, len(g))
                gc[t] = csum[hi] - csum[lo - 1] if hi >= lo else 0
            fg.write(np.uint32(2 * f + 1).tobytes())
            fg.write(gc.tobytes())


def write_param(prefix_fa: str, ref: SynthRef, n_long: int) -> None:
    """The 7-line .param file `align` expects next to the index (src/FASTQuick.cpp:145-151)."""
    with open(prefix_fa + ".param", "w") as fh:
        fh.write("REFERENCE_PATH\t%s\n" % (prefix_fa + ".genome.fa"))
        fh.write("TARGET_REGION_PATH\tEmpty\n")
        fh.write("DBSNP_VCF_PATH\tEmpty\n")
        fh.write("NUM_VAR_LONG\t%d\n" % n_long)
        fh.write("NUM_VAR_SHORT\t%d\n" % (len(ref.names) - n_long))
        fh.write("SHORT_FLANK_LENGTH\t250\n")
        fh.write("LONG_FLANK_LENGTH\t1000\n")
