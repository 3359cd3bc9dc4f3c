#!/usr/bin/env python3
"""Reference-held digests for shapes far above the goldens' (<= 3,000 pairs): the REAL reference (oracle/_ref/fq_ref_driver: the reference's own
stage functions, StatCollector and bwa_print_sam1, compiled from /root/reference where it lies) runs seeded synthetic inputs at its own batch size,
and the SHA-256 of its SAM text and of each of its 13 QC files is committed as tests/golden/large_digests.json.  The inputs are seeded
(fastquick_amd/synth.py), so the GPU tier regenerates them on the GPU box and holds the HIP path and the command line to these digests
(tests/test_large_digests.py) -- the reference itself cannot travel.

  (a) real_batches   2,000 markers, 2 x 262,144 + 40,000 pairs, a fifth on target: three reference batches (src/BwtMapper.h:36), three
                     insert-size inferences, the last_ii chain, the mate-name check at full batches
  (b) cfg3_trim      100,000 markers, one batch of 262,144 pairs of the real-shaped WGS mix (N bases, decaying qualities, adapter tails, duplicates,
                     reads over contig ends), --q 15
  (c) ont76          10,000 markers, one batch of 262,144 pairs of 2x76 indel-rich on-target reads (BASELINE cfg 5's reads)

Run here (build container):  python tests/golden/make_large_digests.py [a b c]      (about 15 minutes, 8 GB)"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

QC_FILES = ["InsertSizeTable", "DepthDist", "GCDist", "EmpRepDist", "EmpCycleDist", "RawInsertSizeDist", "SexChromInfo", "Pileup",
            "FASTQ.csv", "Sequence.csv", "Summary", "AdjustedInsertSizeDist", "vcf"]
B = 262144


def shape_inputs(key, synth, np):
    """(ref, reads, align options) of a shape: shared with tests/test_large_digests.py, which must build exactly the same inputs"""
    if key == "real_batches":
        ref = synth.make_reference(n_markers=2000, n_long=200, seed=501)
        rb = synth.make_reads(ref, 2 * B + 40000, on_target=0.2, seed=502, sub_rate=0.008, del_frac=0.03, ins_frac=0.02, n_rate=0.001, chimera_frac=0.02)
        return ref, rb, dict(read_len=150, trim_qual=0)
    if key == "cfg3_trim":
        ref = synth.make_reference(n_markers=100000, n_long=10000, seed=91)
        rb = synth.make_reads(ref, B, on_target=0.021, seed=92, n_rate=0.001, qual_decay=True, sub_rate=0.008, del_frac=0.02, ins_frac=0.01,
                              adapter_frac=0.01, dup_frac=0.05, edge_frac=0.02)
        good = np.random.default_rng(93).random(rb.qual.shape[:2]) < 0.5
        rb.qual[good] = ord("I")
        return ref, rb, dict(read_len=150, trim_qual=15)
    if key == "ont76":
        ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
        rb = synth.make_reads(ref, B, read_len=76, on_target=1.0, seed=3100, frag_mean=200, frag_sd=20, del_frac=0.05, ins_frac=0.05, indel_len_max=2)
        return ref, rb, dict(read_len=76, trim_qual=0)
    raise KeyError(key)


def qc_digest(path):
    data = open(path, "rb").read()
    if path.endswith(".vcf"):
        data = b"\n".join(ln for ln in data.split(b"\n") if not ln.startswith(b"##fileDate="))
    return hashlib.sha256(data).hexdigest(), len(data)


def summary_digest(path):
    """.Summary without nothing removed; .FASTQ.csv names the input files by base name: the tests write files of the same names"""
    return qc_digest(path)


def main():
    import numpy as np
    import oracle_binding as ob
    from fastquick_amd import api, synth
    keys = sys.argv[1:] or ["real_batches", "cfg3_trim", "ont76"]
    keys = [{"a": "real_batches", "b": "cfg3_trim", "c": "ont76"}.get(k, k) for k in keys]
    out_path = os.path.join(HERE, "large_digests.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for key in keys:
        with tempfile.TemporaryDirectory(dir=os.environ.get("FQ_DIGEST_TMP")) as tmp:
            ref, rb, o = shape_inputs(key, synth, np)
            pre = os.path.join(tmp, "ref.FASTQuick.fa")
            ref.write_fasta(pre)
            subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=tmp)      # the reference's own index builder
            synth.write_qc_inputs(pre, ref)
            L = o["read_len"]
            fq = [os.path.join(tmp, "reads_%d.fq" % (e + 1)) for e in range(2)]
            for e in range(2):
                synth.write_fastq_uniform(rb.seq[e], rb.qual[e], L, fq[e], bgzf=False)
            extra = ["--batch", B, "--genome_size", len(ref.genome), "--read_len", 151] + (["--q", o["trim_qual"]] if o["trim_qual"] else [])
            ob.run_reference(pre, fq[0], fq[1], os.path.join(tmp, "ref"), *extra)
            sam = open(os.path.join(tmp, "ref.sam"), "rb").read()
            d = {"pairs": int(rb.seq.shape[1]), "read_len": L, "trim_qual": o["trim_qual"], "batch": B, "genome_size": int(len(ref.genome)), "qc_read_len": 151,
                 "sam_sha256": hashlib.sha256(sam).hexdigest(), "sam_bytes": len(sam), "sam_lines": sam.count(b"\n"), "qc": {}}
            for f in QC_FILES:
                h, n = qc_digest(os.path.join(tmp, "ref." + f))
                d["qc"][f] = {"sha256": h, "bytes": n}
            out[key] = d
            print(key, json.dumps(d)[:300], flush=True)
            with open(out_path, "w") as fh:
                json.dump(out, fh, indent=1, sort_keys=True)
                fh.write("\n")


if __name__ == "__main__":
    main()
