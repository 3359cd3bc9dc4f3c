#!/usr/bin/env python3
"""The command line end to end on ON-TARGET input (every pair from a marker flank: BASELINE.json cfg 5's regime, the run
/root/reference/bin/FASTQuick_template.sh:474-481 stands for): two BGZF FASTQ files -> SAM text + the 13 QC files, and -> BAM + QC files,
whole-process wall time.  Nearly every pair survives the filter here, so the consumers of the records (SAM / BAM writer, StatCollector)
see every pair -- the WGS mix hides them behind a 0.2 % survival rate.

`measure()` is what bench.py's `front_end.cli_e2e_ontarget` calls; as a script: tools/cli_ontarget.py [pairs] [copies] [read_len] [extra CLI args...]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EOF_MEMBER = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"


def measure(exe, pre, ref, workdir, pairs=1 << 20, copies=8, read_len=150, threads=16, extra=(), modes=("sam_out", "bam_and_qc"), seed=4242, profile_dir=None, repeats=1):
    """pre: <index prefix>.FASTQuick.fa of `ref` (built).  Writes `pairs` seeded on-target pairs as two BGZF files, concatenated `copies` times
    (BGZF members concatenate), the QC inputs of the index, and runs the command line once per mode with stdout to /dev/null."""
    import numpy as np
    from fastquick_amd import synth
    os.makedirs(workdir, exist_ok=True)
    kw = {} if read_len >= 150 else dict(read_len=read_len, frag_mean=200, frag_sd=20, del_frac=0.05, ins_frac=0.05, indel_len_max=2)
    rb = synth.make_reads(ref, pairs, on_target=1.0, seed=seed, **kw)
    qual = np.frombuffer(b"F:,#", dtype=np.uint8)[np.random.default_rng(seed + 1).choice(4, size=(2, pairs, read_len), p=[0.7, 0.15, 0.1, 0.05])]
    big = [os.path.join(workdir, "ont_%d.fq.gz" % (e + 1)) for e in range(2)]
    text_bytes = 0
    for e in range(2):
        one = big[e] + ".one"
        text_bytes += synth.write_fastq_uniform(rb.seq[e], qual[e], read_len, one, threads=max(2, threads))
        blob = open(one, "rb").read()
        os.remove(one)
        blob = blob[:-28] if blob.endswith(EOF_MEMBER) else blob
        with open(big[e], "wb") as fo:
            for _ in range(copies):
                fo.write(blob)
            fo.write(EOF_MEMBER)
    synth.write_qc_inputs(pre, ref)
    synth.write_param(pre, ref, sum(1 for n in ref.names if n.endswith("|L")))
    with open(pre + ".genome.fa.fai", "w") as fh:
        fh.write("1\t%d\t3\t60\t61\n" % len(ref.genome))
    os.sync()      # (the gigabytes of input just written are on their way to the disk: a run timed beside their write-back waits for its own output files' pages)
    total = pairs * copies
    res = {"pairs": total, "distinct_pairs": pairs, "copies": copies, "read_len": read_len, "text_bytes": text_bytes * copies,
           "file_bytes": sum(os.path.getsize(p) for p in big), "input": "two BGZF FASTQ files (zlib level 1), every pair from a marker flank"}
    for mode in modes:
        cmd = [exe, "align", "--index_prefix", pre[:-len(".FASTQuick.fa")], "--fastq_1", big[0], "--fastq_2", big[1], "--out_prefix", os.path.join(workdir, "ont_out"),
               "--read_len", str(max(read_len, 151)), "--t", str(threads)] + (["--sam_out"] if mode == "sam_out" else []) + list(extra)
        if profile_dir:       # rocprofv3 --kernel-trace --stats around the command line itself (FQ_PROFILE_DIR): per-kernel times of the run
            cmd = ["rocprofv3", "--kernel-trace"] + (["--memory-copy-trace"] if os.environ.get("FQ_PROFILE_COPIES") else []) + ["--stats", "--output-format", "csv", "-d", os.path.join(profile_dir, mode), "-o", "p", "--"] + cmd
        walls = []
        for _ in range(max(1, repeats)):      # (whole-process wall time on a shared host: the runs are listed, the best one is the rate)
            # the previous run's output files go first, and their pages with them: truncating a gigabyte of dirty pages when the BAM file is opened, or writing beside
            # their write-back, costs a run up to a second that is not its own
            for fn in os.listdir(workdir):
                if fn.startswith("ont_out."):
                    os.remove(os.path.join(workdir, fn))
            os.sync()
            time.sleep(3.0)       # (a process started right behind another's exit waits for the driver to take that one's device memory back)
            t0 = time.perf_counter()
            run_ = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            walls.append(round(time.perf_counter() - t0, 2))
            if run_.returncode != 0 or len(walls) == 1 or walls[-1] == min(walls):
                run = run_
            if run_.returncode != 0:
                break
        dt = min(walls)
        err = run.stderr.decode(errors="replace").splitlines()
        res[mode] = {"rc": run.returncode, "wall_s": round(dt, 2), "wall_s_runs": walls, "pairs_per_s": round(total / dt, 1) if run.returncode == 0 else None,
                     "output": "SAM text (to /dev/null) + 13 QC files" if mode == "sam_out" else "BAM file + 13 QC files",
                     "notices": [l for l in err if "consumers" in l or "device time" in l or "reading (ms)" in l or "FATAL" in l][-4:]}
        if os.environ.get("FASTQUICK_TRACE"):
            res[mode]["trace"] = [l for l in err if l.startswith("TRACE")][-40:]
        if os.environ.get("FASTQUICK_CTX_TRACE"):
            res[mode]["ctx_trace"] = [l for l in err if l.startswith("[fq]")]
    for p in big:
        os.remove(p)
    return res


if __name__ == "__main__":
    from fastquick_amd import api, synth
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    read_len = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    wd = os.environ.get("FQ_BENCH_DIR", "/tmp/fq_e2e")
    os.makedirs(wd, exist_ok=True)
    fa = os.path.join(wd, "e2e.FASTQuick.fa")
    ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
    if not os.path.exists(fa + ".rsa"):
        ref.write_fasta(fa)
        api.build_index(fa)
    import json
    print(json.dumps(measure(os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd"), fa, ref, wd, pairs, copies, read_len, extra=sys.argv[4:], profile_dir=os.environ.get("FQ_PROFILE_DIR"))))
