#!/usr/bin/env python3
"""The command line over the host-loop library (tests/emu: the kernel bodies as host loops -- TEST INFRASTRUCTURE) built with AddressSanitizer +
UndefinedBehaviorSanitizer, or ThreadSanitizer, and run over the kinds of input the front ends know: BGZF (the device front end's path), gzip
streams, plain text; SAM and BAM + QC; single end; truncated and bit-flipped files; an odd record mid-file (hand-over); CR LF; an --fq_list over
three workers.  Prints the sanitizers' findings per run (a run that ends in a refusal exits with its threads alive: TSan's "thread leak" there is
expected).  CPU only; GPU sanitizers are not available on the pool.
    python tools/sanitize_cli.py --san asan|tsan [--work DIR]"""
import argparse
import gzip
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util  # noqa: E402
from fastquick_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--san", choices=("asan", "tsan"), default="asan")
ap.add_argument("--work", default=None)
a = ap.parse_args()
work = a.work or tempfile.mkdtemp(prefix="fq_" + a.san + "_")
os.makedirs(work, exist_ok=True)
C, E = os.path.join(ROOT, "fastquick_amd", "csrc"), os.path.join(ROOT, "tests", "emu")
flags = ["-fsanitize=address,undefined"] if a.san == "asan" else ["-fsanitize=thread"]
srcs = [os.path.join(E, "fq_emu_backend.cpp")] + [os.path.join(C, f) for f in ("fq_align.cpp", "fq_index.cpp", "fq_sam.cpp", "fq_pack.cpp", "fq_qc.cpp", "fq_bam.cpp", "fq_fastq.cpp", "fq_frontend.cpp")]
common = ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-omit-frame-pointer", "-Wno-unknown-pragmas", "-I" + C, "-I" + os.path.join(ROOT, "include")] + flags
subprocess.check_call(common + ["-fPIC", "-shared", "-o", os.path.join(work, "libfq_emu.so")] + srcs + ["-lpthread", "-lz"])
cli = os.path.join(work, "FASTQuick_emu")
subprocess.check_call(common + ["-o", cli, os.path.join(C, "fq_cli.cpp"), "-L" + work, "-lfq_emu", "-lz", "-lpthread", "-Wl,-rpath," + work])
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=0", TSAN_OPTIONS="halt_on_error=0")
total = 0


def run(tag, prefix, args):
    global total
    cmd = [cli, "align", "--index_prefix", prefix] + args + ["--out_prefix", os.path.join(work, tag), "--batch_pairs", "256", "--chunk_pairs", "256"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=3600)
    err = r.stderr.decode(errors="replace").splitlines()
    found = sorted(set(l.strip()[:160] for l in err if "runtime error" in l or ("SUMMARY:" in l and "Sanitizer" in l)))
    real = [f for f in found if not ("thread leak" in f and r.returncode == 1)]
    total += len(real)
    print("%-28s rc %3d  findings %d %s | %s" % (tag, r.returncode, len(real), real[:4], err[-1][:100] if err else ""), flush=True)


for case in ("edge", "qc"):
    g = golden_util.materialise(case, os.path.join(work, case))
    with open(g["prefix"] + ".param", "w") as fh:
        fh.write("REFERENCE_PATH\t%s\nTARGET_REGION_PATH\tEmpty\nDBSNP_VCF_PATH\tEmpty\nNUM_VAR_LONG\t4\nNUM_VAR_SHORT\t36\nSHORT_FLANK_LENGTH\t250\nLONG_FLANK_LENGTH\t1000\n"
                 % os.path.join(g["dir"], "genome"))
    prefix = g["prefix"][:-len(".FASTQuick.fa")]
    t1, t2 = open(g["fq1"], "rb").read(), open(g["fq2"], "rb").read()

    def w(name, data):
        p = os.path.join(work, case + "_" + name)
        with open(p, "wb") as fh:
            fh.write(data)
        return p
    bg = lambda t: synth.bgzf_compress(t, threads=1, level=6, member=3000)   # noqa: E731
    b1, b2 = w("1.bgzf.gz", bg(t1)), w("2.bgzf.gz", bg(t2))
    z1, z2 = w("1.gz", gzip.compress(t1, 6)), w("2.gz", gzip.compress(t2, 1))
    run(case + "_bgzf_bam", prefix, ["--fastq_1", b1, "--fastq_2", b2])
    run(case + "_bgzf_sam", prefix, ["--fastq_1", b1, "--fastq_2", b2, "--sam_out"])
    run(case + "_bgzf_single_end", prefix, ["--fastq_1", b1, "--sam_out"])
    run(case + "_gzip_sam", prefix, ["--fastq_1", z1, "--fastq_2", z2, "--sam_out"])
    run(case + "_plain_bam", prefix, ["--fastq_1", g["fq1"], "--fastq_2", g["fq2"]])
    bb, zz = open(b1, "rb").read(), open(z1, "rb").read()
    run(case + "_bgzf_truncated", prefix, ["--fastq_1", w("1t.bgzf.gz", bb[:len(bb) // 2 + 7]), "--fastq_2", b2, "--sam_out"])
    fl = bytearray(bb); fl[len(fl) // 3] ^= 0x10
    run(case + "_bgzf_bit_flip", prefix, ["--fastq_1", w("1f.bgzf.gz", bytes(fl)), "--fastq_2", b2, "--sam_out"])
    run(case + "_gzip_truncated", prefix, ["--fastq_1", w("1t.gz", zz[:len(zz) // 2]), "--fastq_2", z2, "--sam_out"])
    fl = bytearray(zz); fl[len(fl) // 2] ^= 0x04
    run(case + "_gzip_bit_flip", prefix, ["--fastq_1", w("1f.gz", bytes(fl)), "--fastq_2", z2, "--sam_out"])
    lines = t1.split(b"\n"); k = 4 * min(300, len(lines) // 8) + 1
    lines[k] = lines[k][:40] + b"\n" + lines[k][40:]                          # a wrapped base line: the device's part ends in front of it
    run(case + "_hand_over", prefix, ["--fastq_1", w("1odd.bgzf.gz", bg(b"\n".join(lines))), "--fastq_2", b2, "--sam_out"])
    run(case + "_cr_lf", prefix, ["--fastq_1", w("1crlf.bgzf.gz", bg(t1.replace(b"\n", b"\r\n"))), "--fastq_2", b2, "--sam_out"])
    lst = os.path.join(work, case + "_list.txt")
    with open(lst, "w") as fh:
        fh.write("".join("%s\t%s\n" % (b1, b2) for _ in range(3)))
    run(case + "_fq_list_3_workers", prefix, ["--fq_list", lst, "--devices", "0,0,0", "--sam_out"])
print("%d findings" % total)
sys.exit(1 if total else 0)
