#!/usr/bin/env python3
"""Randomised check of the command line front end (FASTQ reader, read-slot model, name rules, --fq_list, --I, --q) against the REAL
reference (oracle/_ref/fq_ref_driver), CPU only: `FASTQuick_amd align` is linked over the host-loop library (tests/emu, test
infrastructure), so what is exercised here is fq_cli.cpp + the host pipeline, not the HIP kernels (tests/fuzz_parity.py does those).

    python tests/fuzz_cli_vs_reference.py --seeds 50 --start 0
"""
import argparse, gzip, os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fastquick_amd import synth
import oracle_binding as ob

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=20)
ap.add_argument("--start", type=int, default=0)
args = ap.parse_args()
EMU = os.path.join(ROOT, "tests", "emu")
if not os.path.exists(ob.REF_DRIVER):
    sys.exit("oracle/_ref/fq_ref_driver missing: run `make -C oracle ref` in the build container")
subprocess.check_call(["make", "-s", "-C", EMU, "libfq_emu.so", "FASTQuick_emu"])
CLI = os.path.join(EMU, "FASTQuick_emu")


def sam_cols(text, drop_qual):
    rows = [ln.split(b"\t") for ln in text.split(b"\n")]
    return [[c for k, c in enumerate(r) if not (drop_qual and k == 10)] for r in rows]


bad = 0
for seed in range(args.start, args.start + args.seeds):
    rnd = random.Random(seed)
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix="fqcli%d_" % seed) as d:
        ref = synth.make_reference(n_markers=rnd.choice([12, 40]), n_long=rnd.choice([0, 3]), seed=7000 + seed, repeat_every=rnd.choice([0, 2, 3]))
        pre = os.path.join(d, "ref.FASTQuick.fa")
        ref.write_fasta(pre)
        subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=d)
        synth.write_qc_inputs(pre, ref)   # the reference driver runs the real StatCollector: it needs its input files
        n = rnd.choice([300, 700])
        batch = rnd.choice([64, 100, 256, 1024])
        il13 = rnd.random() < 0.25
        trim = rnd.random() < 0.4
        ragged = rnd.random() < 0.5
        rb = synth.make_reads(ref, n, read_len=150, on_target=rnd.choice([0.6, 0.95]), seed=8000 + seed, sub_rate=rnd.choice([0.005, 0.03]),
                              del_frac=0.06, ins_frac=0.05, chimera_frac=rnd.choice([0.0, 0.15]), n_rate=0.003, qual_decay=trim)
        single_batch = batch > n                     # differently named mates survive the reference's name check only then
        mate_names = single_batch and rnd.random() < 0.6
        name_style = rnd.choice(["fixed", "varying", "slash"])
        wrap = rnd.random() < 0.3
        lo = rnd.choice([15, 40, 100])
        fq = []
        for end in range(2):
            out = []
            for i in range(n):
                ln = int(rb.lens[end, i])
                if ragged:
                    ln = random.Random(seed * 1000003 + end * 7919 + i).randint(lo, 150)
                s = rb.seq[end, i, :ln].tobytes()
                q = rb.qual[end, i, :ln].tobytes()
                if il13:
                    q = bytes(c + 31 for c in q)
                if i % 11 == 0:
                    s = s.lower()
                nm = rb.names[i]
                if name_style == "varying":
                    nm += b":" + b"x" * ((i * 7 + (i // batch) * 5) % 23)
                elif name_style == "slash":
                    nm += (b"/1", b"/2")[end]
                if mate_names:
                    nm += (b".a", b".mate")[end]
                if rnd.random() < 0.2:
                    nm += (b" ", b"\t")[i & 1] + b"comment 1:N:0"
                body = b"\n".join(s[k:k + 61] for k in range(0, len(s), 61)) if wrap else s
                out.append(b"@" + nm + b"\n" + body + b"\n+\n" + q + b"\n")
            path = os.path.join(d, "r%d.fq.gz" % (end + 1))
            with gzip.open(path, "wb", compresslevel=1) as fh:
                fh.write(b"".join(out))
            fq.append(path)
        extra, cli = ["--batch", str(batch)], ["--batch_pairs", str(batch), "--chunk_pairs", str(batch * rnd.choice([1, 2, 3]))]
        if trim:
            extra += ["--q", "15"]; cli += ["--q", "15"]
        if il13:
            extra += ["--I", "0"]; cli += ["--I"]
        refrun = subprocess.run([ob.REF_DRIVER, "align", pre, fq[0], fq[1], os.path.join(d, "ref_out")] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        run = subprocess.run([CLI, "align", "--index_prefix", pre[:-len(".FASTQuick.fa")], "--fastq_1", fq[0], "--fastq_2", fq[1],
                              "--out_prefix", os.path.join(d, "cli"), "--sam_out"] + cli, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if refrun.returncode != 0:
            ok = run.returncode != 0
            why = "reference refused the input; the front end %s" % ("did too" if ok else "did not")
        elif run.returncode != 0:
            ok, why = False, run.stderr.decode(errors="replace")[-300:]
        else:
            with open(os.path.join(d, "ref_out.sam"), "rb") as fh:
                want = fh.read()
            ok = sam_cols(run.stdout, ragged) == sam_cols(want, ragged)
            why = "" if ok else "SAM text differs"
    bad += 0 if ok else 1
    print("seed %4d n %3d batch %4d %s%s%s%s names=%s%s %s %.1fs %s" % (seed, n, batch, "ragged(%d) " % lo if ragged else "", "q15 " if trim else "", "I " if il13 else "",
                                                                       "wrap " if wrap else "", name_style, "+mate" if mate_names else "", "OK  " if ok else "FAIL", time.time() - t0, why), flush=True)
sys.exit(1 if bad else 0)
