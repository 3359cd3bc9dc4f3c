#!/usr/bin/env python3
"""Randomised soak of the two consumers against the REAL reference (oracle/_ref/fq_ref_driver: its own StatCollector and
SetSamRecord in the loop): per seed a random reference, read set and option subset; the 13 QC files fq_qc writes and the BAM records
fq_bam writes (decoded by the independent reader of tests/test_bam_writer.py, tags in file order) must be the reference's.

    python tests/fuzz_consumers_vs_reference.py --seeds 40 --start 0              # host-loop tier (CPU, build container)
    python tests/fuzz_consumers_vs_reference.py --seeds 40 --start 0 --device 0   # the device path (GPU box; oracle/_ref travels there)
Since round 6 the consumers themselves run in kernels (--device_consumers 1, the default): the QC files and the BAM records come out of fq_emit.h / fq_deflate.h.
"""
import argparse, os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fastquick_amd import api, synth
import oracle_binding as ob
from test_bam_writer import check_bgzf, decode_bam
from test_qc_consumer import QC_FILES, qc_bytes

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=20)
ap.add_argument("--start", type=int, default=0)
ap.add_argument("--device", type=int, default=-1, help="HIP device (default: the host-loop library of tests/emu)")
ap.add_argument("--budget", type=float, default=0, help="stop after this many seconds (0: run all seeds)")
ap.add_argument("--device_consumers", type=int, default=1, help="1 (default since round 6): StatCollector's part and the BAM records / BGZF members in the kernels of fq_emit.h / fq_deflate.h "
                "(fq_ctx_attach_qc / fq_ctx_attach_bam); 0: the host statements from the result arrays")
args = ap.parse_args()
if not os.path.exists(ob.REF_DRIVER):
    sys.exit("oracle/_ref/fq_ref_driver missing: run `make -C oracle ref` in the build container")
if args.device >= 0:
    lib, dev = api.load_library(), dict(device=args.device)
else:
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libfq_emu.so"])
    lib = api.load_library(os.path.join(ROOT, "tests", "emu", "libfq_emu.so"))
    dev = {}
bad = done = 0
t_start = time.time()
for seed in range(args.start, args.start + args.seeds):
    if args.budget and time.time() - t_start > args.budget:
        break
    rnd = random.Random(seed)
    refkw = dict(n_markers=rnd.choice([12, 40, 90]), n_long=rnd.choice([0, 2, 6]), seed=5000 + seed, repeat_every=rnd.choice([0, 3, 6]), sex_every=rnd.choice([0, 5, 9]))
    read_len = rnd.choice([76, 100, 150, 150])
    readkw = dict(read_len=read_len, on_target=rnd.choice([0.7, 0.95, 1.0]), seed=6000 + seed, sub_rate=rnd.choice([0.005, 0.02]), del_frac=rnd.choice([0.0, 0.05]),
                  ins_frac=rnd.choice([0.0, 0.05]), n_rate=rnd.choice([0.0, 0.004]), indel_len_max=rnd.choice([1, 2]), chimera_frac=rnd.choice([0.0, 0.08]),
                  qual_decay=rnd.random() < 0.5, dup_frac=rnd.choice([0.0, 0.1]), edge_frac=rnd.choice([0.0, 0.1]), adapter_frac=rnd.choice([0.0, 0.05]))
    if read_len < 150:
        readkw.update(frag_mean=read_len + 120, frag_sd=20)
    n, batch = rnd.choice([(500, 200), (1200, 1200), (2000, 700)])
    se = rnd.random() < 0.25
    packed = rnd.random() < 0.5
    extra, okw = ["--batch", batch], {}
    if readkw["qual_decay"]:
        extra += ["--q", 15]; okw["trim_qual"] = 15
    cal_dup = rnd.random() < 0.8
    if not cal_dup:
        extra += ["--cal_dup", 0]
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix="fqcons%d_" % seed) as d:
        ref = synth.make_reference(**refkw)
        pre = os.path.join(d, "ref.FASTQuick.fa")
        ref.write_fasta(pre)
        subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=d)
        synth.write_qc_inputs(pre, ref)
        fai = os.path.join(d, "genome.fai")
        with open(fai, "w") as fh:
            for chrom in sorted({nm.split(":")[0] for nm in ref.names}):
                fh.write("%s\t%d\t%d\t60\t61\n" % (chrom, len(ref.genome), len(chrom) + 2))
        rb = synth.make_reads(ref, n, **readkw)
        f1, f2 = rb.write_fastq(os.path.join(d, "reads"))
        common = extra + ["--genome_size", len(ref.genome)] + (["--se", 1] if se else [])
        ob.run_reference(pre, f1, f2, os.path.join(d, "ref_out"), *common)
        ob.run_reference(pre, f1, f2, os.path.join(d, "ref_bam"), "--bam_dump", 1, "--fai", fai, *common)
        ix = api.Index(pre, lib=lib, **dev)
        al = api.Aligner(ix, api.default_opts(lib, single_end=1 if se else 0, **okw), max_pairs=max(16, batch))
        qc = api.QC(ix, pre, os.path.join(d, "got"), genome_size=len(ref.genome), read_len=151, cal_dup=1 if cal_dup else 0)
        bam = api.BamWriter(ix, fai, os.path.join(d, "got.bam"), cal_dup=1 if cal_dup else 0)
        if args.device_consumers:
            qc.attach(al); bam.attach(al)
        qc.begin_file(f1, f1 if se else f2)
        if se:
            api.align_stream(al, list(rb.names), rb.seq[:1], rb.qual[:1], rb.lens[:1], batch, None, None, qc=qc, bam=bam, packed=packed)
        else:
            api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, batch, None, None, qc=qc, bam=bam, packed=packed)
        qc.end_file(); qc.write(); qc.close(); bam.close(); al.close(); ix.close()
        why = [f for f in QC_FILES if qc_bytes(os.path.join(d, "got." + f)) != qc_bytes(os.path.join(d, "ref_out." + f))]
        check_bgzf(os.path.join(d, "got.bam"))
        text, _refs, recs = decode_bam(os.path.join(d, "got.bam"), sort_tags=False)
        if text != open(os.path.join(d, "ref_bam.bamhdr")).read():
            why.append("bam header")
        want = [l.rstrip("\n").split("\t") for l in open(os.path.join(d, "ref_bam.bamtxt"))]
        if recs != want:
            k = next((i for i, (a, b) in enumerate(zip(recs, want)) if a != b), min(len(recs), len(want)))
            why.append("bam record %d of %d/%d: %s | %s" % (k, len(recs), len(want), "\t".join(recs[k])[:300] if k < len(recs) else None, "\t".join(want[k])[:300] if k < len(want) else None))
    ok = not why
    bad += 0 if ok else 1
    done += 1
    print("seed %3d %s%s len %3d n %4d batch %4d %-14s %s %.1fs %s" % (seed, "SE" if se else "PE", "+pk" if packed else "   ", read_len, n, batch, " ".join(map(str, extra[2:])),
                                                                 "OK  " if ok else "FAIL", time.time() - t0, "" if ok else str(refkw) + str(readkw) + str(why)), flush=True)
print("%d cases, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
