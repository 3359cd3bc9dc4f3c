#!/usr/bin/env python3
"""Randomised pin of the oracle against the REAL reference (oracle/_ref/fq_ref_driver, built in place from /root/reference by
`make -C oracle ref`; only possible in the build container).  Seeded random references / read sets / the options the driver
exposes; the oracle must reproduce the reference's per-stage dump and SAM text.

    python tests/fuzz_oracle_vs_reference.py --seeds 40 --start 0
"""
import argparse, filecmp, os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: checkers are the only code that may touch oracle/)
from fastquick_amd import synth
import oracle_binding as ob

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=20)
ap.add_argument("--start", type=int, default=0)
ap.add_argument("--adversarial", action="store_true", help="tiny, repeat-rich references and error-rich reads: edge and tie-breaking cases")
ap.add_argument("--ragged", action="store_true", help="reads of mixed length (96 bp or more: DESIGN.md section 6); the SAM QUAL column, which the "
                "reference prints with the tail of an earlier longer read, is left out of the comparison")
ap.add_argument("--se", action="store_true", help="the single-end mapper (BwtMapper::SingleEndMapper) on the first file alone")
args = ap.parse_args()


def sam_without_qual(path):
    with open(path, "rb") as fh:
        return [[c for k, c in enumerate(ln.split(b"\t")) if k != 10] for ln in fh.read().split(b"\n")]



if not os.path.exists(ob.REF_DRIVER):
    sys.exit("oracle/_ref/fq_ref_driver missing: run `make -C oracle ref` in the build container")
bad = 0
for seed in range(args.start, args.start + args.seeds):
    rnd = random.Random(seed)
    refkw = dict(n_markers=rnd.choice([20, 60, 150]), n_long=rnd.choice([0, 3, 8]), seed=3000 + seed, repeat_every=rnd.choice([0, 2, 5]), tandem_every=rnd.choice([0, 7]))
    if args.adversarial:
        refkw.update(n_markers=rnd.choice([3, 5, 8]), n_long=rnd.choice([0, 1]), repeat_every=rnd.choice([1, 2, 3]), tandem_every=rnd.choice([0, 2, 5]))
    read_len = rnd.choice([150, 150, 250]) if args.ragged else rnd.choice([76, 100, 150, 150, 250])
    readkw = dict(read_len=read_len, on_target=rnd.choice([0.5, 0.9, 1.0]), seed=4000 + seed, sub_rate=rnd.choice([0.005, 0.02, 0.04]),
                  del_frac=rnd.choice([0.0, 0.05, 0.1]), ins_frac=rnd.choice([0.0, 0.05, 0.1]), n_rate=rnd.choice([0.0, 0.003, 0.01]),
                  indel_len_max=rnd.choice([1, 2, 3]), chimera_frac=rnd.choice([0.0, 0.05, 0.2]), qual_decay=rnd.random() < 0.4)
    if args.adversarial:
        readkw.update(on_target=1.0, sub_rate=rnd.choice([0.01, 0.03, 0.06]), del_frac=rnd.choice([0.1, 0.2]), ins_frac=rnd.choice([0.1, 0.2]), chimera_frac=rnd.choice([0.1, 0.3]))
    if read_len < 150:
        readkw.update(frag_mean=read_len + 120, frag_sd=20)
    n, batch = rnd.choice([(600, 250), (1200, 1200), (2500, 1000)])
    extra, okw = ["--batch", batch], {}
    if read_len > 150:
        extra += ["--read_len", read_len + 1]   # the reference sizes its read buffers once from --read_len (default 151)
    if readkw["qual_decay"]:
        extra += ["--q", 15]; okw["trim_qual"] = 15
    pick = rnd.random()
    if pick < 0.25:
        md = rnd.choice([2, 4, 6]); extra += ["--n", md]; okw.update(fnr=-1.0, max_diff=md)
    elif pick < 0.4:
        extra += ["--no_sw", 0]; okw.update(is_sw=0)
    elif pick < 0.5:
        th = rnd.choice([1, 5]); extra += ["--thresh", th]; okw.update(filter_thresh=th)
    # the other gap_opt_t / pe_opt_t fields (driver options named after the reference's command line)
    mode = 1 | 2        # BWA_MODE_GAPE | COMPREAD, the defaults
    for _ in range(rnd.choice([0, 0, 1, 2, 3])):
        o = rnd.choice(["o", "e", "i", "d", "l", "k", "m", "R", "N", "L", "I", "scores", "max_isize", "max_occ", "multi"])
        if o == "o": v = rnd.choice([0, 2]); extra += ["--o", v]; okw["max_gapo"] = v
        elif o == "e": v = rnd.choice([2, 6]); extra += ["--e", v]; okw["max_gape"] = v; mode &= ~1
        elif o == "i": v = rnd.choice([1, 10]); extra += ["--i", v]; okw["indel_end_skip"] = v
        elif o == "d": v = rnd.choice([1, 100]); extra += ["--d", v]; okw["max_del_occ"] = v
        elif o == "l": v = rnd.choice([20, 40]); extra += ["--l", v]; okw["seed_len"] = v
        elif o == "k": v = rnd.choice([0, 1, 3]); extra += ["--k", v]; okw["max_seed_diff"] = v
        elif o == "m": v = rnd.choice([200, 2000, 50000]); extra += ["--m", v]; okw["max_entries"] = v
        elif o == "R": v = rnd.choice([1, 5]); extra += ["--R", v]; okw["max_top2"] = v
        elif o == "N": extra += ["--N", 0]; mode |= 0x10; okw["max_top2"] = 0x7fffffff
        elif o == "L": extra += ["--L", 0]; mode |= 4
        elif o == "I" and not (mode & 0x200): extra += ["--I", 0]; mode |= 0x200
        elif o == "scores":
            m_, o_, e_ = rnd.choice([(3, 11, 4), (4, 4, 4), (2, 8, 3), (5, 7, 5)])
            extra += ["--M", m_, "--O", o_, "--E", e_]; okw.update(s_mm=m_, s_gapo=o_, s_gape=e_)
        elif o == "max_isize": v = rnd.choice([200, 1000]); extra += ["--max_isize", v]; okw["max_isize"] = v
        elif o == "max_occ": v = rnd.choice([2, 20]); extra += ["--max_occ", v]; okw["max_occ"] = v
        elif o == "multi": a, b = rnd.choice([(0, 0), (8, 20), (1, 2)]); extra += ["--n_multi", a, "--N_multi", b]; okw.update(n_multi=a, N_multi=b)
    if mode != (1 | 2):
        okw["mode"] = mode
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix="fqref%d_" % seed) as d:
        ref = synth.make_reference(**refkw)
        pre = os.path.join(d, "ref.FASTQuick.fa")
        ref.write_fasta(pre)
        subprocess.check_call([ob.REF_DRIVER, "index", pre], stderr=subprocess.DEVNULL, cwd=d)
        synth.write_qc_inputs(pre, ref)   # the reference driver runs the real StatCollector: it needs its input files
        rb = synth.make_reads(ref, n, **readkw)
        if args.ragged:
            import numpy as np
            lo = (40 if seed % 4 == 1 else 12) if seed % 2 else max(96, read_len - 54)      # odd seeds: reads under 96 bp too (down to 12), rows carry the slot history (Q7)
            rb.lens[:] = np.random.default_rng(seed).integers(lo, read_len + 1, rb.lens.shape)
            if args.se:      # the single-end reader hands out fresh zeroed buffers: nothing behind a short read
                for e in range(2):
                    for i in range(rb.seq.shape[1]):
                        rb.seq[e, i, rb.lens[e, i]:] = 0
            else:
                ob.apply_slot_history(rb.seq, rb.lens, batch)
        if mode & 0x200:
            rb.qual[rb.qual > 0] += 31          # the input then is Phred+64
        f1, f2 = rb.write_fastq(os.path.join(d, "reads"))
        ob.run_reference(pre, f1, f2, os.path.join(d, "ref_out"), *(extra + (["--se", 1] if args.se else [])))
        oa = ob.OracleAligner(pre, ob.default_opts(**okw))
        if args.se:
            oa.align_se(list(rb.names), rb.seq[0], rb.qual[0], rb.lens[0], d + "/o.st", d + "/o.sam", batch=batch)
        else:
            oa.align(rb.names, rb.seq, rb.qual, rb.lens, d + "/o.st", d + "/o.sam", batch=batch)
        oa.close()
        diffs = [x for x in ob.diff_stage_files(d + "/ref_out.stages", d + "/o.st")]
        if args.ragged and not args.se:
            same = sam_without_qual(d + "/ref_out.sam") == sam_without_qual(d + "/o.sam")
        else:
            same = filecmp.cmp(d + "/ref_out.sam", d + "/o.sam", shallow=False)
    ok = not diffs and same
    bad += 0 if ok else 1
    print("seed %3d len %3d n %4d batch %4d %-28s %s %.1fs %s" % (seed, read_len, n, batch, " ".join(map(str, extra[2:])), "OK  " if ok else "FAIL", time.time() - t0,
                                                                  "" if ok else str(refkw) + str(readkw) + str(diffs[:2])), flush=True)
sys.exit(1 if bad else 0)
