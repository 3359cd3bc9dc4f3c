#!/usr/bin/env python3
"""Randomised parity soak: seeded random references / read sets / options, device path vs the oracle, stage dumps and SAM text.

    python tests/fuzz_parity.py --seeds 40 --start 0          # needs a GPU; prints one line per case, exits non-zero on a mismatch
"""
import argparse, filecmp, os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: checkers are the only code that may touch oracle/)
from fastquick_amd import api, synth
import oracle_binding as ob

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=20)
ap.add_argument("--start", type=int, default=0)
ap.add_argument("--adversarial", action="store_true", help="tiny, repeat-rich references and error-rich reads: edge and tie-breaking cases")
ap.add_argument("--ragged", action="store_true", help="reads of mixed length, 96 bp or more")
ap.add_argument("--se", action="store_true", help="single-end contexts (fq_opts_t::single_end) against the oracle's single-end mapper")
args = ap.parse_args()
lib = api.load_library()
bad = 0
for seed in range(args.start, args.start + args.seeds):
    rnd = random.Random(seed)
    d = tempfile.mkdtemp(prefix="fqfuzz%d_" % seed)
    refkw = dict(n_markers=rnd.choice([30, 80, 200]), n_long=rnd.choice([0, 4, 10]), seed=1000 + seed,
                 repeat_every=rnd.choice([0, 2, 5]), tandem_every=rnd.choice([0, 7]))   # (no N in the reference: without a .rollhash the bitmaps of such a reference depend on libc rand())
    refkw["n_long"] = min(refkw["n_long"], refkw["n_markers"])
    if args.adversarial:
        refkw.update(n_markers=rnd.choice([3, 5, 8]), n_long=rnd.choice([0, 1]), repeat_every=rnd.choice([1, 2, 3]), tandem_every=rnd.choice([0, 2, 5]))
    read_len = rnd.choice([150, 150, 250]) if args.ragged else rnd.choice([76, 100, 150, 150, 250])
    readkw = dict(read_len=read_len, on_target=rnd.choice([0.5, 0.9, 1.0]), seed=2000 + seed, sub_rate=rnd.choice([0.005, 0.02, 0.04]),
                  del_frac=rnd.choice([0.0, 0.05, 0.1]), ins_frac=rnd.choice([0.0, 0.05, 0.1]), n_rate=rnd.choice([0.0, 0.003, 0.01]),
                  indel_len_max=rnd.choice([1, 2, 3]), chimera_frac=rnd.choice([0.0, 0.05, 0.2]), qual_decay=rnd.random() < 0.4)
    if args.adversarial:
        readkw.update(on_target=1.0, sub_rate=rnd.choice([0.01, 0.03, 0.06]), del_frac=rnd.choice([0.1, 0.2]), ins_frac=rnd.choice([0.1, 0.2]), chimera_frac=rnd.choice([0.1, 0.3]))
    if read_len < 150:
        readkw.update(frag_mean=read_len + 120, frag_sd=20)
    okw = {}
    if readkw["qual_decay"]:
        okw["trim_qual"] = 15
    pick = rnd.random()
    if pick < 0.15: okw.update(max_entries=rnd.choice([300, 1000, 5000]))
    elif pick < 0.3: okw.update(mode=1 | 2 | 0x10, max_top2=0x7fffffff)
    elif pick < 0.45: okw.update(fnr=-1.0, max_diff=rnd.choice([2, 4, 6]), max_gapo=rnd.choice([1, 2]), max_gape=rnd.choice([3, 6]), mode=rnd.choice([2, 3]))
    elif pick < 0.55: okw.update(s_mm=4, s_gapo=4, s_gape=rnd.choice([2, 4]))
    elif pick < 0.65: okw.update(n_multi=rnd.choice([0, 8]), N_multi=rnd.choice([0, 20]), max_occ=rnd.choice([10, 50, 100000]), is_sw=rnd.choice([0, 1]))
    if rnd.random() < 0.3: okw.update(filter_thresh=rnd.choice([1, 2, 4, 5]))
    for _ in range(rnd.choice([0, 0, 1, 2])):   # the remaining search / pairing options
        o = rnd.choice(["i", "d", "l", "k", "R", "L", "max_isize", "e", "I"])
        if o == "i": okw["indel_end_skip"] = rnd.choice([1, 10])
        elif o == "d": okw["max_del_occ"] = rnd.choice([1, 100])
        elif o == "l": okw["seed_len"] = rnd.choice([20, 40])
        elif o == "k": okw["max_seed_diff"] = rnd.choice([0, 1, 3])
        elif o == "R": okw["max_top2"] = rnd.choice([1, 5])
        elif o == "L": okw["mode"] = okw.get("mode", 3) | 4
        elif o == "I": okw["mode"] = okw.get("mode", 3) | 0x200
        elif o == "max_isize": okw["max_isize"] = rnd.choice([200, 1000])
        elif o == "e": okw["max_gape"] = rnd.choice([2, 6]); okw["mode"] = okw.get("mode", 3) & ~1
    mode = rnd.choice(["lanes", "nogap", "handover", "wave1", "wave64"])
    tuning = {}
    if mode == "nogap":          # every launch begins with the round that searches without gap children
        tuning = {"gap_nogap_min": 0}
    elif mode == "handover":     # searches still running after 8 pops once the queue is dry go to the wavefront-per-read kernel
        tuning = {"gap_long_pops": 8}
    elif mode != "lanes":
        tuning = {"gap_long_pops": 1 if mode == "wave1" else 64, "gap_long_always": 1}
    packed = rnd.random() < 0.5
    if packed:
        tuning["packed_bulk_min"] = rnd.choice([0, 1 << 30])
    n, batch = rnd.choice([(1500, 600), (3000, 3000), (5000, 2048), (12000, 4000)])
    call = batch * rnd.choice([1, 1, 2, 3])            # several reference batches per call
    threads = rnd.choice([0, 0, 3, 8])
    if threads:
        tuning["host_par_min"] = 1
    t0 = time.time()
    ref = synth.make_reference(**refkw)
    pre = os.path.join(d, "ref.FASTQuick.fa")
    ref.write_fasta(pre); api.build_index(pre)
    rb = synth.make_reads(ref, n, **readkw)
    if okw.get("mode", 0) & 0x200:
        rb.qual[rb.qual > 0] += 31              # Phred+64 input
    if args.ragged:
        import numpy as np
        lo = (40 if seed % 4 == 1 else 15) if seed % 2 else max(96, read_len - 54)      # odd seeds: reads under 96 bp too (down to FQ_LMIN), rows carry the slot history (Q7)
        rb.lens[:] = np.random.default_rng(seed).integers(lo, read_len + 1, rb.lens.shape)
        if args.se:
            for e in range(2):
                for i in range(rb.seq.shape[1]):
                    rb.seq[e, i, rb.lens[e, i]:] = 0
        else:
            ob.apply_slot_history(rb.seq, rb.lens, batch)
    ix = api.Index(pre, device=0)
    if args.se:
        packed = False
    al = api.Aligner(ix, api.default_opts(lib, batch_pairs=batch, host_threads=threads, single_end=1 if args.se else 0, **okw), max_pairs=call, debug=True, tuning=tuning)
    oa = ob.OracleAligner(pre, ob.default_opts(**okw))
    if args.se:
        api.align_stream(al, list(rb.names), rb.seq[:1], rb.qual[:1], rb.lens[:1], call, d + "/g.st", d + "/g.sam")
        oa.align_se(list(rb.names), rb.seq[0], rb.qual[0], rb.lens[0], d + "/o.st", d + "/o.sam", batch=batch)
    else:
        api.align_stream(al, rb.names, rb.seq, rb.qual, rb.lens, call, d + "/g.st", d + "/g.sam", packed=packed)
        oa.align(rb.names, rb.seq, rb.qual, rb.lens, d + "/o.st", d + "/o.sam", batch=batch)
    diffs = [x for x in ob.diff_stage_files(d + "/o.st", d + "/g.st")]
    same = filecmp.cmp(d + "/o.sam", d + "/g.sam", shallow=False)
    ok = not diffs and same
    bad += 0 if ok else 1
    print("seed %3d %-8s%s len %3d n %5d call %5d thr %d %s retries %d  %.1fs  %s" % (seed, mode, "+pk" if packed else "   ", read_len, n, call, threads, "OK  " if ok else "FAIL", al.stats()["tier_retries"], time.time() - t0,
                                                                    "" if ok else (str(refkw) + str(readkw) + str(okw) + " " + str(diffs[:3]))), flush=True)
    al.close(); ix.close(); oa.close()
sys.exit(1 if bad else 0)
