#!/usr/bin/env python3
"""bench.py -- paired 150 bp reads aligned per second through the whole `align` hot path on MI355X.

A "step" is one pass of the hot path (encode+trim+filter -> gap search -> SA -> pairing -> mate SW -> refine/MD ->
records) over one batch of synthetic input: one call on each of the --ctxs concurrent streams, a call being --pairs synthetic read
pairs (16 reference batches of 262,144, the reference's READ_BUFFER_SIZE, src/BwtMapper.h:36) against the 10k-marker reduced
reference of BASELINE.json configs[1] (1000 long + 9000 short flanks, l_pac 6,510,000).

Boundary of `value`: packed read batches in pinned host memory in -> result records in host memory out
(fq_packed_prefetch + fq_align_packed: the 48 bytes of filter keys per pair cross PCIe for every pair, the reads themselves
only for surviving pairs; the next batch's upload runs under this batch's kernels).  `resident_value` is the same workload
with the ASCII batch already in HBM (fq_batch_upload outside, fq_align_resident inside): the device-side rate.  `ontarget`
is a short second leg on the on-target-only mix, where the Occ-lookup (gap search) kernel fills the device.  Multi-GPU:
one process per GPU, reads shard by batch, no data-path collective ("weak" scaling); the only
collectives are the barrier and the MAX over ranks of the elapsed time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import subprocess
import threading
import time

# One hardware queue per HIP stream: the library gives every alignment context its own stream (plus two shared copy streams per
# device), and the runtime's default of four hardware queues would make sixteen contexts share queues (a 15 ms persistent
# search kernel then blocks another stream's filter kernel).  Sixteen context streams + two copy streams + the null stream need
# more than sixteen queues: with exactly 16 the last two contexts shared queues and finished 25 % after the others; 32 or more
# oversubscribe the hardware queues and halve the throughput.  Must be set before the HIP runtime initialises: main() asks the library
# for it (fq_runtime_configure) before anything touches the device; the variable is set here too because `import torch` may come first.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_PEAK_GBS = 6290.0     # ... and what a streaming copy reaches on the device (the guide's measured figure): fractions are stated against both
K_NAMES = ("prep", "width", "gap", "sa", "sw", "refine")
K_WIDTH_KERNEL, K_SA_KERNEL, K_SW_KERNEL, K_REFINE_KERNEL, K_MD_KERNEL, K_REC_KERNEL = 9, 10, 11, 12, 13, 14
KERNEL_IDS = {"prep": 6, "gap_full": 7, "gap_nogap": 8, "width": 9, "sa": 10, "sw": 11, "refine": 12, "md": 13, "records": 14}
K_PREP_KERNEL, K_GAP_KERNEL, K_GAP_NOGAP = 6, 7, 8     # single-kernel timings (kernel begin/end timestamps via hipExtLaunchKernelGGL events);
#   7 = the full search kernels (one read per lane / per wavefront), 8 = the first round of a device-filling launch (search without gap children)
STRIDE_PAD = 16                # ASCII rows are padded to 16 bytes: the resident filter kernel loads rows with 16-byte vector loads


def host_budget_child(args):
    """The same run inside a rank's share of the host, 2 CPUs (an 8-GPU node's 16 CPUs over 8 ranks): a child process of this script, pinned to 2 CPUs
    before it touches the device (sched_setaffinity: every thread it and the HIP runtime start inherit it) and the library told so
    (FASTQUICK_HOST_CPUS=2: what LOCAL_WORLD_SIZE=8 works out to on 16 CPUs); headline and on-target legs.  It runs BEFORE this process
    initialises the device: two processes' worth of hardware queues on one device slow both (the library asks for 20 each; 32 or more
    oversubscribe them), and a child started behind the parent's legs measured 0.82-0.93 of what it measures alone."""
    import subprocess
    hb_cpus = 2
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(4, min(args.steps, 16))), "--warmup", str(max(2, min(args.warmup, 8))),
           "--pairs", str(args.pairs), "--ctxs", str(args.ctxs), "--markers", str(args.markers), "--mix", args.mix, "--read-len", str(args.read_len),
           "--boundary", args.boundary, "--host-cpus", str(hb_cpus), "--no-resident", "--no-front-end", "--no-cpu-baseline", "--no-host-budget",
           "--ontarget-tput-ctxs", "0", "--workdir", args.workdir] + (["--no-ontarget"] if args.no_ontarget else []) + (["--tune", args.tune] if args.tune else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    hb = {"cpus": hb_cpus, "how": "child process pinned to %d CPUs (sched_setaffinity) with FASTQUICK_HOST_CPUS=%d, run before this process touched the device; same workload, %d streams" % (hb_cpus, hb_cpus, args.ctxs),
          "wall_s": round(time.perf_counter() - t0, 1)}
    line = [l for l in run.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if run.returncode == 0 and line:
        hb["child"] = json.loads(line[-1])
    else:
        hb["error"] = run.stderr.decode(errors="replace")[-500:]
    return hb


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; one step = one call on every concurrent stream (--ctxs)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=16 * 262144,
                    help="pairs per call: one call carrying this many pairs = pairs/262144 reference batches (READ_BUFFER_SIZE)")
    ap.add_argument("--ctxs", type=int, default=16,
                    help="alignment contexts (independent FASTQ streams) driven concurrently, one host thread + HIP streams each")
    ap.add_argument("--markers", type=int, default=10000)
    ap.add_argument("--mix", choices=("wgs", "ontarget"), default="wgs",
                    help="wgs: on-target fraction l_pac/3.1e9 (SURVEY 8d); ontarget: every pair from a marker flank")
    ap.add_argument("--read-len", type=int, default=150, help="150 (cfg 1-3) or 76 (cfg 4, exome: indel-rich reads)")
    ap.add_argument("--boundary", choices=("host", "resident"), default="host",
                    help="boundary of `value`: host = packed pinned batches in, records out (SURVEY 8d); resident = inputs already in HBM")
    ap.add_argument("--no-resident", action="store_true", help="skip the resident (inputs in HBM) leg of a host-boundary run")
    ap.add_argument("--no-ontarget", action="store_true", help="skip the short on-target leg of a wgs run")
    ap.add_argument("--ontarget-pairs", type=int, default=0, help="pairs per call of the on-target leg (0: the call shape of the headline leg, --pairs; the search "
                    "stage's second round is a fixed tail plus a term in the number of reads, so its roofline fraction grows with the call: DESIGN.md 9)")
    ap.add_argument("--ontarget-ctxs", type=int, default=2)
    ap.add_argument("--ontarget-steps", type=int, default=8)
    ap.add_argument("--ontarget-tput-ctxs", type=int, default=16, help="streams of the on-target throughput leg (1,048,576 pairs per call; 0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-shard", type=int, default=4, help="with --gpus N > 1: also time ONE FASTQ stream sharded over the ranks by reference batch "
                    "(dist.StreamShard; this many batches of 262,144 pairs per rank, 0: skip)")
    ap.add_argument("--no-front-end", action="store_true", help="skip the front-end leg (packer, FASTQ reader, command line end to end)")
    ap.add_argument("--front-end-pairs", type=int, default=1 << 20, help="pairs of the FASTQ files of the front-end leg")
    ap.add_argument("--front-end-copies", type=int, default=64, help="the steady-state command-line run reads those files concatenated this many times (0 / 1: skip)")
    ap.add_argument("--no-cli-ontarget", action="store_true", help="skip front_end.cli_e2e_ontarget (the command line on on-target FASTQ files)")
    ap.add_argument("--cli-ontarget-pairs", type=int, default=1 << 19, help="distinct on-target pairs of that leg's FASTQ files")
    ap.add_argument("--cli-ontarget-copies", type=int, default=64, help="... concatenated this many times (BGZF members concatenate)")
    ap.add_argument("--no-reference-baseline", action="store_true", help="skip cpu_baseline.reference (the real reference, oracle/_ref, timed on the same input)")
    ap.add_argument("--reference-sample-pairs", type=int, default=1 << 20, help="pairs of the WGS-mix sample the real reference aligns (a sixteenth of it for the on-target mix)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="pairs per slice for the CPU baseline (0 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="independent streams (threads) of the CPU baseline")
    ap.add_argument("--tune", default="", help="key=value,... passed to fq_ctx_set_tuning on every context (experiments)")
    ap.add_argument("--workdir", default=os.environ.get("FQ_BENCH_DIR", "/tmp/fq_bench"))
    ap.add_argument("--host-cpus", type=int, default=0, help="run inside a host budget of this many CPUs: the process is pinned to them (sched_setaffinity) and the library told its share "
                    "(FASTQUICK_HOST_CPUS) before anything touches the device; 0: the whole allowance.  The default run times itself once more this way (host_budget)")
    ap.add_argument("--no-host-budget", action="store_true", help="skip the host_budget leg (the run repeated on 2 CPUs)")
    args = ap.parse_args()
    if args.host_cpus > 0:
        allowed = sorted(os.sched_getaffinity(0))
        first = int(os.environ.get("FQ_BENCH_PIN_FIRST", "-1"))
        if first < 0:
            first = min(8, max(0, len(allowed) - args.host_cpus))      # (not the box's first CPUs: those take its interrupts)
        cpus = allowed[first:first + args.host_cpus]
        os.sched_setaffinity(0, set(cpus))                      # (inherited by every thread started from here on: the HIP runtime's, the library's, Python's)
        os.environ["FASTQUICK_HOST_CPUS"] = str(args.host_cpus)
    hb_result = None
    # (under a profiler the preloaded library has initialised the GPU already and a child would write into the same output directory: no host-budget leg there)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ)
    if args.host_cpus == 0 and not args.no_host_budget and not profiled and args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        hb_result = host_budget_child(args)          # (before anything here touches the device)

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    from fastquick_amd import api, synth
    from fastquick_amd import dist as fqd
    api.load_library().fq_runtime_configure(20, 1)   # hardware queues for the contexts' streams, sleeping waits (include/fastquick_amd.h)
    rank, local_rank, world = fqd.init("nccl")      # RCCL; one process per GPU
    if world > 1 and args.stream_shard > 0 and args.mix == "wgs":
        # the gloo group the stream tokens of the stream-shard leg travel over: new_group is a collective, so every rank creates it HERE,
        # unconditionally and before anything rank-specific can fail (created inside that leg's try block, a rank that failed before it
        # left the others waiting in the collective)
        fqd.token_group()
    tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",") if kv}

    # ---- workload (seeded synthetic; the index is built once per node by rank 0) --------------------------------
    os.makedirs(args.workdir, exist_ok=True)
    pre = os.path.join(args.workdir, "m%d.FASTQuick.fa" % args.markers)
    n_long = args.markers // 10
    ref = synth.make_reference(n_markers=args.markers, n_long=n_long, seed=12345)
    if rank == 0 and not os.path.exists(pre + ".rsa"):
        ref.write_fasta(pre)
        api.build_index(pre)
    fqd.barrier()
    L = args.read_len
    stride = (L + STRIDE_PAD - 1) // STRIDE_PAD * STRIDE_PAD
    ix = api.Index(pre, device=local_rank)

    def reads_kw():
        if L >= 150:
            return {}
        # cfg 4 (exome, 2x76): shorter fragments, indel-rich (10 % indel reads)
        return dict(read_len=L, frag_mean=200, frag_sd=20, del_frac=0.05, ins_frac=0.05, indel_len_max=2)

    def make_batch(n_pairs, seed, on_frac):
        """Seeded synthetic batch.  Off-target pairs are i.i.d. random bases drawn on the GPU (fast), on-target
        pairs come from fastquick_amd.synth (fragments of the marker flanks with errors) at seeded random slots."""
        if on_frac >= 1.0:
            rb = synth.make_reads(ref, n_pairs, on_target=1.0, seed=seed, **reads_kw())
            seq = np.zeros((2, n_pairs, stride), dtype=np.uint8)
            qual = np.zeros((2, n_pairs, stride), dtype=np.uint8)
            seq[:, :, :L] = rb.seq
            qual[:, :, :L] = rb.qual
            return synth.ReadBatch(seq, qual, rb.lens, None)
        rng = np.random.default_rng(seed)
        n_on = int(rng.binomial(n_pairs, on_frac))
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
        seq = np.zeros((2, n_pairs, stride), dtype=np.uint8)
        step = 1 << 20
        for e in range(2):
            for a in range(0, n_pairs, step):
                b = min(n_pairs, a + step)
                codes = torch.randint(0, 4, (b - a, L), device="cuda", generator=g, dtype=torch.uint8)
                seq[e, a:b, :L] = lut[codes.long()].cpu().numpy()
        on = synth.make_reads(ref, max(n_on, 1), on_target=1.0, seed=seed + 1, **reads_kw())
        slots = np.sort(rng.choice(n_pairs, size=n_on, replace=False))
        seq[:, slots, :L] = on.seq[:, :n_on]
        qual = np.full((2, n_pairs, stride), ord("I"), dtype=np.uint8)
        lens = np.full((2, n_pairs), L, dtype=np.int32)
        return synth.ReadBatch(seq, qual, lens, None)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            fqd.barrier()
            torch.cuda.synchronize()

    def sum_stats(ctxs):
        agg = None
        for al in ctxs:
            s = al.stats()
            if agg is None:
                agg = s
            else:
                for k, v in s.items():
                    agg[k] = [a + b for a, b in zip(agg[k], v)] if isinstance(v, list) else (max(agg[k], v) if k.startswith("max_") else agg[k] + v)
        return agg

    def run_leg(mix, pairs, n_ctx, steps, warmup, boundary, seed0, keep=False, max_distinct=4):
        """One measured leg: n_ctx contexts (independent FASTQ streams), each driven by its own host thread; a step = one call
        of `pairs` pairs on every context.  boundary "host": every call takes a packed batch from pinned host memory (the next
        one prefetched under it) and returns records in host memory; "resident": the ASCII batch was uploaded before."""
        on_frac = 1.0 if mix == "ontarget" else ref.l_pac / 3.1e9
        # distinct host batches are large (2.7 GB of ASCII each at the default size): at most four per rank, taken round-robin
        n_distinct = min(n_ctx, max_distinct) if boundary == "resident" else min(max(n_ctx, 2), max_distinct)
        distinct = [make_batch(pairs, seed0 + 17 * rank + b, on_frac) for b in range(n_distinct)]
        packs = None
        ctxs = [api.Aligner(ix, max_pairs=pairs, tuning=tuning) for _ in range(n_ctx)]
        if boundary == "host":
            packs = [api.HostPacked(b.seq, b.qual, b.lens, None) for b in distinct]
            for b in distinct:
                b.seq = None          # the ASCII bases are not needed any more (qualities stay: the packed batch refers to them)
        else:
            for c, al in enumerate(ctxs):
                b = distinct[c % n_distinct]
                al.upload(b.seq, b.qual, b.lens, None)   # inputs resident in HBM

        t_run0 = [0.0]
        cur_of = [c % n_distinct for c in range(n_ctx)]     # each context walks the batches round-robin, across run_steps calls

        def run_steps(k_steps):
            if boundary == "host":
                # the streams' driver loops (prefetch the next batch, align this one) run inside the library: fq_stream_run, one library thread per
                # stream asleep while the device works; no Python thread per stream
                t_run0[0] = time.perf_counter()
                surv = api.stream_run(ctxs, [packs] * n_ctx, k_steps, first=list(cur_of), prefetch_beyond=True)
                for c in range(n_ctx):
                    cur_of[c] = (cur_of[c] + k_steps) % n_distinct
                return sum(surv)
            recs = [0] * n_ctx
            errs = []

            def worker(c):
                try:
                    al = ctxs[c]
                    for _ in range(k_steps):
                        recs[c] += al.align_resident().n_survivors
                except Exception as e:      # noqa: BLE001
                    errs.append(e)
            t_run0[0] = time.perf_counter()
            th = [threading.Thread(target=worker, args=(c,)) for c in range(n_ctx)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
            return sum(recs)

        run_steps(1)                    # set-up: the first call of a context sizes its device buffers
        if warmup > 0:
            run_steps(warmup)
        for al in ctxs:
            al.reset_stats()
        sync_all()
        t0 = time.perf_counter()
        n_records = run_steps(steps)
        sync_all()
        elapsed = fqd.max_over_ranks(time.perf_counter() - t0)
        agg = sum_stats(ctxs)
        calls = steps * n_ctx
        leg = {"elapsed": elapsed, "calls": calls, "agg": agg, "n_records": n_records, "on_frac": on_frac, "pairs": pairs, "n_ctx": n_ctx, "n_distinct": n_distinct,
               "value": pairs * calls * world / elapsed}
        # the same kernels alone on the device (one stream, after the timed region): launch duration without other streams' kernels
        if n_ctx > 1:
            ctxs[0].reset_stats()
            for _ in range(3):
                if boundary == "host":
                    ctxs[0].align_packed(packs[cur_of[0]])
                else:
                    ctxs[0].align_resident()
            leg["solo"] = dict(ctxs[0].stats(), calls=3)
        if keep:
            leg["ctxs"], leg["packs"], leg["distinct"] = ctxs, packs, distinct
        else:
            for al in ctxs:
                al.close()
            for p in packs or []:
                p.free()
        return leg

    def prep_bytes(agg, pairs_total, boundary):
        per_read = 24.0 + 1.0 if boundary == "host" else 96.0 + 5.0      # packed: 3 k-mers in, 1 verdict byte out; ASCII: 96 bases in, length + verdict out
        return 64.0 * agg["filter_probes"] + per_read * 2 * pairs_total

    def gap_parts(st):
        """(name, summed ms, launches, algorithmic bytes) of the search stage and of its two kernels: a device-filling launch runs the search
        without gap children first (k_gap_nogap_stock / _lds) and the full search (k_gap_persist_stock / _lds, k_gap_coop) on what that leaves."""
        ms_f, ms_n = st["kernel_ms"][K_GAP_KERNEL], st["kernel_ms"][K_GAP_NOGAP]
        nl_f, nl_n = int(st["kernel_launches"][K_GAP_KERNEL]), int(st["kernel_launches"][K_GAP_NOGAP])
        b_all, b_n = 48.0 * st["gap_occ_touches"], 48.0 * st["gap_nogap_touches"]
        return [("gap", ms_f + ms_n, nl_f + nl_n, b_all), ("gap_nogap", ms_n, nl_n, b_n), ("gap_full", ms_f, nl_f, b_all - b_n)]

    def dp_parts(st):
        """(name, summed ms, launches, algorithmic bytes) of the kernels that are neither the filter nor the search: widths (48 B per Occ
        block touch, as the search), refinement (per task: the read's row, the 2-bit reference window, task + result + CIGAR) and
        MD / NM (per mapped read: the row, the window, the record in and out, the string).  The last two compute on chip; their
        byte figures say how far from the memory roofline such kernels sit, nothing else."""
        ref_b = (L + (L + 3) // 4 + 16 + 8 + 8) * float(st["refine_tasks"])
        md_b = (L + (L + 3) // 4 + 64 + 64 + 8) * float(st["md_reads"])
        return [("width", st["kernel_ms"][K_WIDTH_KERNEL], int(st["kernel_launches"][K_WIDTH_KERNEL]), 48.0 * st["width_occ_touches"]),
                ("refine", st["kernel_ms"][K_REFINE_KERNEL], int(st["kernel_launches"][K_REFINE_KERNEL]), ref_b),
                ("md", st["kernel_ms"][K_MD_KERNEL], int(st["kernel_launches"][K_MD_KERNEL]), md_b)]

    def device_ms(st, calls):
        """summed kernel time per call by kernel (begin / end timestamps of every launch: what rocprofv3 --kernel-trace sums)"""
        return {k: round(st["kernel_ms"][i] / max(1, calls), 4) for k, i in KERNEL_IDS.items()}

    def kernel_rooflines(agg, pairs_total, boundary):
        out = {}
        parts = [("prep", agg["kernel_ms"][K_PREP_KERNEL], int(agg["kernel_launches"][K_PREP_KERNEL]), prep_bytes(agg, pairs_total, boundary))] + gap_parts(agg) + dp_parts(agg)
        for kname, ms_sum, nl, byts in parts:
            if nl == 0 and kname != "prep" and kname != "gap":
                continue
            nl = max(1, nl)
            ms = ms_sum / nl
            gbs = (byts / nl) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out["fq_" + kname] = {"avg_launch_ms": round(ms, 4), "launches": nl, "alg_bytes_per_launch": round(byts / nl, 1),
                                  "alg_GBps": round(gbs, 2), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5), "frac_of_measured_copy_peak": round(gbs / HBM_COPY_PEAK_GBS, 5)}
        return out

    # ---- the headline leg ---------------------------------------------------------------------------------------------
    n_ctx = max(1, args.ctxs)
    main_leg = run_leg(args.mix, args.pairs, n_ctx, args.steps, args.warmup, args.boundary, 1000, keep=True)
    elapsed, calls, agg = main_leg["elapsed"], main_leg["calls"], main_leg["agg"]
    pairs_total = args.pairs * calls
    kms = agg["kernel_ms"]
    per_kernel = kernel_rooflines(agg, pairs_total, args.boundary)
    # Dominant kernel: (a) by rocprof's measure -- summed device time of the kernel over the timed region; (b) weighted by the share
    # of the device's resident lanes a launch can occupy (a search launch over the few thousand on-target reads of a WGS-like call
    # holds a few percent of the wavefront slots for as long as its longest search lasts; the filter kernel fills the device).
    # The roofline object is stated for (b); (a) is reported beside it.
    n_launch = [max(1, int(x)) for x in agg["kernel_launches"]]
    items = {"prep": 2.0 * pairs_total / n_launch[0], "width": 2.0 * agg["reads_searched"] / n_launch[1],
             "gap": float(agg["reads_searched"]) / n_launch[2], "sa": float(agg["sa_rows"]) / n_launch[3],
             "sw": float(agg["sw_tasks"]) / n_launch[4], "refine": float(agg["refine_tasks"]) / n_launch[5]}
    capacity = {"prep": 524288.0, "width": 524288.0, "gap": 262144.0, "sa": 524288.0, "sw": 256.0, "refine": 16384.0}   # resident work items
    share = {k: min(1.0, items[k] / capacity[k]) for k in items}
    dom = max(range(len(K_NAMES)), key=lambda k: kms[k] * share[K_NAMES[k]])
    dev_ms = device_ms(agg, calls)
    dom_by_time = max(dev_ms, key=lambda k: dev_ms[k])
    dname = K_NAMES[dom] if K_NAMES[dom] in ("prep", "gap") else "prep"
    pk = per_kernel["fq_" + dname]
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected separately,
    # tools/pmc_summarize.py), scaled by this run's units per launch; null when no measurement exists for this mix/kernel/boundary.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pmc = pmc[args.mix + ("_packed" if args.boundary == "host" else "")]["fq_" + dname]
        unit_count = {"prep": 2.0 * pairs_total, "gap": float(agg["reads_searched"])}[dname]
        traffic = round(pmc["bytes_per_unit"] * unit_count / pk["launches"], 1)
    except (OSError, KeyError, ValueError):
        pass
    alg_bytes = pk["alg_bytes_per_launch"] * pk["launches"]
    roofline = {"bound": "hbm", "kernel": "fq_" + dname, "achieved": pk["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(pk["alg_GBps"] / HBM_PEAK_GBS, 6), "peak_measured_copy": HBM_COPY_PEAK_GBS, "frac_of_measured_copy_peak": round(pk["alg_GBps"] / HBM_COPY_PEAK_GBS, 6), "traffic": traffic, "avg_launch_ms": pk["avg_launch_ms"],
                "alg_bytes_per_launch": pk["alg_bytes_per_launch"],
                "model": ("64 B x bitmap probes + 24 B of filter keys/read + 1 B/read out" if args.boundary == "host" else "64 B x bitmap probes + 96 B of bases/read + 5 B/read out")
                if dname == "prep" else "48 B x Occ block touches (reference block definition, SURVEY 8d)",
                "aggregate_achieved": round(alg_bytes / elapsed / 1e9, 3),   # all concurrent launches together, over the wall time
                "device_ms_per_call": dev_ms,
                # ONE dominant kernel, named in "kernel": the largest occupancy-weighted device time.  (By raw summed begin-to-end time
                # another kernel may lead -- it is named inside this string, not in a field of its own: a search launch over a few
                # thousand reads holds a few percent of the wave slots for as long as its longest search lasts.)
                "dominance": "kernel = argmax of summed device ms per call x share of resident-lane capacity a launch occupies: " +
                             ", ".join("%s %.2f x %.2f" % (K_NAMES[k], kms[k] / calls, share[K_NAMES[k]]) for k in range(len(K_NAMES))) +
                             "; largest raw summed device time: fq_" + dom_by_time}
    if "solo" in main_leg:
        s1 = main_leg["solo"]
        if dname == "prep":
            nl1, ms_sum1, byts1 = int(s1["kernel_launches"][K_PREP_KERNEL]), s1["kernel_ms"][K_PREP_KERNEL], prep_bytes(s1, args.pairs * 3, args.boundary)
        else:
            _, ms_sum1, nl1, byts1 = gap_parts(s1)[0]
        nl1 = max(1, nl1)
        ms1 = ms_sum1 / nl1
        if ms1 > 0:
            roofline["solo_avg_launch_ms"] = round(ms1, 4)
            roofline["solo_achieved"] = round(byts1 / nl1 / (ms1 * 1e-3) / 1e9, 3)
            roofline["solo_frac"] = round(byts1 / nl1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)

    value = main_leg["value"]
    out = {
        "metric": "paired_%dbp_reads_aligned_per_sec" % L, "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%dk-marker reduced ref (l_pac %d), %d x 2x%dbp pairs/step = %d concurrent streams x one call of %d pairs (%d reference batches of 262144), %s mix (on-target %.4f)"
                   % (args.markers // 1000, ref.l_pac, args.pairs * n_ctx, L, n_ctx, args.pairs, (args.pairs + 262143) // 262144, args.mix, main_leg["on_frac"]),
                   "boundary": ("packed read batches in pinned host memory -> result records in host memory (H2D and D2H inside the timed region; next batch's upload overlapped)"
                                if args.boundary == "host" else "inputs resident in HBM -> result records in host memory"),
                   "pairs_per_step": args.pairs * n_ctx, "pairs_per_call": args.pairs, "calls_per_step": n_ctx,
                   "markers": args.markers, "mix": args.mix, "read_len": L, "concurrent_streams": n_ctx,
                   "sharding": "batches per rank, no data-path collective"},
        "roofline": roofline,
        "kernel_rooflines": per_kernel,
        "pcie": {"h2d_bytes_per_pair": round(agg["h2d_bytes"] / pairs_total, 2), "d2h_bytes_per_pair": round(agg["d2h_bytes"] / pairs_total, 3),
                 "h2d_GBps": round(agg["h2d_bytes"] * world / elapsed / 1e9 / world, 2), "d2h_GBps": round(agg["d2h_bytes"] / elapsed / 1e9, 3)},
        "stage_ms_per_call": {K_NAMES[k]: round(kms[k] / calls, 4) for k in range(len(K_NAMES))},
        "host_ms_per_call": round(agg["host_ms_total"] / calls, 3), "wall_ms_per_call": round(agg["wall_ms_total"] / calls, 3),
        "host_cpu_ms_per_call": round(agg["host_cpu_ms"] / calls, 3), "device_wait_ms_per_call": round(agg["device_wait_ms"] / calls, 3),
        "survivor_pairs_per_call": round(main_leg["n_records"] / calls, 1),
        "work_per_call": {k: round(agg[k] / calls, 1) for k in ("filter_probes", "occ_block_touches", "gap_occ_touches", "gap_nogap_touches", "stack_pops",
                                                                     "stack_pushes", "sa_rows", "reads_searched", "sw_tasks", "refine_tasks", "tier_retries", "reads_over_4k_pops")},
        "max_pops_per_read": agg["max_pops_per_read"], "max_wave_trips": agg["max_wave_trips"],
        "gap_wave_trips_per_call": round(agg["wave_trips"] / calls, 1), "gap_lane_trips_per_call": round(agg["lane_trips"] / calls, 1),
    }

    # ---- what a multi-GPU run hands to rank 0 (outside the timed region): the SAM text of every rank's last call, in rank order, and
    #      the summed stream counters (the reference's FileStatCollector sums); RCCL all_gather / all_reduce, a few MB per rank.
    #      Whether to gather is decided from the run's arguments, the same on every rank (a per-rank decision could leave some
    #      ranks outside the collective).
    ctxs = main_leg["ctxs"]
    if args.mix == "wgs":
        last = ctxs[0].result
        parts = fqd.gather_bytes_to_rank0(ctxs[0].sam_text())
        tot = fqd.sum_counters({"pairs": int(last.n_pairs), "survivor_pairs": int(last.n_survivors), "both_filtered": int(last.n_both_filtered),
                                "both_unmapped": int(last.n_both_unmapped), "bases": int(last.n_bases)})
        if rank == 0:
            out["gather"] = {"ranks": len(parts), "backend_world_size": fqd.world_size(), "sam_bytes": [len(p) for p in parts], "counters_sum": tot}
            assert len(parts) == world and fqd.world_size() == args.gpus, "RCCL did not see every rank"
    cpu_batch = main_leg["distinct"][0]
    for al in ctxs:
        al.close()

    # ---- the other partitioning: ONE stream over the ranks, sharded by reference batch; the stream state (drand48, last_ii,
    #      (k,l) cache) goes from the owner of batch b to the owner of b + 1 around the short order-dependent part of each call.  No
    #      data-path collective: point-to-point tokens of a few hundred bytes.  Reported beside `value`, never instead of it.
    if world > 1 and args.stream_shard > 0 and args.mix == "wgs":
        Bs = 262144
        nb = args.stream_shard * world
        shard_err, el, recs, sp, al = "", 0.0, 0, {}, None
        try:      # (an extra leg: whatever goes wrong in it must not cost the run its headline line -- every rank reaches the barrier below)
            al = api.Aligner(ix, max_pairs=Bs, tuning=tuning)
            sh = fqd.StreamShard(al, rank, world)
            sh.n_batches = nb
            mine = [b for b in range(nb) if sh.owns(b)]
            for b in mine:
                bt = make_batch(Bs, 5000 + b, main_leg["on_frac"])
                sp[b] = api.HostPacked(bt.seq, bt.qual, bt.lens, None)
            sync_all()
            t0 = time.perf_counter()
            for b in mine:
                sh.batch_index = b
                recs += al.align_packed(sp[b]).n_survivors
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        except Exception as e:      # noqa: BLE001
            shard_err = repr(e)[:300]
        sync_all()
        el = fqd.max_over_ranks(el)
        tot = fqd.sum_counters({"survivor_pairs": recs, "failed_ranks": 1 if shard_err else 0})
        if rank == 0:
            out["stream_shard"] = {"value": round(nb * Bs / el, 1) if el > 0 and not tot["failed_ranks"] else None, "unit": "pairs/s", "batches": nb, "pairs_per_batch": Bs,
                                   "ranks": world, "s": round(el, 4), "survivor_pairs": tot["survivor_pairs"], "failed_ranks": tot["failed_ranks"], "error": shard_err,
                                   "note": "one FASTQ stream, batches dealt round-robin, state tokens over a gloo group"}
        for p_ in sp.values():
            p_.free()
        if al is not None:
            al.close()
    cpu_seq = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_seq = cpu_batch.seq if cpu_batch.seq is not None else make_batch(args.pairs, 1000 + 17 * rank, main_leg["on_frac"]).seq
    for p in main_leg["packs"] or []:
        p.free()
    main_leg["ctxs"] = main_leg["packs"] = None

    # ---- the device-side rate of the same workload (inputs already in HBM), fewer steps --------------------------------------------
    if args.boundary == "host" and not args.no_resident:
        rs = max(2, args.steps // 4)
        leg = run_leg(args.mix, args.pairs, n_ctx, rs, max(1, min(args.warmup, 2)), "resident", 1000)
        out["resident_value"] = round(leg["value"], 1)
        out["resident"] = {"steps": rs, "ms_per_step": round(1e3 * leg["elapsed"] / rs, 3),
                           "kernel_rooflines": kernel_rooflines(leg["agg"], args.pairs * leg["calls"], "resident")}

    # ---- on-target leg: every pair from a marker flank, one device-filling search launch per call (the >= 40 % criterion of
    #      BASELINE.json is about this kernel on this mix) -------------------------------------------------------------------------------
    if args.mix == "wgs" and not args.no_ontarget:
        if args.ontarget_pairs <= 0:
            args.ontarget_pairs = args.pairs
        leg = run_leg("ontarget", args.ontarget_pairs, args.ontarget_ctxs, args.ontarget_steps, 1, args.boundary, 3000, max_distinct=2)
        a2 = leg["agg"]
        out["ontarget"] = {"value": round(leg["value"], 1), "unit": "pairs/s", "pairs_per_call": args.ontarget_pairs, "concurrent_streams": args.ontarget_ctxs,
                           "steps": args.ontarget_steps, "ms_per_step": round(1e3 * leg["elapsed"] / args.ontarget_steps, 3),
                           "kernel_rooflines": kernel_rooflines(a2, args.ontarget_pairs * leg["calls"], args.boundary),
                           "stage_ms_per_call": {K_NAMES[k]: round(a2["kernel_ms"][k] / leg["calls"], 3) for k in range(len(K_NAMES))},
                           "device_ms_per_call": device_ms(a2, leg["calls"]),
                           "distinct_batches": leg["n_distinct"],
                           "host_ms_per_call": round(a2["host_ms_total"] / leg["calls"], 3),
                           "host_cpu_ms_per_call": round(a2["host_cpu_ms"] / leg["calls"], 3), "device_wait_ms_per_call": round(a2["device_wait_ms"] / leg["calls"], 3),
                           "reads_searched_per_call": round(a2["reads_searched"] / leg["calls"], 1),
                           "stack_pops_per_read": round(a2["stack_pops"] / max(1, a2["reads_searched"]), 1),
                           "occ_touches_per_read": round(a2["gap_occ_touches"] / max(1, a2["reads_searched"]), 1)}
        try:      # HBM-side bytes of the search stage from the committed PMC passes (FETCH_SIZE + WRITE_SIZE per searched read)
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = "ontarget" + ("_packed" if args.boundary == "host" else "")
            key4 = key + "_4m"      # the pass made at the headline call shape (4,194,304 pairs per call); smaller calls use the 1 M-pair pass
            if args.ontarget_pairs >= 4194304 and key4 in pmc_all:
                key = key4
            pmc2 = pmc_all[key]["fq_gap"]
            out["ontarget"]["kernel_rooflines"]["fq_gap"]["traffic_source"] = "profiles/pmc_traffic.json[%s]: FETCH_SIZE + WRITE_SIZE per searched read, measured at %d reads per launch" % (key, int(pmc2["units_per_launch"]))
            out["ontarget"]["kernel_rooflines"]["fq_gap"]["traffic_per_call"] = round(pmc2["bytes_per_unit"] * a2["reads_searched"] / leg["calls"], 1)
            out["ontarget"]["kernel_rooflines"]["fq_gap"]["alg_bytes_per_call"] = round(48.0 * a2["gap_occ_touches"] / leg["calls"], 1)
        except (OSError, KeyError, ValueError):
            pass
        if "solo" in leg:     # the search kernels with nothing else on the device
            s1 = leg["solo"]
            solo = {}
            for kname, ms_sum, nl, byts in gap_parts(s1):
                if nl and ms_sum > 0:
                    g1 = byts / (ms_sum * 1e-3) / 1e9
                    solo["fq_" + kname] = {"avg_launch_ms": round(ms_sum / nl, 4), "launches_per_call": round(nl / max(1, s1["calls"]), 2) if "calls" in s1 else None,
                                           "alg_GBps": round(g1, 2), "frac_of_hbm_peak": round(g1 / HBM_PEAK_GBS, 5), "frac_of_measured_copy_peak": round(g1 / HBM_COPY_PEAK_GBS, 5)}
            if "fq_gap" in solo:
                out["ontarget"]["gap_solo"] = dict(solo["fq_gap"], kernels={k: v for k, v in solo.items() if k != "fq_gap"})
        # the same mix as a throughput job: many streams of ordinary calls, so that one stream's host phases (main-hit choice in read
        # order, insert-size inference, record assembly) run under the kernels of the others
        if args.ontarget_tput_ctxs > 0:
            leg = run_leg("ontarget", 1 << 20, args.ontarget_tput_ctxs, 2, 1, args.boundary, 3000, max_distinct=2)
            out["ontarget"]["throughput"] = {"value": round(leg["value"], 1), "unit": "pairs/s", "pairs_per_call": 1 << 20, "concurrent_streams": args.ontarget_tput_ctxs,
                                             "steps": 2, "ms_per_step": round(1e3 * leg["elapsed"] / 2, 3),
                                             "host_ms_per_call": round(leg["agg"]["host_ms_total"] / leg["calls"], 3)}

    # ---- host budget: measured by a child process before this one touched the device (below, host_budget_child)
    if hb_result is not None:
        hb = dict(hb_result)
        ch = hb.pop("child", None)
        if ch is not None:
            hb.update({"value": ch["value"], "unit": ch["unit"], "ratio_to_value": round(ch["value"] / out["value"], 4), "steps": ch["steps"], "ms_per_step": ch["ms_per_step"],
                       "host_cpu_ms_per_call": ch.get("host_cpu_ms_per_call"), "wall_ms_per_call": ch.get("wall_ms_per_call"), "device_wait_ms_per_call": ch.get("device_wait_ms_per_call")})
            if "ontarget" in ch and "ontarget" in out:
                hb["ontarget"] = {"value": ch["ontarget"]["value"], "ratio_to_ontarget_value": round(ch["ontarget"]["value"] / out["ontarget"]["value"], 4),
                                  "ms_per_step": ch["ontarget"]["ms_per_step"], "host_cpu_ms_per_call": ch["ontarget"].get("host_cpu_ms_per_call")}
        out["host_budget"] = hb

    # ---- front end: what feeds the packed boundary, measured on this box's host cores beside `value` -----------------
    #   pack_pairs_per_s              fq_pack_reads_into on the headline call's batch (ASCII rows -> packed batch in reused pinned storage)
    #   tokenise_inflate_pairs_per_s  two BGZF FASTQ files -> ASCII rows (fq_fastq_read: member-parallel inflate + tokeniser), both files at once
    #   cli_e2e_pairs_per_s           the command line on those files: FASTQ -> SAM text, whole-process wall time (index staging included)
    if rank == 0 and world == 1 and not args.no_front_end:
        if cpu_batch.seq is None:     # (the headline leg dropped its ASCII rows once they were packed)
            cpu_batch.seq = cpu_seq if cpu_seq is not None else make_batch(args.pairs, 1000 + 17 * rank, main_leg["on_frac"]).seq
        cores = os.cpu_count() or 1
        usable = int(api.load_library().fq_host_cpus())     # (the container's CPU quota, not the hardware threads it shows)
        fe = {"cores": cores, "cpus_usable": usable}
        pt = max(2, min(2 * usable, 64))
        hp = api.HostPacked(cpu_batch.seq, cpu_batch.qual, cpu_batch.lens, None, threads=pt)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            hp.repack(pt)
            best = min(best, time.perf_counter() - t0)
        hp.free()
        fe["pack_pairs_per_s"] = round(args.pairs / best, 1)
        fe["pack"] = {"threads": pt, "pairs": args.pairs, "ms": round(1e3 * best, 2), "read_len": L}
        nfe = min(args.front_end_pairs, args.pairs)
        fdir = os.path.join(args.workdir, "front_end")
        os.makedirs(fdir, exist_ok=True)
        paths = [os.path.join(fdir, "reads_%d.fq.gz" % (e + 1)) for e in range(2)]
        t0 = time.perf_counter()
        # qualities as a NovaSeq writes them (four bins), so that the files inflate at a realistic rate (the batches above carry a constant)
        qual_fe = np.frombuffer(b"F:,#", dtype=np.uint8)[np.random.default_rng(99).choice(4, size=(2, nfe, L), p=[0.7, 0.15, 0.1, 0.05])]
        text_bytes = sum(synth.write_fastq_uniform(cpu_batch.seq[e, :nfe], qual_fe[e], L, paths[e], threads=pt) for e in range(2))
        fe["fastq_files"] = {"pairs": nfe, "text_bytes": text_bytes, "file_bytes": sum(os.path.getsize(p) for p in paths), "container": "BGZF (zlib level 1)", "qualities": "4 bins",
                             "write_s": round(time.perf_counter() - t0, 1)}
        rows = [(np.zeros((nfe, stride), np.uint8), np.zeros((nfe, stride), np.uint8), np.zeros(nfe, np.int32), np.zeros((nfe, 64), np.uint8)) for _ in range(2)]
        for r_ in rows:        # (pages touched before the timed region: the command line reuses its buffers, too)
            for a_ in r_:
                a_.fill(1)
        rt = max(1, pt // 2)
        got = [0, 0]

        def read_file(e):
            f = api.FastqFile(paths[e], threads=rt, stride=stride, name_stride=64)
            got[e] = f.read_into(*rows[e])
            f.close()
        t0 = time.perf_counter()
        th = [threading.Thread(target=read_file, args=(e,)) for e in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        assert got == [nfe, nfe] and np.array_equal(rows[0][0][:, :L], cpu_batch.seq[0, :nfe, :L]) and np.array_equal(rows[1][1][:, :L], qual_fe[1])
        fe["host_reader_pairs_per_s"] = round(nfe / dt, 1)        # (round 4's tokenise_inflate_pairs_per_s: the front end on the host's threads)
        fe["host_reader"] = {"threads_per_file": rt, "s": round(dt, 3), "text_GBps": round(text_bytes / dt / 1e9, 3),
                             "inflate": "zlib" if os.environ.get("FASTQUICK_ZLIB_INFLATE", "0") not in ("", "0") else "fq_inflate.h (zlib for members it refuses)"}
        # ---- a single-member gzip file (`gzip -6`: what is not bgzip'ed): the stream decoder on the reader's producer thread (fq_fastq.cpp,
        #      fill_gz_stream) against zlib's gzread (FASTQUICK_ZLIB_INFLATE=1), 8 threads
        try:
            n_gz = min(nfe, 400000)
            gz_path, txt_path = os.path.join(fdir, "single_member.fq.gz"), os.path.join(fdir, "single_member.fq")
            gz_text = synth.write_fastq_uniform(cpu_batch.seq[0, :n_gz], qual_fe[0, :n_gz], L, txt_path, bgzf=False)
            subprocess.check_call("gzip -6 -c %s > %s" % (txt_path, gz_path), shell=True)
            os.remove(txt_path)
            gz_res = {"reads": n_gz, "threads": 8, "container": "one gzip member, level 6", "text_bytes": gz_text, "file_bytes": os.path.getsize(gz_path)}
            for mode in ("stream_decoder", "gzread"):
                if mode == "gzread":
                    os.environ["FASTQUICK_ZLIB_INFLATE"] = "1"
                try:
                    best = 1e9
                    for _ in range(3):
                        f_ = api.FastqFile(gz_path, threads=8, stride=stride, name_stride=64)
                        t0 = time.perf_counter()
                        got_ = f_.read_into(*rows[0])
                        best = min(best, time.perf_counter() - t0)
                        f_.close()
                        assert got_ == n_gz, got_
                finally:
                    os.environ.pop("FASTQUICK_ZLIB_INFLATE", None)
                gz_res[mode] = {"reads_per_s": round(n_gz / best, 1), "text_MBps": round(gz_text / best / 1e6, 1)}
            os.remove(gz_path)
            fe["gzip_single_member"] = gz_res
        except Exception as e:      # noqa: BLE001
            fe["gzip_single_member"] = {"error": repr(e)[:300]}
        del rows
        exe = os.path.join(ROOT, "fastquick_amd", "bin", "FASTQuick_amd")
        if os.path.exists(exe):
            cmd = [exe, "align", "--index_prefix", pre[:-len(".FASTQuick.fa")], "--fastq_1", paths[0], "--fastq_2", paths[1], "--out_prefix", os.path.join(fdir, "out"),
                   "--sam_out", "--read_len", str(max(L, 151)), "--t", str(pt)]
            t0 = time.perf_counter()
            with open(os.path.join(fdir, "out.sam"), "wb") as so:
                run = subprocess.run(cmd, stdout=so, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            notes = [l for l in run.stderr.decode(errors="replace").splitlines() if "device time" in l or "consumers" in l or "index staged" in l or "reading (ms)" in l]
            fe["cli_e2e_pairs_per_s"] = round(nfe / dt, 1) if run.returncode == 0 else None
            fe["cli_e2e"] = {"rc": run.returncode, "pairs": nfe, "wall_s": round(dt, 2), "mix": args.mix, "output": "SAM text", "notices": notes}
            # ---- the same at a steady state: the two BGZF files concatenated `--front-end-copies` times (BGZF members concatenate), so that index
            #      staging (about a second: 3 GiB of filter bitmaps built on the device) is amortised; once to SAM text, once to BAM + the 13 QC files
            copies = args.front_end_copies
            if copies > 1 and run.returncode == 0:
                import shutil
                need = copies * sum(os.path.getsize(p_) for p_ in paths)
                free = shutil.disk_usage(fdir).free
                while copies > 1 and copies * sum(os.path.getsize(p_) for p_ in paths) > 0.5 * free:
                    copies //= 2
                big = [os.path.join(fdir, "big_%d.fq.gz" % (e + 1)) for e in range(2)]
                for e in range(2):
                    with open(big[e], "wb") as fo:
                        blob = open(paths[e], "rb").read()
                        blob = blob[:-28] if blob.endswith(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00") else blob   # (an end-of-file member in the middle is an empty member: harmless, dropped anyway)
                        for _ in range(copies):
                            fo.write(blob)
                synth.write_qc_inputs(pre, ref)
                synth.write_param(pre, ref, 1000)
                with open(pre + ".genome.fa.fai", "w") as fh:
                    fh.write("1\t%d\t3\t60\t61\n" % len(ref.genome))
                # ---- the front end on the device (fq_frontend_*): BGZF files -> filter keys, lengths, names of every read resident in HBM, with
                #      nothing of the text touched by the host (it reads compressed bytes and walks member headers).  tokenise_inflate_pairs_per_s =
                #      pairs / wall time from opening the two files to the last batch (batches released as they come: no alignment beside it)
                def run_front_end():
                    dfe = api.DeviceFrontEnd(big[0], big[1], batch_pairs=262144, chunk_pairs=16 * 262144, slot_mode=0, max_read_len=stride)
                    t0 = time.perf_counter()
                    got, t_first = 0, None
                    while True:
                        m_, b_ = dfe.next()
                        if m_ <= 0:
                            break
                        if t_first is None:
                            t_first = time.perf_counter() - t0
                        got += m_
                        dfe.release(b_)
                    dt = time.perf_counter() - t0
                    st = dfe.stats()
                    dfe.close()
                    assert m_ == 0 and got == nfe * copies, (m_, got)
                    return got, dt, t_first, st
                # the product's way: the next chunk's members are inflated beside this chunk's kernels (two streams)
                got_pairs, dt_fe, t_first, fst = run_front_end()
                fe["tokenise_inflate_pairs_per_s"] = round(got_pairs / dt_fe, 1)
                # ... and once more with the kernels one after the other (FASTQUICK_FE_OVERLAP=0), for each kernel group's time on its own
                os.environ["FASTQUICK_FE_OVERLAP"] = "0"
                try:
                    _, dt_solo, _, fso = run_front_end()
                finally:
                    del os.environ["FASTQUICK_FE_OVERLAP"]
                tb_, cb_ = float(fst["text_bytes"]), float(fst["comp_bytes"])

                def fe_roof(ms, byts, launches):
                    g_ = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                    return {"ms_total": round(ms, 2), "launch_groups": int(launches), "bytes": int(byts), "GBps": round(g_, 2), "frac_of_hbm_peak": round(g_ / HBM_PEAK_GBS, 5)}
                rows_ = 2.0 * got_pairs
                fe["tokenise_inflate"] = {"where": "device (k_inflate_bgzf + line index + record / key / slot kernels)", "pairs": got_pairs, "s": round(dt_fe, 3),
                                          "first_batch_s": round(t_first or 0.0, 3),
                                          "text_GBps": round(tb_ / dt_fe / 1e9, 3), "file_GBps": round(cb_ / dt_fe / 1e9, 3),
                                          "members": fst["members"], "members_left_to_the_host_decoder": fst["refused"], "chunks": fst["chunks"],
                                          # the host side of it: the producer thread's waits and the reader threads' time (wall ms; readers summed over the two files)
                                          "host_side_ms": {"producer_waited_for_compressed_bytes": round(fst["ms_wait_reader"], 1), "producer_waited_for_a_batch_slot": round(fst["ms_wait_slot"], 1),
                                                           "readers_pread": round(fst["ms_read"], 1), "readers_upload": round(fst["ms_upload"], 1)},
                                          "device_ms_overlapped": {"inflate": round(fst["ms_inflate"], 1), "lines": round(fst["ms_lines"], 1), "records": round(fst["ms_records"], 1), "slots": round(fst["ms_slots"], 1)},
                                          "one_after_the_other_s": round(dt_solo, 3),
                                          # bytes in + out of each kernel group / its summed device time with nothing beside it (HIP events on the front end's stream, second run)
                                          "kernel_rooflines": {
                                              "fq_inflate": dict(fe_roof(fso["ms_inflate"], tb_ + cb_, fso["inflate_launches"]), model="compressed bytes in + text bytes out; bound by the CU's one scalar unit (a symbol's decoding is a chain of scalar steps; profiles/round5_inflate_sq_counters_v3.txt), not by HBM"),
                                              "fq_lines": dict(fe_roof(fso["ms_lines"], 2 * tb_ + 4 * 4 * rows_, fso["chunks"]), model="text read twice (count, fill) + 4 B per line end out"),
                                              "fq_records": dict(fe_roof(fso["ms_records"], 16 * rows_ + 16 * rows_ + (150 + 12) * rows_ + 26 * rows_, fso["chunks"]), model="4 line ends in, 16 B record + 2 B length out; base line + name in, 24 B of filter keys out"),
                                              "fq_slots": dict(fe_roof(fso["ms_slots"], (16 + 96 + 96 + 12 + 12 + 16) * rows_, fso["chunks"]), model="record in, 96 B of the slot's bases in and out, name in, slot name + printed name out")}}
                steady = {"pairs": nfe * copies, "copies": copies, "requested_bytes": need}
                for label, extra in (("sam_out", ["--sam_out"]), ("bam_and_qc", [])):
                    cmd2 = [exe, "align", "--index_prefix", pre[:-len(".FASTQuick.fa")], "--fastq_1", big[0], "--fastq_2", big[1], "--out_prefix", os.path.join(fdir, "big_out"),
                            "--read_len", str(max(L, 151)), "--t", str(pt)] + extra
                    time.sleep(3.0)     # (a process started right behind another's exit waits for the driver to take back that one's device memory:
                    #                      the same command took 1.7-1.9 s right behind its predecessor and 1.42-1.49 s three seconds later)
                    t0 = time.perf_counter()
                    run2 = subprocess.run(cmd2, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                    dt2 = time.perf_counter() - t0
                    steady[label] = {"rc": run2.returncode, "wall_s": round(dt2, 2), "pairs_per_s": round(nfe * copies / dt2, 1) if run2.returncode == 0 else None,
                                     "notices": [l for l in run2.stderr.decode(errors="replace").splitlines() if "consumers" in l or "index staged" in l or "reading (ms)" in l][-3:]}
                for b_ in big:
                    os.remove(b_)
                # ---- the command line on ON-TARGET input (BASELINE.json cfg 5's regime; bin/FASTQuick_template.sh:474-481 is the run it stands for): every
                #      pair survives the filter, so the consumers see every pair -- SAM text and StatCollector's sums come out of the kernels of fq_emit.h
                if not args.no_cli_ontarget:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import cli_ontarget
                    ont = {}
                    for key, rl, n_, cp_ in (("2x150", 150, args.cli_ontarget_pairs, args.cli_ontarget_copies), ("2x76_indel_rich", 76, args.cli_ontarget_pairs, max(1, args.cli_ontarget_copies // 2))):
                        if args.markers != 10000 and rl != L:
                            continue
                        try:
                            ont[key] = cli_ontarget.measure(exe, pre, ref, os.path.join(fdir, "ont"), pairs=n_, copies=cp_, read_len=rl, threads=pt, repeats=2)
                        except Exception as e:      # noqa: BLE001
                            ont[key] = {"error": repr(e)[:300]}
                    fe["cli_e2e_ontarget"] = ont
                for ext in (".SelectedSite.vcf", ".dbSNP.subset.vcf", ".gc", ".param", ".genome.fa.fai"):      # (the other legs run without the QC consumer)
                    if os.path.exists(pre + ext):
                        os.remove(pre + ext)
                fe["cli_e2e_steady"] = steady
        out["front_end"] = fe

    # ---- CPU baseline: the oracle (a port) on a bounded sample of the same workload, rank 0, N=1 only -----------
    # Three geometries, a few seconds each:
    #   value    one stream under the reference's pool geometry (src/BwtMapper.cpp:1452-1537): the reader + filter and everything after
    #            the search on one thread, stage A sliced over --t workers (half per end); --t = every core of the box
    #   t4       the same with --t 4 (the reference's documented invocation)
    #   streams  T independent streams (one oracle context each, shared read-only index), the reference's --fq_list thread pool:
    #            the most the host can do, since nothing is serial across streams
    if cpu_seq is not None:
        import oracle_binding as ob
        b = cpu_batch
        n_cpu = args.cpu_sample_pairs or (131072 if args.mix == "wgs" else 8192)
        n_cpu = min(n_cpu, args.pairs)
        cores = os.cpu_count() or 1
        T = max(1, min(args.cpu_threads, cores, args.pairs // n_cpu))
        names = [b"r%09d" % i for i in range(n_cpu)]
        first = ob.OracleAligner(pre)
        oas = [first] + [ob.OracleAligner(pre, share=first) for _ in range(T - 1)]

        def one_stream(n_threads, budget):
            first.set_threads(n_threads)
            off, done, t0 = 0, 0, time.perf_counter()
            while time.perf_counter() - t0 < budget:
                if off + n_cpu > args.pairs:
                    off = 0
                first.align(names, cpu_seq[:, off:off + n_cpu], b.qual[:, off:off + n_cpu], b.lens[:, off:off + n_cpu], None, None, batch=n_cpu)
                done += n_cpu
                off += n_cpu
            first.set_threads(1)
            return done, time.perf_counter() - t0

        pool_t = max(2, min(args.cpu_threads, cores))
        d_pool, dt_pool = one_stream(pool_t, 6.0)
        d_t4, dt_t4 = one_stream(4, 5.0)
        done = [0] * T
        budget = 8.0

        def cpu_worker(t):
            off = (t * n_cpu) % max(1, args.pairs - n_cpu + 1)
            t_end = time.perf_counter() + budget
            while time.perf_counter() < t_end:
                if off + n_cpu > args.pairs:
                    off = 0
                oas[t].align(names, cpu_seq[:, off:off + n_cpu], b.qual[:, off:off + n_cpu], b.lens[:, off:off + n_cpu], None, None, batch=n_cpu)
                done[t] += n_cpu
                off += n_cpu * T
        t1 = time.perf_counter()
        th = [threading.Thread(target=cpu_worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t1
        cpu_model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "unknown") if os.path.exists("/proc/cpuinfo") else "unknown"
        out["cpu_baseline"] = {"value": round(d_pool / dt_pool, 1), "unit": "pairs/s", "cores": pool_t, "kind": "port", "cpu_model": cpu_model, "host_cpus": cores, "cpus_usable": int(api.load_library().fq_host_cpus()),
                               "sample": "%d pairs of the same %s-mix input through oracle/fq_oracle.c, one stream in batches of %d pairs under the "
                                         "reference's pool geometry (stage A sliced over --t %d workers as src/BwtMapper.cpp:1490-1513, the rest on "
                                         "one thread), %.1f s" % (d_pool, args.mix, n_cpu, pool_t, dt_pool),
                               "t4": {"value": round(d_t4 / dt_t4, 1), "cores": 4, "sample": "%d pairs, --t 4, %.1f s" % (d_t4, dt_t4)},
                               "streams": {"value": round(sum(done) / dt, 1), "cores": T,
                                           "sample": "%d pairs: %d independent streams (the --fq_list pool) x batches of %d pairs, one thread "
                                                     "per stream, shared index, %.1f s" % (sum(done), T, n_cpu, dt)}}
        for oa in oas[1:]:
            oa.close()
        first.close()
        # ---- the REAL reference beside the port: oracle/_ref/fq_ref_driver is the reference's own bwa_read_seq_with_hash_dev /
        #      bwa_cal_sa_reg_gap / bwa_cal_pac_pos_pe / bwa_paired_sw / bwa_refine_gapped / StatCollector / bwa_print_sam1 compiled where they lie; it
        #      travelled with the snapshot.  Same input as FASTQ text, batches of 262,144 pairs, stage A over the reference's pool geometry with
        #      --t 4 (its documented invocation) and --t <usable CPUs>; wall time from the first read to the last record (SAM text and StatCollector
        #      included, as PairEndMapper's loop has them).  Outside every timed region.
        drv = os.path.join(ROOT, "oracle", "_ref", "fq_ref_driver")
        if os.path.exists(drv) and not args.no_reference_baseline:
            try:
                rdir = os.path.join(args.workdir, "refrun")
                os.makedirs(rdir, exist_ok=True)
                rpre = os.path.join(rdir, "m%d.FASTQuick.fa" % args.markers)
                if not os.path.exists(rpre + ".rollhash.sparse"):
                    ref.write_fasta(rpre)
                    subprocess.check_call([drv, "index", rpre], stderr=subprocess.DEVNULL, cwd=rdir)      # the reference's own index builder (its filter tables as a list of set bits)
                    synth.write_qc_inputs(rpre, ref)
                n_ref = min(args.pairs, args.reference_sample_pairs if args.mix == "wgs" else args.reference_sample_pairs // 16)
                rfq = [os.path.join(rdir, "sample_%d.fq" % (e + 1)) for e in range(2)]
                for e in range(2):
                    synth.write_fastq_uniform(cpu_seq[e, :n_ref, :L], b.qual[e, :n_ref, :L], L, rfq[e], bgzf=False)
                usable = int(api.load_library().fq_host_cpus())
                refres = {"driver": "oracle/_ref/fq_ref_driver align (the reference's own stage functions, StatCollector and bwa_print_sam1)", "pairs": n_ref, "batch": 262144}
                for label, t_ in (("t4", 4), ("t_cpus", max(4, usable))):
                    run = subprocess.run([drv, "align", rpre, rfq[0], rfq[1], os.path.join(rdir, "out"), "--batch", "262144", "--t", str(t_), "--bench", "1", "--genome_size", str(len(ref.genome))],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                    tl = [l for l in run.stderr.decode(errors="replace").splitlines() if l.startswith("TIMING")]
                    if run.returncode == 0 and tl:
                        kv = dict(x.split("=") for x in tl[-1].split()[1:])
                        refres[label] = {"value": round(n_ref / (float(kv["reader_to_records_ms"]) * 1e-3), 1), "unit": "pairs/s", "threads": t_,
                                         "stage_a_ms": float(kv["stage_a_ms"]), "reader_to_records_ms": float(kv["reader_to_records_ms"])}
                    else:
                        refres[label] = {"error": run.stderr.decode(errors="replace")[-300:]}
                for f_ in rfq:
                    os.remove(f_)
                out["cpu_baseline"]["reference"] = refres
            except Exception as e:      # noqa: BLE001
                out["cpu_baseline"]["reference"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(out))
    ix.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
