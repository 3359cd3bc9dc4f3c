#!/usr/bin/env python3
"""bench.py -- paired 150 bp reads aligned per second through the whole `align` hot path on MI355X.

A "step" is one pass of the hot path (encode+trim+filter -> gap search -> SA -> pairing -> mate SW -> refine/MD ->
records) over the resident input: one call on each of the --ctxs concurrent streams, a call being --pairs synthetic read
pairs (16 reference batches of 262,144, the reference's READ_BUFFER_SIZE, src/BwtMapper.h:36) against the 10k-marker reduced
reference of BASELINE.json configs[1] (1000 long + 9000 short flanks, l_pac 6,510,000).  Inputs are resident in HBM when the timed region starts (fq_batch_upload outside, fq_align_resident
inside).  Multi-GPU: one process per GPU, reads shard by batch, no data-path collective ("weak" scaling); the only
collectives are the barrier and the MAX over ranks of the elapsed time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

# One hardware queue per HIP stream: the library gives every alignment context its own stream, and the runtime's default of
# four hardware queues would make eight or sixteen streams share queues (a 15 ms persistent search kernel then blocks another
# stream's filter kernel).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
K_NAMES = ("prep", "width", "gap", "sa", "sw", "refine")
K_PREP_KERNEL, K_GAP_KERNEL = 6, 7     # single-kernel timings (kernel begin/end timestamps via hipExtLaunchKernelGGL events)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; one step = one call on every concurrent stream (--ctxs)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=16 * 262144,
                    help="pairs per step: one call carrying this many pairs = pairs/262144 reference batches (READ_BUFFER_SIZE)")
    ap.add_argument("--ctxs", type=int, default=16,
                    help="alignment contexts (independent FASTQ streams) driven concurrently, one host thread + HIP stream each")
    ap.add_argument("--markers", type=int, default=10000)
    ap.add_argument("--mix", choices=("wgs", "ontarget"), default="wgs",
                    help="wgs: on-target fraction l_pac/3.1e9 (SURVEY 8d); ontarget: every pair from a marker flank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="pairs per slice for the CPU baseline (0 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="independent streams (threads) of the CPU baseline")
    ap.add_argument("--workdir", default=os.environ.get("FQ_BENCH_DIR", "/tmp/fq_bench"))
    args = ap.parse_args()

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    from fastquick_amd import api, synth
    from fastquick_amd import dist as fqd
    rank, local_rank, world = fqd.init("nccl")      # RCCL; one process per GPU

    # ---- workload (seeded synthetic; built once per node by rank 0) ------------------------------------------
    os.makedirs(args.workdir, exist_ok=True)
    pre = os.path.join(args.workdir, "m%d.FASTQuick.fa" % args.markers)
    n_long = args.markers // 10
    ref = synth.make_reference(n_markers=args.markers, n_long=n_long, seed=12345)
    if rank == 0 and not os.path.exists(pre + ".rsa"):
        ref.write_fasta(pre)
        api.build_index(pre)
    fqd.barrier()
    on_frac = 1.0 if args.mix == "ontarget" else ref.l_pac / 3.1e9
    n_ctx = max(1, args.ctxs)
    # One step = one call on every resident stream (n_ctx calls of args.pairs pairs each): the timed region then holds
    # steps x n_ctx calls, so the pipeline of concurrent streams is in steady state for any K the caller picks.
    calls = args.steps * n_ctx

    def make_batch(n_pairs, seed):
        """Seeded synthetic batch.  Off-target pairs are i.i.d. random bases drawn on the GPU (fast), on-target
        pairs come from fastquick_amd.synth (fragments of the marker flanks with errors) at seeded random slots."""
        STRIDE = 160   # 16-byte aligned rows: the filter kernel then loads each read with 16-byte vector loads
        if on_frac >= 1.0:
            rb = synth.make_reads(ref, n_pairs, on_target=1.0, seed=seed)
            seq = np.zeros((2, n_pairs, STRIDE), dtype=np.uint8)
            qual = np.zeros((2, n_pairs, STRIDE), dtype=np.uint8)
            seq[:, :, :150] = rb.seq
            qual[:, :, :150] = rb.qual
            return synth.ReadBatch(seq, qual, rb.lens, None)
        rng = np.random.default_rng(seed)
        n_on = int(rng.binomial(n_pairs, on_frac))
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
        seq = np.zeros((2, n_pairs, STRIDE), dtype=np.uint8)
        step = 1 << 20
        for e in range(2):
            for a in range(0, n_pairs, step):
                b = min(n_pairs, a + step)
                codes = torch.randint(0, 4, (b - a, 150), device="cuda", generator=g, dtype=torch.uint8)
                seq[e, a:b, :150] = lut[codes.long()].cpu().numpy()
        on = synth.make_reads(ref, max(n_on, 1), on_target=1.0, seed=seed + 1)
        slots = np.sort(rng.choice(n_pairs, size=n_on, replace=False))
        seq[:, slots, :150] = on.seq[:, :n_on]
        qual = np.full((2, n_pairs, STRIDE), ord("I"), dtype=np.uint8)
        lens = np.full((2, n_pairs), 150, dtype=np.int32)
        return synth.ReadBatch(seq, qual, lens, None)

    # distinct host batches are 2.7 GB each at the default size: at most four per rank, shared round-robin by the contexts (every
    # context still owns its device copy, its stream state and its results)
    n_distinct = min(n_ctx, 4)
    distinct = [make_batch(args.pairs, 1000 + 17 * rank + b) for b in range(n_distinct)]
    batches = [distinct[b % n_distinct] for b in range(n_ctx)]
    ix = api.Index(pre, device=local_rank)
    ctxs = []
    for b in range(n_ctx):
        al = api.Aligner(ix, max_pairs=args.pairs)
        al.upload(batches[b].seq, batches[b].qual, batches[b].lens, None)   # inputs resident in HBM
        ctxs.append(al)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            fqd.barrier()
            torch.cuda.synchronize()

    import threading

    def run_steps(total):
        """`total` steps spread round-robin over the contexts; each context is driven by its own host thread (its own
        HIP stream inside the library), so the latency-bound stages of one stream overlap the others' work."""
        counts = [total // n_ctx + (1 if c < total % n_ctx else 0) for c in range(n_ctx)]
        recs = [0] * n_ctx
        errs = []

        def worker(c):
            try:
                for _ in range(counts[c]):
                    recs[c] += ctxs[c].align_resident().n_survivors
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=worker, args=(c,)) for c in range(n_ctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        return sum(recs)

    for al in ctxs:                 # set-up, like the upload: the first call of a context sizes its device buffers
        al.align_resident()
    run_steps(max(args.warmup, 0) * n_ctx)
    for al in ctxs:
        al.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    n_records = run_steps(calls)
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = fqd.max_over_ranks(elapsed)

    # ---- per-kernel device time (HIP events inside the library, on its own stream) + algorithmic work --------
    agg = None
    for al in ctxs:
        s = al.stats()
        if agg is None:
            agg = s
        else:
            for k, v in s.items():
                agg[k] = [a + b for a, b in zip(agg[k], v)] if isinstance(v, list) else agg[k] + v
    kms = agg["kernel_ms"]
    # Dominant kernel = the one that consumes the most of the device: summed device time x the share of the device's resident
    # lanes one launch can occupy.  (Several streams run concurrently; a search launch over the few thousand on-target reads of
    # a WGS-like batch occupies a few percent of the wavefront slots for as long as its longest search lasts, while the filter
    # kernel fills the whole device.)
    n_launch = [max(1, int(x)) for x in agg["kernel_launches"]]
    items = {"prep": 2.0 * args.pairs * calls / n_launch[0], "width": 2.0 * agg["reads_searched"] / n_launch[1],
             "gap": float(agg["reads_searched"]) / n_launch[2], "sa": float(agg["sa_rows"]) / n_launch[3],
             "sw": float(agg["sw_tasks"]) / n_launch[4], "refine": float(agg["refine_tasks"]) / n_launch[5]}
    capacity = {"prep": 524288.0, "width": 524288.0, "gap": 262144.0, "sa": 524288.0, "sw": 256.0, "refine": 16384.0}   # resident work items
    share = {k: min(1.0, items[k] / capacity[k]) for k in items}
    dom = max(range(len(K_NAMES)), key=lambda k: kms[k] * share[K_NAMES[k]])
    KSRC = {"prep": K_PREP_KERNEL, "gap": K_GAP_KERNEL}   # stage -> the kernel whose own timestamps price it
    seq_bytes = float(sum(int(b.lens.sum()) for b in batches)) / n_ctx * calls
    if K_NAMES[dom] == "prep":
        alg_bytes = 64.0 * agg["filter_probes"] + 96.0 * 2 * args.pairs * calls + 5.0 * 2 * args.pairs * calls
        model = "64 B x bitmap probes + 96 B of bases/read + 5 B/read out"
    elif K_NAMES[dom] == "gap":
        alg_bytes = 48.0 * agg["gap_occ_touches"]
        model = "48 B x Occ block touches (reference block definition, SURVEY 8d)"
    else:
        alg_bytes = 48.0 * agg["occ_block_touches"]
        model = "48 B x Occ block touches"
    # every kernel's algorithmic rate (the roofline object below repeats the dominant one)
    per_kernel = {}
    for kname, byts in (("prep", 64.0 * agg["filter_probes"] + 96.0 * 2 * args.pairs * calls + 5.0 * 2 * args.pairs * calls),
                        ("gap", 48.0 * agg["gap_occ_touches"])):
        ki = KSRC[kname]
        nl = max(1, int(agg["kernel_launches"][ki]))
        ms = kms[ki] / nl
        per_kernel["fq_" + kname] = {"avg_launch_ms": round(ms, 4), "alg_GBps": round((byts / nl) / (ms * 1e-3) / 1e9, 2) if ms > 0 else 0.0,
                                     "frac_of_hbm_peak": round((byts / nl) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else 0.0}
    dsrc = KSRC.get(K_NAMES[dom], dom)
    launches = max(1, int(agg["kernel_launches"][dsrc]))
    avg_ms = kms[dsrc] / launches
    achieved = (alg_bytes / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected separately,
    # tools/pmc_summarize.py), scaled by this run's units per launch; null when no measurement exists for this mix/kernel.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[args.mix]["fq_" + K_NAMES[dom]]
        unit_count = {"prep": 2.0 * args.pairs * calls, "gap": float(agg["reads_searched"])}.get(K_NAMES[dom])
        if unit_count:
            traffic = round(pmc["bytes_per_unit"] * unit_count / launches, 1)
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": "fq_" + K_NAMES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                "alg_bytes_per_launch": round(alg_bytes / launches, 1), "model": model,
                "aggregate_achieved": round(alg_bytes / elapsed / 1e9, 3),   # all concurrent launches together, over the wall time
                "dominance": "summed device ms x share of resident-lane capacity a launch occupies: " +
                             ", ".join("%s %.2f" % (K_NAMES[k], kms[k] * share[K_NAMES[k]] / calls) for k in range(len(K_NAMES)))}
    # the same kernel alone on the device (one stream, after the timed region): launch duration without other streams' kernels
    if n_ctx > 1:
        ctxs[0].reset_stats()
        for _ in range(3):
            ctxs[0].align_resident()
        s1 = ctxs[0].stats()
        nl1 = max(1, int(s1["kernel_launches"][dsrc]))
        ms1 = s1["kernel_ms"][dsrc] / nl1
        byts1 = (64.0 * s1["filter_probes"] + 101.0 * 2 * args.pairs * 3) if K_NAMES[dom] == "prep" else 48.0 * (s1["gap_occ_touches"] if K_NAMES[dom] == "gap" else s1["occ_block_touches"])
        if ms1 > 0:
            roofline["solo_avg_launch_ms"] = round(ms1, 4)
            roofline["solo_achieved"] = round(byts1 / nl1 / (ms1 * 1e-3) / 1e9, 3)
            roofline["solo_frac"] = round(byts1 / nl1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)

    total_pairs = args.pairs * calls * world
    value = total_pairs / elapsed
    out = {
        "metric": "paired_150bp_reads_aligned_per_sec", "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "10k-marker reduced ref (l_pac %d), %d x 2x150bp pairs/step = %d concurrent streams x one call of %d pairs (%d reference batches of 262144), %s mix (on-target %.4f)"
                   % (ref.l_pac, args.pairs * n_ctx, n_ctx, args.pairs, (args.pairs + 262143) // 262144, args.mix, on_frac),
                   "pairs_per_step": args.pairs * n_ctx, "pairs_per_call": args.pairs, "calls_per_step": n_ctx,
                   "markers": args.markers, "mix": args.mix, "concurrent_streams": n_ctx,
                   "sharding": "batches per rank, no data-path collective"},
        "roofline": roofline,
        "kernel_rooflines": per_kernel,
        "stage_ms_per_call": {K_NAMES[k]: round(kms[k] / calls, 4) for k in range(len(K_NAMES))},
        "host_ms_per_call": round(agg["host_ms_total"] / calls, 3),
        "survivor_pairs_per_call": round(n_records / calls, 1),
        "work_per_call": {k: round(agg[k] / calls, 1) for k in ("filter_probes", "occ_block_touches", "gap_occ_touches", "stack_pops",
                                                                     "stack_pushes", "sa_rows", "reads_searched", "sw_tasks", "refine_tasks", "tier_retries", "reads_over_4k_pops")},
        "max_pops_per_read": agg["max_pops_per_read"], "max_wave_trips": agg["max_wave_trips"],
        "gap_wave_trips_per_call": round(agg["wave_trips"] / calls, 1), "gap_lane_trips_per_call": round(agg["lane_trips"] / calls, 1),
    }

    # ---- what a multi-GPU run hands to rank 0 (outside the timed region): the SAM text of every rank's last call, in rank order, and
    #      the summed stream counters (the reference's FileStatCollector sums); RCCL all_gather / all_reduce, a few MB per rank
    last = ctxs[0].result
    if last.n_survivors <= 65536:
        parts = fqd.gather_bytes_to_rank0(ctxs[0].sam_text())
        tot = fqd.sum_counters({"pairs": int(last.n_pairs), "survivor_pairs": int(last.n_survivors), "both_filtered": int(last.n_both_filtered),
                                "both_unmapped": int(last.n_both_unmapped), "bases": int(last.n_bases)})
        if rank == 0:
            out["gather"] = {"ranks": len(parts), "sam_bytes": [len(p) for p in parts], "counters_sum": tot}

    # ---- CPU baseline: the oracle (a port) on a bounded sample of the same workload, rank 0, N=1 only -----------
    # Like the reference's thread pool over --fq_list lines: T independent streams (one oracle context each, shared read-only
    # index), every stream aligning consecutive slices of the same batch for about ten seconds.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_binding as ob
        b = batches[0]
        n_cpu = args.cpu_sample_pairs or (131072 if args.mix == "wgs" else 8192)
        n_cpu = min(n_cpu, args.pairs)
        T = max(1, min(args.cpu_threads, os.cpu_count() or 1, args.pairs // n_cpu))
        names = [b"r%09d" % i for i in range(n_cpu)]
        first = ob.OracleAligner(pre)
        oas = [first] + [ob.OracleAligner(pre, share=first) for _ in range(T - 1)]
        done = [0] * T
        budget = 10.0

        def cpu_worker(t):
            off = (t * n_cpu) % max(1, args.pairs - n_cpu + 1)
            t_end = time.perf_counter() + budget
            while time.perf_counter() < t_end:
                if off + n_cpu > args.pairs:
                    off = 0
                oas[t].align(names, b.seq[:, off:off + n_cpu], b.qual[:, off:off + n_cpu], b.lens[:, off:off + n_cpu], None, None, batch=n_cpu)
                done[t] += n_cpu
                off += n_cpu * T
        t1 = time.perf_counter()
        th = [threading.Thread(target=cpu_worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(sum(done) / dt, 1), "unit": "pairs/s", "cores": T, "kind": "port",
                               "sample": "%d pairs: %d independent streams x slices of %d pairs of the same %s-mix input through oracle/fq_oracle.c "
                                         "(one thread per stream, shared index), %.1f s" % (sum(done), T, n_cpu, args.mix, dt)}
        for oa in oas[1:]:
            oa.close()
        first.close()
    if rank == 0:
        print(json.dumps(out))
    for al in ctxs:
        al.close()
    ix.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
