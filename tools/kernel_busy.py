#!/usr/bin/env python3
"""How busy the device was over a run: reads the kernel trace of `rocprofv3 --kernel-trace --output-format csv` (…_kernel_trace.csv) and prints the
span from the first kernel's start to the last one's end, the union of the kernels' intervals (time in which at least one kernel ran), how much of it had
two or more kernels side by side, the longest idle gaps with the kernels either side of them, and the kernels' summed times.

  tools/kernel_busy.py <dir or csv> [--from-ms A --to-ms B] [--mid F] [--gaps N] [--min-gap-ms G] [--no-copies]"""
import csv
import glob
import os
import sys


def load(path):
    if os.path.isdir(path):
        c = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
        if not c:
            raise SystemExit("no *kernel_trace.csv under " + path)
        path = c[0]
    rows = []
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("fqdev::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    # the copies of the same run, when it was traced with --memory-copy-trace as well (…_memory_copy_trace.csv beside the kernel trace): listed as "copy:<direction>"
    cp = path.replace("kernel_trace.csv", "memory_copy_trace.csv")
    if os.path.exists(cp) and "--no-copies" not in sys.argv:
        with open(cp, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?").replace("MEMORY_COPY_", ""), "copy"))
    rows.sort()
    return path, rows


def main():
    a = [x for x in sys.argv[1:] if x != "--no-copies"]
    if not a:
        raise SystemExit(__doc__)
    opt = {"--from-ms": None, "--to-ms": None, "--gaps": 12, "--min-gap-ms": 0.5, "--mid": None}
    i = 1
    while i < len(a):
        opt[a[i]] = float(a[i + 1]); i += 2
    path, rows = load(a[0])
    t0 = rows[0][0]
    lo = t0 + int(1e6 * opt["--from-ms"]) if opt["--from-ms"] is not None else t0
    hi = t0 + int(1e6 * opt["--to-ms"]) if opt["--to-ms"] is not None else max(r[1] for r in rows)
    if opt["--mid"] is not None:      # the middle of the run: --mid F keeps the fraction F of the span around its centre (the steady chunks, without start-up and tail)
        end = max(r[1] for r in rows); f = opt["--mid"]
        lo = t0 + int((end - t0) * (0.5 - f / 2)); hi = t0 + int((end - t0) * (0.5 + f / 2))
    rows = [r for r in rows if r[1] > lo and r[0] < hi]
    span = (hi - lo) / 1e6
    # sweep: busy union, time with >= 2 kernels
    ev = []
    for s, e, _, _ in rows:
        ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
    ev.sort()
    depth = 0; last = lo; busy = 0; multi = 0
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: multi += t - last
        depth += d; last = t
    print("%s" % path)
    print("kernels %d   span %.1f ms   at least one kernel running %.1f ms (%.3f)   two or more %.1f ms   summed kernel time %.1f ms"
          % (len(rows), span, busy / 1e6, busy / 1e6 / span, multi / 1e6, sum(min(e, hi) - max(s, lo) for s, e, _, _ in rows) / 1e6))
    # idle gaps
    gaps = []
    end = lo; prev = "(start)"
    for s, e, n, q in rows:
        if s > end:
            gaps.append((s - end, end, prev, n))
        if e > end:
            end = e; prev = n
    gaps.sort(reverse=True)
    tot = sum(g[0] for g in gaps) / 1e6
    mg = opt["--min-gap-ms"]
    print("idle %.1f ms in %d gaps; %.1f ms of it in gaps of %.1f ms or more; the longest:" % (tot, len(gaps), sum(g[0] for g in gaps if g[0] >= mg * 1e6) / 1e6, mg))
    for g in gaps[:int(opt["--gaps"])]:
        print("   %8.2f ms idle at %9.1f ms   after %-28s before %s" % (g[0] / 1e6, (g[1] - t0) / 1e6, g[2], g[3]))
    # histogram of gaps by the kernel that ended them / started them
    by = {}
    for g in gaps:
        by[g[2]] = by.get(g[2], 0) + g[0]
    print("idle time by the kernel BEFORE the gap:")
    for n, v in sorted(by.items(), key=lambda x: -x[1])[:14]:
        print("   %8.1f ms  %s" % (v / 1e6, n))
    sums = {}
    for s, e, n, _ in rows:
        v = sums.setdefault(n, [0, 0]); v[0] += e - s; v[1] += 1
    print("kernels by summed time:")
    for n, v in sorted(sums.items(), key=lambda x: -x[1][0])[:24]:
        print("   %9.1f ms  %6d x  %s" % (v[0] / 1e6, v[1], n))


if __name__ == "__main__":
    main()
