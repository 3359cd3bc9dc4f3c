#!/bin/bash
# kernel timeline of any bench mix: device-busy share of the timed span and the kernels by the time they alone keep the device busy
# usage: tools/mix_timeline.sh <tag> <bench args...>      -> gpurun_out/<tag>_mix_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:?tag}; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_mtl
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_mtl -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-resident --no-ontarget --no-front-end > $O/${TAG}_mtl.json 2> $O/${TAG}_mtl.err
python3 - "$@" <<PY > $O/${TAG}_mix_timeline.txt
import csv, glob, json, collections, sys
f = glob.glob("$O/${TAG}_mtl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sk = [k for k in rows[0] if "Start" in k][0]; ek = [k for k in rows[0] if "End" in k][0]
ev = sorted((int(r[sk]), int(r[ek]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in rows)
d = json.loads(open("$O/${TAG}_mtl.json").read().strip().splitlines()[-1])
steps, ctxs = d["steps"], d["config"]["concurrent_streams"]
print("# bench.py %s under rocprofv3 --kernel-trace: value %.4g pairs/s, %.1f ms per step of %d calls" % (" ".join(sys.argv[1:]), d["value"], d["ms_per_step"], ctxs))
preps = [e[0] for e in ev if e[2].startswith("k_prep")]
lo, hi = preps[-(3 + steps * ctxs)], preps[-3]          # the timed steps (3 solo calls follow them)
sel = [e for e in ev if lo <= e[0] < hi]
# union of all kernels, and per kernel the time during which it is the ONLY kind of kernel running
pts = []
for s, e, n in sel: pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
active = collections.Counter(); busy = 0; alone = collections.Counter(); last = pts[0][0]; conc = collections.Counter()
for t, dlt, n in pts:
    if t > last:
        kinds = [k for k, v in active.items() if v > 0]
        tot = sum(active.values())
        if kinds: busy += t - last
        if len(kinds) == 1: alone[kinds[0]] += t - last
        conc[min(tot, 16)] += t - last
        last = t
    active[n] += dlt
span = hi - lo
print("span %.1f ms, device busy (union of kernels) %.1f ms = %.1f %%" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
print("kernels running at once (share of the span):", {k: round(100.0 * v / span, 1) for k, v in sorted(conc.items())})
by = collections.Counter()
for s, e, n in sel: by[n] += e - s
print("summed duration / alone on the device, ms per call:")
for n, v in by.most_common(14): print("  %-26s %8.2f  %8.2f" % (n, v / 1e6 / (steps * ctxs), alone[n] / 1e6 / (steps * ctxs)))
PY
rm -rf $O/${TAG}_mtl
cat $O/${TAG}_mix_timeline.txt
