#!/bin/bash
# kernel timeline of single-stream WGS calls: per call, the summed kernel time, the gaps between kernels and the largest of both
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_wtl
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/${TAG}_wtl -o p -- python3 $R/bench.py --ctxs 1 --steps 10 --warmup 3 --no-cpu-baseline --no-resident --no-ontarget --no-front-end > $O/${TAG}_wtl.json 2> $O/${TAG}_wtl.err
python3 - <<PY > $O/${TAG}_wgs_call_timeline.txt
import csv, glob, json, collections
f = glob.glob("$O/${TAG}_wtl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sk = [k for k in rows[0] if "Start" in k][0]; ek = [k for k in rows[0] if "End" in k][0]
ev = sorted((int(r[sk]), int(r[ek]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in rows)
mc = glob.glob("$O/${TAG}_wtl/**/*memory_copy_trace.csv", recursive=True)
cp = []
if mc:
    mr = list(csv.DictReader(open(mc[0])))
    if mr:
        s2 = [k for k in mr[0] if "Start" in k][0]; e2 = [k for k in mr[0] if "End" in k][0]
        cp = sorted((int(r[s2]), int(r[e2]), "memcpy " + r.get("Direction", "")) for r in mr)
d = json.loads(open("$O/${TAG}_wtl.json").read().strip().splitlines()[-1])
print("# single-stream WGS calls (bench.py --ctxs 1 --steps 10 --warmup 3 under rocprofv3 --kernel-trace --memory-copy-trace): value %.4g pairs/s, %.2f ms per step" % (d["value"], d["ms_per_step"]))
preps = [i for i, e in enumerate(ev) if e[2].startswith("k_prep")]
# the last 13 calls are the 3 solo calls + 10 timed; take the 10 timed ones
calls = preps[-13:-3]
allev = sorted(ev + cp)
for ci in range(len(calls) - 1):
    lo, hi = ev[calls[ci]][0], ev[calls[ci + 1]][0]
    sel = [e for e in allev if lo <= e[0] < hi]
    busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
    for s, e, n in sel[1:]:
        if s > cur_e: busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel: by[n][0] += e - s; by[n][1] += 1
    if ci == 0 or ci == len(calls) - 2:
        print("call %d: span %.2f ms, kernels+copies busy %.2f ms, %d launches, gaps %.2f ms in %d (largest %s)" % (ci, (hi - lo) / 1e6, busy / 1e6, len(sel), sum(g for g, _ in gaps) / 1e6, len(gaps), [(round(g / 1e3), n) for g, n in sorted(gaps, reverse=True)[:6]]))
        for n, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:14]: print("    %-28s %7.3f ms in %3d" % (n, t / 1e6, c))
    else:
        print("call %d: span %.2f ms, busy %.2f ms, %d launches, gaps %.2f ms" % (ci, (hi - lo) / 1e6, busy / 1e6, len(sel), sum(g for g, _ in gaps) / 1e6))
PY
rm -rf $O/${TAG}_wtl
cat $O/${TAG}_wgs_call_timeline.txt
