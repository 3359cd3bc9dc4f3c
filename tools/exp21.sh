# kernel timeline of two on-target streams (4.2 M-pair calls): how much of the wall time the device is busy, and with what
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
for n in 2 3; do
rm -rf $O/exp21_t$n
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/exp21_t$n -o p -- python3 $R/bench.py --mix ontarget --pairs 4194304 --ctxs $n --steps 3 --warmup 1 $Q > $O/exp21_t$n.json 2> $O/exp21_t$n.err
python3 - <<PY > $O/exp21_timeline_$n.txt
import csv, glob, json, collections
f = glob.glob("$O/exp21_t$n/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sk = [k for k in rows[0] if "Start" in k][0]; ek = [k for k in rows[0] if "End" in k][0]
qk = [k for k in rows[0] if "Queue" in k or "Stream" in k]
ev = sorted(((int(r[sk]), int(r[ek]), r["Kernel_Name"].split("(")[0].split("::")[-1], r.get(qk[0], "") if qk else "") for r in rows))
d = json.loads(open("$O/exp21_t$n.json").read().strip().splitlines()[-1])
print("ctxs $n value %.4g ms_per_step %.1f" % (d["value"], d["ms_per_step"]))
# the last 60 % of the trace ~ the timed region
t0 = ev[0][0]; t1 = max(e[1] for e in ev)
lo = t0 + int(0.45 * (t1 - t0))
sel = [e for e in ev if e[0] >= lo]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
gaps = []
for s, e, nme, q in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e, nme)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - sel[0][0]
print("span %.1f ms, device busy (union of kernels) %.1f ms = %.1f %%" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
by = collections.defaultdict(float)
for s, e, nme, q in sel: by[nme] += e - s
for k, v in sorted(by.items(), key=lambda x: -x[1])[:14]: print("  %-28s %8.1f ms summed (%.1f %% of span)" % (k, v / 1e6, 100.0 * v / span))
gaps.sort(reverse=True)
print("largest idle gaps (ms, followed by):", [(round(g / 1e6, 2), nme) for g, _, nme in gaps[:12]])
print("idle gaps > 1 ms: %d, summing %.1f ms" % (sum(1 for g in gaps if g[0] > 1e6), sum(g[0] for g in gaps if g[0] > 1e6) / 1e6))
# sequence of big kernels with start offsets (ms) for a look at the interleaving
big = [(s, e, nme, q) for s, e, nme, q in sel if e - s > 3e6]
print("big kernels (start ms, dur ms, name, queue):")
for s, e, nme, q in big[:60]: print("   %9.1f %7.1f  %-24s %s" % ((s - sel[0][0]) / 1e6, (e - s) / 1e6, nme, q))
PY
find $O/exp21_t$n -name '*.csv' -delete
done
cat $O/exp21_timeline_2.txt | head -60
