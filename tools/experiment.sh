#!/bin/bash
# One parametrised script for the device experiments (it replaces the thirty one-off exp<N>.sh of round 3, whose results are in
# profiles/round3_search_kernel_analysis.txt).  Run on the GPU box:  gpurun -- 'bash tools/experiment.sh <what> [args]'
#   timeline <tag> <ctxs> [bench args]     kernel timeline of <ctxs> on-target streams of 4.2 M-pair calls: device-busy share, idle gaps,
#                                          summed time per kernel (rocprofv3 --kernel-trace)           -> gpurun_out/<tag>_timeline_<ctxs>.txt
#   knobs <tag> <pairs> "<k=v,..>" ...     the two rounds of the search stage for tuning variants on one resident on-target batch
#                                          (tools/exp_gap.py; "-" = defaults)                          -> gpurun_out/<tag>_knobs.txt
#   threads <tag> "<ctxs> <threads>" ...   on-target throughput per (streams, host threads per call) on this box's CPU quota
#                                                                                                      -> gpurun_out/<tag>_threads.txt
#   counters <tag> "<group>" ...           SQ / TCC / TCP counter groups over the kernels of one on-target call, one --pmc pass per group
#                                          (--kernel-trace only)                                       -> gpurun_out/<tag>_counters.txt
#   stats <tag> <name> [bench args]        rocprofv3 --kernel-trace --stats of one bench run           -> gpurun_out/<tag>_<name>_kernel_stats.csv + _bench.json
#   trace <tag> <pairs> [tuning]           host phases of one on-target call (tools/gap_paths.py trace=1) -> gpurun_out/<tag>_trace_<pairs>.txt
#   streams <tag> "<ctxs> [k=v,..]" ...    on-target throughput (4.2 M-pair calls) per (streams, tuning): value, per-launch times of the search
#                                          rounds and the width kernel in the timed region                -> gpurun_out/<tag>_streams.txt
#   libs <tag> <pairs> <lib.so> ... -- "<k=v,..>" ...   the `knobs` experiment for several builds of the library (compile-time variants:
#                                          make -C fastquick_amd/csrc OUT=../libfq_x.so BUILD=build_x EXTRA=-D...)  -> gpurun_out/<tag>_libs.txt
#   reader <tag>                           the FASTQ reader and the command line at a steady state, zlib's inflate against the front end's own
#                                          decoder (FASTQUICK_ZLIB_INFLATE=1 / 0) on the same box          -> gpurun_out/<tag>_reader.txt
set -u
WHAT=${1:?what}; TAG=${2:?tag}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end --no-host-budget"
cd /tmp && export TMPDIR=/tmp
case $WHAT in
timeline)
  N=${1:?ctxs}; shift
  rm -rf $O/${TAG}_tl$N
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_tl$N -o p -- python3 $R/bench.py --mix ontarget --pairs 4194304 --ctxs $N --steps 6 --warmup 2 $Q "$@" > $O/${TAG}_tl$N.json 2> $O/${TAG}_tl$N.err
  python3 - <<PY > $O/${TAG}_timeline_$N.txt
import csv, glob, json, collections
f = glob.glob("$O/${TAG}_tl$N/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sk = [k for k in rows[0] if "Start" in k][0]; ek = [k for k in rows[0] if "End" in k][0]
qk = [k for k in rows[0] if "Queue" in k or "Stream" in k]
ev = sorted(((int(r[sk]), int(r[ek]), r["Kernel_Name"].split("(")[0].split("::")[-1], r.get(qk[0], "") if qk else "") for r in rows))
d = json.loads(open("$O/${TAG}_tl$N.json").read().strip().splitlines()[-1])
print("# kernel timeline of $N on-target streams, 4,194,304 pairs per call (rocprofv3 --kernel-trace around bench.py --mix ontarget --ctxs $N --steps 6 --warmup 2)")
print("ctxs $N value %.4g pairs/s ms_per_step %.1f host_ms_per_call %s" % (d["value"], d["ms_per_step"], d.get("host_ms_per_call")))
# the timed region: a call starts with its filter kernel; the run ends with 3 solo calls (bench.py's "same kernels alone" leg) behind the
# 6 timed steps of $N calls each
preps = [e[0] for e in ev if e[2].startswith("k_prep")]
lo, hi = preps[-(3 + 6 * $N)], preps[-3]
sel = [e for e in ev if lo <= e[0] < hi]
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
for k, v in sorted(by.items(), key=lambda x: -x[1])[:16]: print("  %-28s %8.1f ms summed (%.1f %% of span)" % (k, v / 1e6, 100.0 * v / span))
gaps.sort(reverse=True)
print("largest idle gaps (ms, followed by):", [(round(g / 1e6, 2), nme) for g, _, nme in gaps[:12]])
print("idle gaps > 1 ms: %d, summing %.1f ms" % (sum(1 for g in gaps if g[0] > 1e6), sum(g[0] for g in gaps if g[0] > 1e6) / 1e6))
big = [(s, e, nme, q) for s, e, nme, q in sel if e - s > 3e6]
print("big kernels (start ms, dur ms, name, queue):")
for s, e, nme, q in big[:60]: print("   %9.1f %7.1f  %-24s %s" % ((s - sel[0][0]) / 1e6, (e - s) / 1e6, nme, q))
PY
  rm -rf $O/${TAG}_tl$N
  head -40 $O/${TAG}_timeline_$N.txt ;;
knobs)
  P=${1:?pairs}; shift
  cd $R && timeout 1500 python tools/exp_gap.py $P "$@" > $O/${TAG}_knobs.txt 2>&1
  grep -v "^reads made" $O/${TAG}_knobs.txt | cut -c1-260 ;;
threads)
  cd $R
  { echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "nproc: $(nproc)"; } > $O/${TAG}_threads.txt
  for cfg in "$@"; do
    set -- $cfg
    timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs $1 --steps 3 --warmup 1 $Q --tune host_threads=$2 > $O/${TAG}_thr.json 2>> $O/${TAG}_threads.err
    python3 -c "
import json
d = json.loads(open('$O/${TAG}_thr.json').read().strip().splitlines()[-1])
print('ctxs $1 host_threads $2: value %.4g ms/step %.1f host_ms_per_call %.1f' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call', -1)))" >> $O/${TAG}_threads.txt
  done
  cat $O/${TAG}_threads.txt ;;
counters)
  i=0
  for grp in "$@"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_g$i -o p -- python3 $R/bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 1 --warmup 1 $Q > $O/${TAG}_g$i.json 2> $O/${TAG}_g$i.err
    find $O/${TAG}_g$i -name '*kernel_trace.csv' -delete
  done
  python3 - <<PY > $O/${TAG}_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$O/${TAG}_g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        a = agg[(name, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if v / n >= 1e6: print("%-22s %-32s per launch %.6g  (launches %d)" % (k, c, v / n, n))
PY
  rm -rf $O/${TAG}_g[0-9]*
  head -80 $O/${TAG}_counters.txt ;;
stats)
  NAME=${1:?name}; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${NAME}_prof -o p -- python3 $R/bench.py "$@" > $O/${TAG}_${NAME}_bench.json 2> $O/${TAG}_${NAME}_bench.err
  cp $(find $O/${TAG}_${NAME}_prof -name '*kernel_stats.csv' | head -1) $O/${TAG}_${NAME}_kernel_stats.csv 2>/dev/null
  rm -rf $O/${TAG}_${NAME}_prof
  head -25 $O/${TAG}_${NAME}_kernel_stats.csv | cut -c1-160 ;;
trace)
  P=${1:?pairs}; shift
  cd $R && timeout 600 python tools/gap_paths.py $P trace=1${1:+,$1} 2>&1 | tail -34 | grep -v arena > $O/${TAG}_trace_$P.txt
  cat $O/${TAG}_trace_$P.txt ;;
streams)
  cd $R
  : > $O/${TAG}_streams.txt
  for cfg in "$@"; do
    set -- $cfg
    T=""; [ -n "${2:-}" ] && T="--tune $2"
    timeout 600 python bench.py --mix ontarget --pairs 4194304 --ctxs $1 --steps 4 --warmup 1 $Q $T > $O/${TAG}_str.json 2>> $O/${TAG}_streams.err
    python3 -c "
import json
d = json.loads(open('$O/${TAG}_str.json').read().strip().splitlines()[-1])
k = d['kernel_rooflines']
print('ctxs $1 tuning ${2:--}: value %.4g pairs/s ms/step %.1f host_ms/call %s | first round %.2f second %.2f width %.2f ms per launch | fq_gap %.3f of 8 TB/s' % (d['value'], d['ms_per_step'], d.get('host_ms_per_call'), k['fq_gap_nogap']['avg_launch_ms'], k['fq_gap_full']['avg_launch_ms'], k['fq_width']['avg_launch_ms'], k['fq_gap']['frac_of_hbm_peak']))" >> $O/${TAG}_streams.txt
  done
  cat $O/${TAG}_streams.txt ;;
libs)
  P=${1:?pairs}; shift
  LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
  [ "${1:-}" = "--" ] && shift
  cd $R
  : > $O/${TAG}_libs.txt
  for L in "${LIBS[@]}"; do
    echo "## $L" >> $O/${TAG}_libs.txt
    FQ_LIB_EXPERIMENT=$R/$L timeout 900 python tools/exp_gap.py $P "${@:--}" 2>&1 | grep -v "^reads made" | cut -c1-330 >> $O/${TAG}_libs.txt
  done
  cat $O/${TAG}_libs.txt ;;
reader)
  cd $R
  : > $O/${TAG}_reader.txt
  for z in 1 0; do
    FASTQUICK_ZLIB_INFLATE=$z timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-resident --no-ontarget > $O/${TAG}_rd_$z.json 2> $O/${TAG}_rd_$z.err
    python3 -c "
import json
d = json.loads(open('$O/${TAG}_rd_$z.json').read().strip().splitlines()[-1])
fe = d['front_end']
print('zlib_only=$z tokenise_inflate', fe.get('tokenise_inflate_pairs_per_s'), fe.get('tokenise_inflate'))
for k, v in (fe.get('cli_e2e_steady') or {}).items():
    if isinstance(v, dict): print('  ', k, v.get('pairs_per_s'), v.get('wall_s'), v.get('notices'))" >> $O/${TAG}_reader.txt
  done
  cat $O/${TAG}_reader.txt ;;
*) echo "unknown experiment $WHAT"; exit 2 ;;
esac
