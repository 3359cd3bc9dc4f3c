# stock-option kernels + width-one step + cold lane state in LDS: timing of the two rounds (default build, generic-option kernels, 5 waves per SIMD
# in the first round), parity tests of the search modes, VALU instruction counts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1200 python tools/exp_gap.py 4194304 - gap_generic_opts=1 > $O/exp13_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_wpe5.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp13_gap.txt 2>&1
timeout 600 python tools/exp_gap.py 1048576 - >> $O/exp13_gap.txt 2>&1
grep -v "^reads made" $O/exp13_gap.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/exp13_g$i -o p -- python3 $R/bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 1 --warmup 1 $Q > $O/exp13_g$i.json 2> $O/exp13_g$i.err
  find $O/exp13_g$i -name '*kernel_trace.csv' -delete
done
python3 - <<PY > $O/exp13_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$O/exp13_g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        if name.startswith("k_gap") or name in ("k_width",):
            a = agg[(name, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-20s %-32s per launch %.6g  (launches %d)" % (k, c, v / n, n))
PY
cat $O/exp13_summary.txt
