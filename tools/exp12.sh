# SQ / cache counters of the two search kernels at the headline call shape (one --pmc pass per counter group, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1200 python tools/exp_gap.py 4194304 - gap_generic_opts=1 > $O/exp12_gap.txt 2>&1
FQ_LIB_EXPERIMENT=$R/fastquick_amd/libfastquick_amd_wpe5.so timeout 900 python tools/exp_gap.py 4194304 - >> $O/exp12_gap.txt 2>&1
grep -v "^reads made" $O/exp12_gap.txt | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mode or option or golden" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\(SQ\|TCC\|TCP\|TA\|TD\|GRBM\)_[A-Z0-9_a-z]*" | sort -u > $O/exp12_avail.txt
Q="--no-cpu-baseline --no-resident --no-ontarget --no-front-end"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/exp12_g$i -o p -- python3 $R/bench.py --mix ontarget --pairs 4194304 --ctxs 1 --steps 1 --warmup 1 $Q > $O/exp12_g$i.json 2> $O/exp12_g$i.err
  find $O/exp12_g$i -name '*kernel_trace.csv' -delete
done
python3 - <<PY > $O/exp12_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$O/exp12_g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        if name.startswith("k_gap") or name in ("k_width", "k_refine_lds", "k_unpack", "k_md"):
            a = agg[(name, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-20s %-32s per launch %.6g  (launches %d)" % (k, c, v / n, n))
PY
cat $O/exp12_summary.txt | head -80
tail -3 $O/exp12_g1.err
