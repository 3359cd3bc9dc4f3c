#!/bin/bash
# per-kernel durations of one on-target call under rocprofv3 --kernel-trace --stats: tools/prof_gap.sh <tag> [pairs] [tuning]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o p -- python3 $R/tools/gap_paths.py ${2:-1048576} $3 > $R/gpurun_out/prof_$1.log 2>&1
python3 - "$R/gpurun_out/prof_$1" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:12]:
    print("%-40s calls %5s total %10.3f ms avg %10.3f ms" % (row["Name"].split("(")[0][-40:], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e6))
PY
