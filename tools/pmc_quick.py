#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 --pmc output directories (counter_collection.csv).
usage: pmc_quick.py [kernel-name-prefix] dir...   (default prefix: k_gap, the search kernels)"""
import collections, csv, glob, sys
args = sys.argv[1:]
prefix = "k_gap"
if args and not glob.glob(args[0] + "/*") and not args[0].startswith("/"):
    prefix, args = args[0], args[1:]
for d in args:
    fs = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith(prefix):
            n = len(list(v.values())[0])
            print(d.split("/")[-1], k, "launches=%d" % n, {c: "%.4g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
