#!/usr/bin/env python3
"""Print per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv (scratch helper)."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("k_gap") or k.startswith("k_width") or (len(sys.argv) > 2 and False):
            print(d, k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()}, "n=%d" % len(list(v.values())[0]))
