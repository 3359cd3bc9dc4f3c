#!/usr/bin/env python3
"""Mean / median / max of the per-stage host timings a traced run prints on stderr (`--tune trace=1`)."""
import collections, re, sys
for path in sys.argv[1:]:
    d = collections.defaultdict(list)
    for l in open(path, errors="replace"):
        m = re.match(r"\[fq\] (.+?)\s+([\d.]+) ms", l)
        if m:
            d[m.group(1)].append(float(m.group(2)))
    tot = 0.0
    print(path)
    for k, v in d.items():
        v2 = v[len(v) // 3:]
        print("  %-24s n=%4d mean %7.2f  med %7.2f max %7.2f" % (k, len(v2), sum(v2) / len(v2), sorted(v2)[len(v2) // 2], max(v2)))
        tot += sum(v2) / len(v2)
    print("  sum of means %.2f ms" % tot)
