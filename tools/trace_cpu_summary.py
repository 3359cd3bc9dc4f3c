import collections, re, sys
d = collections.defaultdict(list); c = collections.defaultdict(list)
for l in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[fq\] (.+?)\s+([\d.]+) ms\s+cpu\s+([\d.]+) core-ms\s+this thread\s+([\d.]+) ms", l)
    if m:
        d[m.group(1)].append(float(m.group(2))); c[m.group(1)].append(float(m.group(4)))
tw = tc = 0
for k in d:
    v = d[k][len(d[k]) // 3:]; w = c[k][len(c[k]) // 3:]
    w = [x for x in w if x < 500]
    print("  %-44s n=%4d wall mean %7.2f   thread cpu mean %6.3f" % (k, len(v), sum(v) / len(v), sum(w) / max(1, len(w))))
    tc += sum(w) / max(1, len(w))
print("  sum of thread-cpu means %.2f ms" % tc)
