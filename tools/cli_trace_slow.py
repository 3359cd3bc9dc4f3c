#!/usr/bin/env python3
"""(experiment) the on-target command-line run several times with both traces on: per call, the stages that took more than 60 ms"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, FASTQUICK_CTX_TRACE="1", FASTQUICK_TRACE="1")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_ontarget.py"), "1048576", "16", "150"] + sys.argv[2:], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    for m in ("sam_out", "bam_and_qc"):
        if m not in d:
            continue
        print("run", rep, m, d[m]["wall_s"], d[m]["pairs_per_s"])
        print("   ", [l for l in d[m]["notices"] if "alignment calls" in l or "consumers (ms)" in l])
        calls = [float(l.split()[2]) for l in d[m].get("trace", []) if "call done" in l]
        print("    call done at", calls)
        seg = []
        for l in d[m].get("ctx_trace", []):
            if "arena:" in l:
                if seg:
                    print("    call:", " | ".join(seg))
                seg = []
                continue
            mm = re.search(r"\[fq\]\s+(.*?)\s+([0-9.]+) ms   cpu", l)
            if mm and float(mm.group(2)) > 60:
                seg.append("%s %.0f" % (mm.group(1)[:28], float(mm.group(2))))
        if seg:
            print("    call:", " | ".join(seg))
