#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, MI355X_MICROARCH.md "HBM" and
"rocprofv3 PMC slots") into profiles/pmc_traffic.json: HBM-side bytes per unit of work for each kernel.

    python tools/pmc_summarize.py <mix> <units.json> gpurun_out/pmc_<mix>_FETCH_SIZE gpurun_out/pmc_<mix>_WRITE_SIZE

units.json = the bench JSON line of the same command (gives reads searched / reads prepped per launch).
Corrections: counters are in KB (x1024).  The guide's x2 read-side correction applies to wide coalesced streams; the
dominant accesses here are isolated 1-byte probes (64-byte requests) and 32-byte Occ blocks, for which FETCH_SIZE was
checked against the exact algorithmic byte count of the filter kernel (10.35 GB measured vs 10.45 GB algorithmic per
launch) -- so no factor is applied and the figure is labelled "uncorrected"."""
import collections
import csv
import glob
import json
import os
import sys

KMAP = {"k_prep": "fq_prep", "k_prep_packed": "fq_prep", "k_gap_nogap_lds": "fq_gap", "k_gap_nogap_stock": "fq_gap", "k_gap_persist_stock": "fq_gap", "k_gap_lds": "fq_gap", "k_gap": "fq_gap", "k_gap_persist": "fq_gap", "k_gap_persist_lds": "fq_gap", "k_gap_coop": "fq_gap_wave", "k_width": "fq_width", "k_width_strand": "fq_width", "k_sa": "fq_sa",
        "k_sw_wave": "fq_sw", "k_refine_lds": "fq_refine", "k_refine_wave": "fq_refine", "k_md_rec": "fq_md", "k_md_mask": "fq_md"}


def mean_kb(d):
    """KB per launch of each kernel GROUP (KMAP): the sum over the group's launches / their number -- the same "per launch"
    as bench.py's, which counts the gap-free first round and the full round of the search as launches of one stage."""
    f = (glob.glob(os.path.join(d, "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))[0]
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        if name in KMAP:
            agg[KMAP[name]].append(float(row["Counter_Value"]))
    return {k: (sum(v), len(v)) for k, v in agg.items()}


def main():
    mix, units_path, dfetch, dwrite = sys.argv[1:5]
    bench = json.loads([l for l in open(units_path) if l.startswith("{")][-1])
    per_step = bench.get("work_per_call") or bench["work_per_step"]      # per call of one stream (older lines: a step was one call)
    units = {"fq_prep": 2 * bench["config"].get("pairs_per_call", bench["config"]["pairs_per_step"]), "fq_gap": per_step["reads_searched"], "fq_gap_wave": max(1.0, per_step["tier_retries"]), "fq_width": per_step["reads_searched"],
             "fq_sa": max(1.0, per_step["sa_rows"]), "fq_sw": max(1.0, per_step["sw_tasks"]), "fq_refine": max(1.0, per_step["refine_tasks"]),
             "fq_md": max(1.0, per_step["reads_searched"])}      # (MD: one string per mapped read; searched reads are the closest count the bench line carries)
    fe, wr = mean_kb(dfetch), mean_kb(dwrite)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    n_calls = fe["fq_prep"][1]          # one filter launch per call: units below are per call, so is the byte total
    for name in sorted(set(KMAP.values())):
        if name in fe:
            per_call = (fe[name][0] + wr.get(name, (0.0, 0))[0]) * 1024.0 / n_calls
            nl = fe[name][1] / n_calls
            out.setdefault(mix, {})[name] = {"bytes_per_launch": per_call / nl, "launches_per_call": nl, "units_per_launch": units[name] / nl,
                                             "bytes_per_unit": per_call / units[name], "fetch_KB": fe[name][0] / fe[name][1],
                                             "write_KB": wr.get(name, (0.0, 1))[0] / max(1, wr.get(name, (0.0, 1))[1]),
                                             "correction": "uncorrected (see tools/pmc_summarize.py)"}
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out.get(mix, {}), indent=1))


if __name__ == "__main__":
    main()
