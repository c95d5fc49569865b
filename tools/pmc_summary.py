"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of the kernels matching a pattern.  Dispatches that returned at
their first instruction (the launches queued behind a fit's stop step, vt_stream_set_skip_flag: counter value below 2 % of the kernel's maximum) are
counted apart and left out of the mean.
usage: pmc_summary.py <pattern> <csv> [<csv> ...]   -> JSON on stdout"""
import csv, sys, json, collections
pat = sys.argv[1]; acc = collections.defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in sorted(acc.items()):
    ex = [x for x in v if x >= 0.02 * max(v)] or v
    out[k] = {"mean": sum(ex) / len(ex), "dispatches": len(ex), "returned_at_once": len(v) - len(ex)}
print(json.dumps(out, indent=1))
