"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of the kernels matching a pattern.
usage: pmc_summary.py <pattern> <csv> [<csv> ...]   -> JSON on stdout"""
import csv, sys, json, collections
pat = sys.argv[1]; acc = collections.defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {"mean": sum(v) / len(v), "dispatches": len(v)} for k, v in sorted(acc.items())}, indent=1))
