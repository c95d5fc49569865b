"""Per-kernel launch statistics from a rocprofv3 kernel_trace.csv with the early-returned launches (device-side skip behind a fit's stop step,
< 50 us) counted apart.  usage: trace_summary.py <kernel_trace.csv> <name pattern> [<pattern> ...]  -> JSON on stdout"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for pat in sys.argv[2:]:
    d = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3 for r in rows if pat in r["Kernel_Name"]]
    ex = [x for x in d if x >= 50.0]
    out[pat] = {"launches": len(d), "returned_at_once": len(d) - len(ex), "mean_us_all": sum(d) / max(len(d), 1), "mean_us_executed": sum(ex) / max(len(ex), 1)}
print(json.dumps(out, indent=1))
