export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; d=$(mktemp -d /tmp/pmc.XXXX)
( cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d -o p -- python $R/tools/bench_scripts/qcmp.py run /tmp/q.npz 10 ) > /tmp/clk.log 2>&1
f=$(find $d -name '*counter_collection.csv' | head -1); head -1 $f; grep "query_kernel" $f | head -3 | cut -c1-150; grep "query_kernel" $f | awk -F, '{print $(NF-3), $(NF-2), $(NF-1), $NF}' | head -5
python - $f <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "query_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); acc[r["Kernel_Name"][:40]].append((float(r["Counter_Value"]), dt))
for k, v in acc.items():
    v = v[2:]
    print(k, "launches", len(v), "mean cycles", sum(a for a, _ in v) / len(v), "mean ns", sum(b for _, b in v) / len(v), "cycles/ns", sum(a for a, _ in v) / sum(b for _, b in v))
PY
