export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/bench_scripts/chamferbench.py ) > /tmp/chprof.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); grep -i "chamfer" $f | cut -c1-60,200-400
f2=$(find $d -name '*kernel_stats.csv' | head -1); python - $f2 <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'chamfer' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
