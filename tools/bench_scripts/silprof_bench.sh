#!/bin/bash
# silhouette kernels inside one bench batch (the fit's geometry), single stream, for the library in VT_LIB_PATH
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/bench.py --streams 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline ) > /tmp/silprofb.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith('sil_'): print(f"{r['Name'][:28]:28s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
