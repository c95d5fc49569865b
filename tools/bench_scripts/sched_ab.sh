#!/bin/bash
# bench line: whole batches per stream (default) against SMPL stages and object stages on separate streams
for a in "--streams 2" "--streams 1 --schedule staged" "--streams 2 --schedule staged" "--streams 2" "--streams 2 --schedule staged"; do
  echo "== $a"
  python bench.py $a --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('frames/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'frac', round(r['frac'], 4), 'launches', r['launches'])"
done
