#!/bin/bash
# batches in flight: bench line with 1..4 streams on one box
for s in 2 3 4 1 2; do
  echo "== --streams $s"
  python bench.py --streams $s --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('frames/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'frac', round(r['frac'], 4), 'launches', r['launches'])"
done
