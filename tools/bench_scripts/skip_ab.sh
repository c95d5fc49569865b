#!/bin/bash
# same-box A/B of the device-side skip of the launches queued behind a fit's stop step (FitContext.device_skip / VT_DEVICE_SKIP)
for s in 0 1 0 1; do
  echo "== VT_DEVICE_SKIP=$s"
  VT_DEVICE_SKIP=$s python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('frames/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'frac', round(r['frac'], 4), 'avg launch ms', r.get('avg_launch_ms'), 'launches', r['launches'])"
done
