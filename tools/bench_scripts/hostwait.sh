#!/bin/bash
# how much slack the launching host threads have: bench line with the fraction of their time spent blocked on the device-side stop flag
for a in "--streams 2" "--streams 1" "--streams 2"; do
  echo "== $a"
  python bench.py $a --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('frames/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'host_wait_frac', round(d['config']['host_wait_frac'], 3), 'solo', r.get('solo_launch_ms'))"
done
nproc; python -c "import os; print(os.sched_getaffinity(0).__len__())"; uptime
