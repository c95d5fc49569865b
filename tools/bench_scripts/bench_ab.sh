#!/bin/bash
# Same-box A/B of the bench line under different settings: every argument is one variant, "ENV=... ENV=... -- bench.py flags", run in the order given
# (repeat a variant to see the box's run-to-run spread).  Replaces the one-off skip / streams / schedule / host-wait shells of round 3.
#   tools/bench_scripts/bench_ab.sh "VT_DEVICE_SKIP=0 --" "VT_DEVICE_SKIP=1 --" "VT_DEVICE_SKIP=0 --"
#   tools/bench_scripts/bench_ab.sh "-- --streams 1" "-- --streams 2" "-- --streams 2 --schedule staged"
#   tools/bench_scripts/bench_ab.sh "VT_QUERY_HUMAN_KERNEL=256 --" "VT_QUERY_HUMAN_KERNEL=128 --"
#   BENCH_PREFIX="taskset -c 0" tools/bench_scripts/bench_ab.sh ...      (a starved host: every thread of the process on one core)
for v in "$@"; do
  envs=${v%%--*}; flags=${v#*--}
  echo "== $v"
  env $envs $BENCH_PREFIX python bench.py $flags --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; c = d['config']
print('frames/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'frac', round(r['frac'], 4), 'avg launch ms', round(r['avg_launch_ms'], 3), 'launches', r['launches'],
      'host_wait_frac', round(c['host_wait_frac'], 3), 'frame-steps/s', round(c['frame_steps_per_s']))"
done
