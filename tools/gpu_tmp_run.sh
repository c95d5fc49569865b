export TMPDIR=/tmp; d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $GRAFT_REPO_ROOT/tools/bench_scripts/stage4_bench.py ) > gpurun_out/r04_stage4_prof.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/r04_stage4_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04_stage4_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
