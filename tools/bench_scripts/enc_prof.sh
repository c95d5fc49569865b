# kernel mix of the encoder pass (1 warm-up + 8 timed passes of 16 frames: the warm-up's weight uploads show as __amd_rocclr_copyBuffer, 1/9 of the calls)
# usage: tools/bench_scripts/enc_prof.sh <tag> -> gpurun_out/<tag>_encoder_kernel_stats.csv
tag=${1:-enc}; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/bench_scripts/encbench.py 16 ) > gpurun_out/${tag}_encoder.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/${tag}_encoder_kernel_stats.csv; tail -2 gpurun_out/${tag}_encoder.log
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_encoder_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); print('kernel time per pass (9 passes) ms', tot/9e6)
for r in rows[:16]: print(f"{float(r['TotalDurationNs'])/9e6:8.3f} ms/pass {float(r['Percentage']):6.2f}% calls/pass {int(r['Calls'])/9:7.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:70]}")
PY
