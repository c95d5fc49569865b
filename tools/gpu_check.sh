#!/bin/bash
# Standard GPU-box run (through gpurun): the -m gpu suite, the bench line of the DRIVER'S EXACT COMMAND (python bench.py --gpus 1 --steps 20 --warmup 5: the command
# BENCH_rNN.json records; VERDICT r04: it had never been run by the builder and its demo_pipeline leg OOM'd), a single-stream kernel trace of the bench command.
# usage: tools/gpu_check.sh [tag]   -> gpurun_out/<tag>_{pytest.log,bench.json,kernel_stats_1stream.csv,...}
tag=${1:-run}
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/${tag}_pytest.log 2>&1; python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/${tag}_pytest.log 2>&1
tail -30 gpurun_out/${tag}_pytest.log
# the measured-negative kernel variants (512 thin waves, producer / consumer) as INDEPENDENT cross-checks of the default kernel at bench size: the
# experiments library is not shipped to the box (gpurunignore), so it is built here; the tests pick the variants up when the library they load has them
( make -s -C vistracker_amd/csrc experiments > gpurun_out/${tag}_exp_build.log 2>&1 && VT_LIB_PATH=$GRAFT_REPO_ROOT/tools/bench_scripts/_exp/libvistracker_hip_exp.so timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k fused_query 2>&1 | grep -v amdgpu | tail -3 ) | tee gpurun_out/${tag}_exp_crosscheck.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${tag}_bench_driver_cmd.json 2> gpurun_out/${tag}_bench.err
tail -c 3000 gpurun_out/${tag}_bench_driver_cmd.json; tail -5 gpurun_out/${tag}_bench.err
grep -h '^{' gpurun_out/${tag}_bench_driver_cmd.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('HEADLINE', round(d['value'], 2), 'frames/s', round(d['ms_per_step'], 1), 'ms/batch; solo', round(r['solo_launch_ms'], 4), 'ms frac', round(r['frac_single_stream'], 4), 'in situ frac', round(r['frac'], 4),
          '; legs:', {k: ('ERROR ' + str(d[k]['error'])[:80] if isinstance(d.get(k), dict) and 'error' in d[k] else 'ok') for k in ('full_schedule', 'smplt_prefit', 'strict_fp32', 'sifnet_inference', 'demo_pipeline', 'cpu_baseline')})
"
d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/${tag}_prof.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats_1stream.csv && head -12 $f
# the launches queued behind a fit's stop step return at their first instruction (vt_stream_set_skip_flag) but are launches of the same kernel: the table's
# average includes them; the per-dispatch trace gives the average of the launches that did the work (what bench.py's avg_launch_ms measures)
t=$(find $d -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python tools/trace_summary.py $t "query_kernel<2, 2" "query_kernel<1, 3" > gpurun_out/${tag}_query_launches_1stream.json && cat gpurun_out/${tag}_query_launches_1stream.json
tail -3 gpurun_out/${tag}_prof.log
# the same trace of the DEFAULT command's timed configuration (two batches in flight): the average the bench line reports as roofline.avg_launch_ms
d=$(mktemp -d /tmp/prof2.XXXX)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/${tag}_prof2.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats.csv && head -4 $f
t=$(find $d -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python tools/trace_summary.py $t "query_kernel<2, 2" "query_kernel<1, 3" > gpurun_out/${tag}_query_launches.json && cat gpurun_out/${tag}_query_launches.json
grep -h '^{' gpurun_out/${tag}_prof2.log | python -c "import sys, json; [print('bench under the profiler: avg_launch_ms', json.loads(l)['roofline']['avg_launch_ms']) for l in sys.stdin]"
# HBM-side traffic of the query kernels: separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass), one stream, one batch
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline ) > gpurun_out/${tag}_pmc_$c.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_pmc_$c.csv
done
python tools/pmc_summary.py "query_kernel<2, 2" gpurun_out/${tag}_pmc_FETCH_SIZE.csv gpurun_out/${tag}_pmc_WRITE_SIZE.csv > gpurun_out/${tag}_pmc_query_human.json
python tools/pmc_summary.py "query_kernel<1, 3" gpurun_out/${tag}_pmc_FETCH_SIZE.csv gpurun_out/${tag}_pmc_WRITE_SIZE.csv > gpurun_out/${tag}_pmc_query_object.json
cat gpurun_out/${tag}_pmc_query_human.json
# (the raw counter files stay in gpurun_out/ for the session)
