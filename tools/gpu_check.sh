#!/bin/bash
# Standard GPU-box run (through gpurun): the -m gpu suite, the default bench line, a single-stream kernel trace of the bench command.
# usage: tools/gpu_check.sh [tag]   -> gpurun_out/<tag>_{pytest.log,bench.json,kernel_stats_1stream.csv,...}
tag=${1:-run}
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/${tag}_pytest.log 2>&1
tail -30 gpurun_out/${tag}_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 3000 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err
d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline ) > gpurun_out/${tag}_prof.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats_1stream.csv && head -12 $f
tail -3 gpurun_out/${tag}_prof.log
