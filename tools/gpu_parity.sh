#!/bin/bash
# GPU-box run of the parity evidence: the -m gpu suite (writes gpurun_out/fullsched_parity.json) and the bench-size full-schedule parity script.
# usage: tools/gpu_parity.sh [tag] [B]
tag=${1:-par}; B=${2:-8}
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -q -x --durations=12 ) > gpurun_out/${tag}_pytest.log 2>&1
tail -25 gpurun_out/${tag}_pytest.log
cat gpurun_out/fullsched_parity.json
( time timeout 1500 python tools/fullsize_parity.py $B gpurun_out/${tag}_fullsize_parity.json ) > gpurun_out/${tag}_fullsize.log 2>&1
tail -12 gpurun_out/${tag}_fullsize.log
