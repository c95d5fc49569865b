#!/bin/bash
# PMC counters of a command via rocprofv3 (one pass per counter group; no tracing domains besides kernel-trace). usage: tools/gpu_pmc.sh <tag> "<counters;counters>" <cmd...>
tag=$1; shift; groups=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
IFS=';' read -ra G <<< "$groups"
for g in "${G[@]}"; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $g --output-format csv -d $d -o p -- "$@" ) > gpurun_out/${tag}_pmc_$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f gpurun_out/${tag}_pmc_$i.csv
  i=$((i+1))
done
python tools/pmc_summary.py query_kernel gpurun_out/${tag}_pmc_*.csv > gpurun_out/${tag}_pmc_query256.json 2>/dev/null
python tools/pmc_summary.py query_human8 gpurun_out/${tag}_pmc_*.csv > gpurun_out/${tag}_pmc_human8.json 2>/dev/null
cat gpurun_out/${tag}_pmc_query256.json gpurun_out/${tag}_pmc_human8.json
