#!/bin/bash
# rocprofv3 evidence for everything that is not the fit loop (VERDICT r02 "missing" #5): kernel tables of the image encoders (configs[3]), the SIF-Net
# neural-only pass (encoders + surface-point generator, pipeline stage 4), the interpenetration term; counters of the convolution kernel.
# usage: tools/gpu_profiles.sh <tag>  -> gpurun_out/<tag>_*.csv|json
tag=${1:-prof}; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
trace() {   # name, command...
  n=$1; shift; d=$(mktemp -d /tmp/prof.XXXX)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- "$@" ) > gpurun_out/${tag}_${n}.log 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${n}_kernel_stats.csv
  tail -2 gpurun_out/${tag}_${n}.log
}
trace encoder python $R/tools/bench_scripts/encbench.py 16
trace stage4 python $R/tools/bench_scripts/stage4_bench.py
trace collide python $R/tools/bench_scripts/collidebench.py 96
trace inference python $R/tools/bench_scripts/sifnet_inference_bench.py
# counters of the convolution kernels (encoder pass): separate passes per group
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  ( cd /tmp && timeout 900 rocprofv3 --pmc $g --output-format csv -d $d -o p -- python $R/tools/bench_scripts/encbench.py 16 ) > gpurun_out/${tag}_pmc_conv_$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f /tmp/${tag}_pmc_conv_$i.csv
  i=$((i+1))
done
python tools/pmc_summary.py "conv3x3_kernel<2, true" /tmp/${tag}_pmc_conv_*.csv > gpurun_out/${tag}_pmc_conv3x3_128.json
python tools/pmc_summary.py "conv3x3_kernel<1, true" /tmp/${tag}_pmc_conv_*.csv > gpurun_out/${tag}_pmc_conv3x3_64.json
python tools/pmc_summary.py "conv3x3_kernel<2, false, 8" /tmp/${tag}_pmc_conv_*.csv > gpurun_out/${tag}_pmc_conv1x1_256in.json
cat gpurun_out/${tag}_pmc_conv3x3_128.json
