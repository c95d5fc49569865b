export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; d=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/bench_scripts/encbench.py 16 ) > gpurun_out/r03n_encoder.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/r03n_encoder_kernel_stats.csv; tail -2 gpurun_out/r03n_encoder.log
