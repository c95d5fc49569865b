#!/bin/bash
# A/B of the fused query kernels: baseline library (vistracker_amd/libvistracker_hip_base.so, built from an earlier commit) against the current one,
# plus SQ_INSTS_VALU / SQ_INSTS_MFMA per launch of both.  usage: tools/gpu_ab.sh <tag>
tag=${1:-ab}; export TMPDIR=/tmp; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for v in base new; do
  lib=$R/vistracker_amd/libvistracker_hip.so; [ $v = base ] && lib=$R/vistracker_amd/libvistracker_hip_base.so
  VT_LIB_PATH=$lib python tools/bench_scripts/qcmp.py run /tmp/q_$v.npz 30 2>&1 | tail -3
done
python tools/bench_scripts/qcmp.py cmp /tmp/q_base.npz /tmp/q_new.npz | tee gpurun_out/${tag}_ab.txt
for v in base new; do
  lib=$R/vistracker_amd/libvistracker_hip.so; [ $v = base ] && lib=$R/vistracker_amd/libvistracker_hip_base.so
  d=$(mktemp -d /tmp/pmc.XXXX)
  ( cd /tmp && VT_LIB_PATH=$lib timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $d -o p -- python $R/tools/bench_scripts/qcmp.py run /tmp/q_pmc.npz 2 ) > gpurun_out/${tag}_pmc_$v.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "query_kernel<2" $f > gpurun_out/${tag}_insts_human_$v.json && python tools/pmc_summary.py "query_kernel<1" $f > gpurun_out/${tag}_insts_object_$v.json
  echo "== $v"; cat gpurun_out/${tag}_insts_human_$v.json
done
