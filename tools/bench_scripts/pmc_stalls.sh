#!/bin/bash
# stall / busy counters of the two fused query kernels at the bench shape (one stream, tools/bench_scripts/qcmp.py), separate --pmc passes
# usage: tools/bench_scripts/pmc_stalls.sh <tag> [human-kernel name pattern]  -> gpurun_out/<tag>_pmc_stalls_query_{human,object}.json
# (VT_QUERY_HUMAN_KERNEL=128 tools/bench_scripts/pmc_stalls.sh r04_pc query_human_pc_kernel : the producer / consumer variant)
tag=${1:-stalls}; hk=${2:-query_kernel<2, 2}; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; i=0
for g in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $g --output-format csv -d $d -o p -- python $R/tools/bench_scripts/qcmp.py run /tmp/q_pmc.npz 3 ) > gpurun_out/${tag}_pmc_stalls_$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f /tmp/${tag}_stalls_$i.csv
  i=$((i+1))
done
python tools/pmc_summary.py "$hk" /tmp/${tag}_stalls_*.csv > gpurun_out/${tag}_pmc_stalls_query_human.json
python tools/pmc_summary.py "query_kernel<1, 3" /tmp/${tag}_stalls_*.csv > gpurun_out/${tag}_pmc_stalls_query_object.json
cat gpurun_out/${tag}_pmc_stalls_query_human.json
