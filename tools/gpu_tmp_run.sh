R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base new; do
  lib=$R/vistracker_amd/libvistracker_hip.so; [ $v = base ] && lib=$R/vistracker_amd/libvistracker_hip_base.so
  VT_LIB_PATH=$lib python tools/bench_scripts/qcmp.py run /tmp/q_$v.npz 30 2>&1 | tail -2
done
done
python tools/bench_scripts/qcmp.py cmp /tmp/q_base.npz /tmp/q_new.npz | tee gpurun_out/r04e_ab.txt
