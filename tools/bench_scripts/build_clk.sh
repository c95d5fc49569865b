#!/bin/bash
# instrumented library (per-phase shader clocks of the query kernels): tools/bench_scripts/_exp/libvistracker_hip_clk<suffix>.so (git-ignored; VT_LIB_PATH selects it) ; extra -D flags as arguments
# usage: build_clk.sh [suffix [extra hipcc flags ...]]
set -e
cd "$(dirname "$0")/../../vistracker_amd/csrc"
suf=$1; shift || true
mkdir -p ../../tools/bench_scripts/_exp
F="-O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off"
/opt/rocm/bin/hipcc $F -DPHASE_CLK "$@" -c query.hip -o /tmp/query_clk$suf.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bench_scripts/_exp/libvistracker_hip_clk$suf.so /tmp/query_clk$suf.o misc.o smplh.o query_f32.o chamfer.o collide.o conv.o sil.o gen.o
