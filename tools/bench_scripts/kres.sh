#!/bin/bash
# registers / scratch of every query kernel of the current source (metadata of hipcc -S): kres.sh [extra -D flags]
cd "$(dirname "$0")/../../vistracker_amd/csrc"
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=off "$@" -S --cuda-device-only query.hip -o /tmp/kres_query.s 2>/dev/null
grep -E "^\s+\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count)|^\s+\.name:" /tmp/kres_query.s | paste - - - - | awk '{print $2, "scratch", $4, "vgprs", $6, "spilled", $8}' | grep query
