#!/usr/bin/env python3
"""stdin: hipcc -Rpass-analysis=kernel-resource-usage remarks of the product's sources; stdout: one table row per kernel (demangled name, VGPRs, AGPRs, SGPRs,
scratch bytes per lane, occupancy, static LDS) + a list of the kernels that spill.  Exit code 1 when a kernel named in NO_SCRATCH has scratch (the chip-filling
query kernels: their scratch traffic shows up as HBM writes -- 44 B x 256 lanes x 10 368 workgroups = 117 MB per launch in round 5)."""
import re
import subprocess
import sys

NO_SCRATCH = ("query_kernel<2, 2, true>", "query_kernel<2, 2, false>", "query_kernel<1, 3, true>", "query_kernel<1, 3, false>")

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (?:Function Name: (\S+)|\s+(\w[\w \[\]/]*?): (\d+))", line)
    if not m:
        continue
    if m.group(1):
        cur = {"name": m.group(1)}; rows.append(cur)
    elif cur is not None:
        cur[m.group(2).strip()] = int(m.group(3))
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines() if names else []
except Exception:      # noqa: BLE001
    dem = names
bad = []
print(f"{'kernel':<86} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>15} {'waves/SIMD':>11} {'LDS B':>7}")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*\)$", "", d).replace("void ", "")
    sc = r.get("ScratchSize [bytes/lane]", 0)
    print(f"{d[:86]:<86} {r.get('VGPRs', 0):>5} {r.get('AGPRs', 0):>5} {r.get('TotalSGPRs', 0):>5} {sc:>15} {r.get('Occupancy [waves/SIMD]', 0):>11} {r.get('LDS Size [bytes/block]', 0):>7}")
    if sc and any(n in d for n in NO_SCRATCH):
        bad.append((d, sc))
spill = [(re.sub(r"\(.*\)$", "", d), r.get("ScratchSize [bytes/lane]", 0)) for r, d in zip(rows, dem) if r.get("ScratchSize [bytes/lane]", 0)]
print("\nkernels with scratch:", spill if spill else "none")
if bad:
    print("FAIL: scratch in a no-scratch kernel:", bad)
    sys.exit(1)
