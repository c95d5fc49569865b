"""Dynamic opcode mix of one kernel from hipcc's -S output: every basic block weighted by the trip count of the loop it sits in.
usage: isa_mix.py <file.s> <mangled kernel name> [loop trips, default 11]   (prints VALU opcode histogram, per-block totals)"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n"); name = sys.argv[2]; trips = int(sys.argv[3]) if len(sys.argv) > 3 else 11
start = next(i for i, l in enumerate(src) if l.startswith(name + ":"))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks = []; cur = ["entry", 1, collections.Counter()]
for l in src[start + 1:end]:
    if l.startswith(".LBB") or l.startswith("; %bb."):
        blocks.append(cur)
        w = trips if ("in Loop" in l or "Loop Header" in l) else 1
        cur = [l.split(":")[0].strip("; "), w, collections.Counter()]
        continue
    t = l.strip().split()
    if not t or t[0][0] in ";." or t[0].endswith(":"): continue
    cur[2][t[0]] += 1
blocks.append(cur)
tot = collections.Counter(); kinds = collections.Counter()
def kind(op):
    return ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "scratch_", "buffer_")) else "other")
print("block weight valu mfma lds vmem salu")
for n, w, c in blocks:
    k = collections.Counter()
    for op, v in c.items(): k[kind(op)] += v; tot[op] += v * w; kinds[kind(op)] += v * w
    if sum(k.values()) > 40: print(f"{n:12s} x{w:<3d} {k['valu']:5d} {k['mfma']:4d} {k['lds']:4d} {k['vmem']:4d} {k['salu']:5d}")
print("dynamic per wave:", dict(kinds))
for op, v in sorted(tot.items(), key=lambda x: -x[1])[:45]:
    if kind(op) in ("valu",): print(f"  {op:28s} {v}")
