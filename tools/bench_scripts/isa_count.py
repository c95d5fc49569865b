"""dynamic VALU estimate of a query kernel from its ISA: instructions of blocks inside each loop x trip count + the rest"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
trips = {}; order = []
cur = None
cnt = collections.defaultdict(lambda: collections.Counter())
for l in lines:
    m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", l) or re.match(r"^; %bb\.\d+:\s*;\s*(.*)$", l)
    if l.startswith(".LBB") or l.startswith("; %bb."):
        c = l.split(";", 1)[1] if ";" in l[1:] else ""
        mh = re.search(r"Header=BB(\d+_\d+)", c)
        if "Loop Header" in l: cur = l.split(":")[0].replace(".LBB", ""); order.append(cur)
        elif mh: cur = mh.group(1)
        else: cur = None
        continue
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."): continue
    op = t[0]
    kind = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "scratch_", "buffer_")) else "other"
    cnt[cur][kind] += 1
tr = [19, 19]
dyn = collections.Counter(cnt[None])
print("straight-line:", dict(cnt[None]))
for h, n in zip(order, tr):
    print(f"loop {h} (x{n}):", dict(cnt[h]))
    for k, v in cnt[h].items(): dyn[k] += v * n
print("dynamic per wave:", dict(dyn))
