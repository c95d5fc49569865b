"""Frame sharding of one sequence over the GPUs of a node and the parameter gather at a pipeline barrier.

The reference parallelises by launching processes with disjoint ``--start/--end`` (README.md:55;
recon/recon_fit_base.py:411-419): frames are cut into batches of ``bs`` consecutive frames and temporal terms couple
frames only inside a batch.  ``shard_batches`` gives every rank a contiguous run of WHOLE batches, so an
N-GPU run produces exactly the single-process result; ``gather_params`` is the one collective (RCCL all_gather over xGMI,
latency-bound: 182 floats per frame).
"""
from __future__ import annotations

from typing import List, Tuple


def batches_of(num_frames: int, bs: int, start: int = 0, end: int | None = None) -> List[Tuple[int, int]]:
    end = num_frames if end is None else min(end, num_frames)
    return [(s, min(s + bs, end)) for s in range(start, end, bs)]


def shard_batches(num_frames: int, bs: int, world: int, rank: int, start: int = 0, end: int | None = None) -> List[Tuple[int, int]]:
    """Contiguous, batch-aligned share of rank ``rank``: the first (nb % world) ranks get one extra batch."""
    allb = batches_of(num_frames, bs, start, end)
    nb = len(allb)
    base, extra = divmod(nb, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return allb[lo:hi]


def shard_units(num_frames: int, bs_a: int, bs_b: int, world: int, rank: int) -> List[Tuple[int, int]]:
    """Contiguous share of rank ``rank`` in units of lcm(bs_a, bs_b) frames: two per-batch stages with different batch sizes (the SIF-Net pass at 64,
    the joint fit at 96: scripts/demo.sh:27, recon_fit_triplane.py:257) can then run over the SAME frames of a rank -- every unit boundary is a
    batch boundary of both stages, so each stage cuts exactly the batches the single-process run cuts (same results), and what the first stage
    leaves in HBM (the feature maps) is where the second one needs it."""
    import math
    return shard_batches(num_frames, math.lcm(bs_a, bs_b), world, rank)


def frame_range(batches: List[Tuple[int, int]]) -> Tuple[int, int]:
    """the --start/--end pair equivalent to a shard (empty shard -> (0, 0))"""
    return (batches[0][0], batches[-1][1]) if batches else (0, 0)


def gather_params(local, num_frames: int, bs: int, start: int = 0, end: int | None = None):
    """all_gather of per-frame parameter rows (T_rank, D) from every rank -> (T, D) in frame order on every rank.
    Works with any initialised torch.distributed backend (nccl == RCCL on ROCm; gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    import os
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and os.environ.get("VT_FORCE_DIST") != "1"):
        return local            # (VT_FORCE_DIST=1: a group of ONE still goes through the collective -- executes the RCCL path on a one-GPU test box)
    world = dist.get_world_size()
    counts = [sum(e - s for s, e in shard_batches(num_frames, bs, world, r, start, end)) for r in range(world)]
    D = local.shape[1]
    mx = max(counts)
    pad = torch.zeros(mx, D, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    # RCCL ("nccl") gathers device tensors in place; gloo (CPU tests, single-GPU dry runs with several ranks) has no device all_gather:
    # stage through the host there
    via_host = pad.is_cuda and dist.get_backend() == "gloo"
    send = pad.cpu() if via_host else pad
    outs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(outs, send)
    full = torch.cat([o[:c] for o, c in zip(outs, counts)], 0)
    return full.to(local.device) if via_host else full


class WorkQueue:
    """Run-time hand-out of work items (batch indices) to whichever rank / stream asks next.

    The reference leaves the split of a sequence to the user (``--start/--end`` per process, README.md:50-52, recon/recon_fit_base.py:411-419): contiguous
    equal-count shards leave the rank with the 60-frame tail batch idle for 38 % of a batch, and the stop rules make batches cost 734 .. 2580 Adam steps.
    Here every rank (and every stream of a rank) pulls the next index from ONE atomic counter in the process group's key-value store (``store.add``: no
    collective, no extra thread); items are served in ``order`` (e.g. longest first).  Results cannot depend on who fits what: a batch is an independent
    unit with its own random stream.  Without a process group (or with one rank) the counter is local.  Every rank must construct its queues in the same
    order (the key is numbered per process)."""
    _serial = 0

    def __init__(self, n_items: int, order=None, name: str = "vt_workqueue"):
        import threading
        self.n = int(n_items)
        self.order = list(order) if order is not None else list(range(self.n))
        assert sorted(self.order) == list(range(self.n)), "order must be a permutation of the items"
        self._lock = threading.Lock(); self._next = 0
        self.store = None
        WorkQueue._serial += 1
        self.key = f"{name}/{WorkQueue._serial}"
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from torch.distributed import distributed_c10d as c10d
                self.store = c10d._get_default_store()
        except Exception:          # noqa: BLE001 -- no store: every rank would serve itself the whole list; the caller must fall back to static shards
            self.store = None

    @property
    def shared(self) -> bool:
        return self.store is not None

    def next(self):
        """the next item, or None when all have been handed out"""
        if self.store is None:
            with self._lock:
                k = self._next; self._next += 1
        else:
            k = int(self.store.add(self.key, 1)) - 1
        return self.order[k] if k < self.n else None


class StealQueue:
    """Per-rank work lists with stealing: a rank takes its OWN items from the front (the batches whose feature maps it holds) and, when it has none
    left, takes items from the BACK of the list of the rank with the most left (paying whatever the caller pays for a foreign item -- the pipeline
    encodes the stolen batch's maps again, 0.26 s against a 0.65 .. 2.8 s fit).

    One packed atomic counter per owner in the process group's store: low ``SHIFT`` bits = items taken from the front, high bits = items taken from
    the back; ``store.add`` returns the value AFTER the add, a consistent snapshot of both, and a take is valid iff front + back <= count -- an
    overshooting take leaves the list exhausted for everybody, which it already was.  No collective, no server thread; without a process group the
    counters are local (a world of one rank never steals).  Every rank must construct its queues in the same order (the keys are numbered)."""
    SHIFT = 20
    MASK = (1 << 20) - 1
    _serial = 0

    def __init__(self, counts, rank: int, name: str = "vt_stealqueue"):
        import threading
        self.counts = [int(c) for c in counts]; self.rank = int(rank)
        assert max(self.counts + [0]) <= self.MASK
        StealQueue._serial += 1
        self.keys = [f"{name}/{StealQueue._serial}/{r}" for r in range(len(self.counts))]
        self._lock = threading.Lock(); self._local = [0] * len(self.counts); self._own_empty = False
        self.store = None
        self.stolen = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from torch.distributed import distributed_c10d as c10d
                self.store = c10d._get_default_store()
        except Exception:          # noqa: BLE001 -- no store: local counters, nobody else sees them; the caller must not rely on stealing then
            self.store = None

    @property
    def shared(self) -> bool:
        return self.store is not None

    def _add(self, r, inc):
        if self.store is not None:
            return int(self.store.add(self.keys[r], inc))
        with self._lock:
            self._local[r] += inc
            return self._local[r]

    def _split(self, v):
        return v & self.MASK, v >> self.SHIFT

    def next(self):
        """-> (owner rank, index in the owner's list) or None when every list is exhausted"""
        if not self._own_empty:
            f, b = self._split(self._add(self.rank, 1))
            if f + b <= self.counts[self.rank]:
                return self.rank, f - 1
            self._own_empty = True
        if self.store is None:
            return None
        while True:
            best = None
            for r in range(len(self.counts)):
                if r == self.rank:
                    continue
                f, b = self._split(self._add(r, 0))
                left = self.counts[r] - f - b
                if left > 0 and (best is None or left > best[0]):
                    best = (left, r)
            if best is None:
                return None
            r = best[1]
            f, b = self._split(self._add(r, 1 << self.SHIFT))
            if f + b <= self.counts[r]:
                with self._lock:
                    self.stolen += 1
                return r, self.counts[r] - b
            # lost the race for that owner's last item: look again


def all_ranks_agree(flag: bool, device=None) -> bool:
    """True iff ``flag`` is true on EVERY rank (one all_reduce MIN; a world of one rank: the flag itself).  Ranks that pick between two collective
    patterns from a rank-local fact -- e.g. "did I get the process group's store" -- must agree first, or the job hangs in mismatched collectives."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    # the flag lives where the group's backend can reduce it: host memory for gloo; for nccl (= RCCL, no CPU backend) the given device or, when the
    # caller named none, this process's current GPU -- never the CPU ("No backend type associated with device type cpu")
    if dist.get_backend() == "gloo":
        where = "cpu"
    else:
        where = device if (device is not None and torch.device(device).type == "cuda") else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=where)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def reduce_rows_exact(table, filled_rows):
    """Every rank holds ``table`` (T, D) float32 with the rows it produced (``filled_rows``: bool (T,)) and zeros elsewhere; returns the table with every
    row from the rank that produced it, bit for bit (the sum runs on the int32 view: adding integer zeros cannot change a bit pattern, not even the sign
    of a zero), and checks that every row was produced exactly once."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        assert bool(filled_rows.all()), "rows missing"
        return table
    via_host = table.is_cuda and dist.get_backend() == "gloo"
    bits = table.contiguous().view(torch.int32).clone(); cnt = filled_rows.to(torch.int32).clone()
    bits[~filled_rows] = 0
    if via_host:
        bits, cnt = bits.cpu(), cnt.cpu()
    dist.all_reduce(bits); dist.all_reduce(cnt)
    assert bool((cnt == 1).all()), "a batch was fitted by no rank or by two"
    return bits.to(table.device).view(torch.float32)
