"""Streams that really run concurrently.

HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); two streams that land on the same queue execute one after the
other, and which streams share a queue is not a simple function of their creation order (measured with torch's stream pool on the MI355X boxes,
tools/bench_scripts/stream_queues.py: of twelve consecutive streams the pairs (3, 4), (0, 7) and (0, 11) serialise).  The fit keeps two batches in flight
on two streams and the pipeline runs the encoders beside the surface-point generator: on a colliding pair all of that overlap is silently lost (the
full-schedule leg of bench.py: 2.79 s per batch on two fresh streams that collided, 2.50 s on two that did not).

``concurrent_streams(n)`` hands out n streams that were TESTED: every candidate runs a short spin kernel beside the streams already accepted (and beside
the current stream, which the caller usually keeps working on), and is accepted only if the pair takes the time of one.  The accepted streams are cached
per device and per current stream, so a process pays the test (a few milliseconds) once."""
from __future__ import annotations

import os
import threading
import time

import torch

_cache = {}
_lock = threading.Lock()
_SPIN = int(4e6)            # cycles of torch.cuda._sleep: ~2 ms


def _timed(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(_SPIN)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def overlaps(a, b) -> bool:
    """do streams a and b execute concurrently (different hardware queues)?  Wall-clock test around device-wide synchronisations: call it while the
    device is otherwise IDLE (another thread's GPU work lengthens both measurements and can make an overlapping pair look serial); the minimum over
    four repetitions keeps a stray host hiccup from rejecting a good pair."""
    one = min(_timed([a]) for _ in range(4))
    two = min(_timed([a, b]) for _ in range(4))
    return two < 1.5 * one


def concurrent_streams(n: int, device=None, with_current: bool = True, max_tries: int = 24):
    """n streams of ``device`` (default: the current device) that run concurrently with each other and, ``with_current``, with the current stream.
    Falls back to untested streams (with a warning) if the device does not offer that many queues."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if os.environ.get("VT_UNTESTED_STREAMS") == "1":          # A/B switch: plain new streams, as before round 4
        return [torch.cuda.Stream(device=dev) for _ in range(n)]
    with torch.cuda.device(dev):
        cur = torch.cuda.current_stream()
        key = (dev.index, cur.cuda_stream if with_current else None)
        with _lock:
            good = _cache.setdefault(key, [])
            tries = 0
            _timed([cur])                                    # clocks up before the first comparison
            try:
                while len(good) < n and tries < max_tries:
                    tries += 1
                    s = torch.cuda.Stream(device=dev)
                    if all(overlaps(g, s) for g in ([cur] if with_current else []) + good):
                        good.append(s)
            except (AttributeError, RuntimeError):          # no spin kernel in this torch build / the device is being captured: untested streams
                pass
            if len(good) < n:
                import warnings
                warnings.warn(f"concurrent_streams: only {len(good)} of {n} mutually concurrent streams found on {dev}; the rest may share a hardware queue")
                # the untested fill-ins are NOT cached: a later call must not take them for tested streams (nor test new candidates against them)
                return list(good) + [torch.cuda.Stream(device=dev) for _ in range(n - len(good))]
            return list(good[:n])
