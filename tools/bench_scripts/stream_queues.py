"""Which torch streams run concurrently?  HIP multiplexes streams onto a few hardware queues; two streams on the same queue serialise.  Every stream of a
list runs one spin kernel (torch.cuda._sleep); a pair that overlaps takes the time of one.  usage: stream_queues.py [n_streams=12]"""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ss = [torch.cuda.Stream() for _ in range(n)]
cyc = int(2e8)
def run(ids):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in ids:
        with torch.cuda.stream(ss[i]):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return time.perf_counter() - t0
base = run([0]); print(f"one spin kernel: {base * 1e3:.1f} ms")
print("pair (0, j) / one:", [round(run([0, j]) / base, 2) for j in range(1, n)])
print("pair (j, j + 1) / one:", [round(run([j, j + 1]) / base, 2) for j in range(n - 1)])
print("first k streams together / one:", [round(run(list(range(k))) / base, 2) for k in (2, 3, 4, 5, 6, 8)])
