"""Do two HIP streams run concurrently?  torch hands out streams from a 32-entry pool per device; the runtime maps streams onto a
few hardware queues.  For every pair among (the default stream, the first 12 pool streams): two torch.cuda._sleep spin kernels
(one workgroup each), one per stream, timed together — ~T if they overlap, ~2T if they share a queue.
usage (GPU box): python tools/stream_overlap_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
cur = torch.cuda.current_stream(dev)
streams = [cur] + [torch.cuda.Stream(device=dev) for _ in range(12)]
N = 2_000_000


def pair(a, b):
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(N)
    with torch.cuda.stream(b):
        torch.cuda._sleep(N)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


pair(streams[0], streams[1])
single = min(pair(streams[1], streams[1]) for _ in range(3)) / 2
print("one spin kernel: %.3f ms; handles: %s" % (single, [hex(s.cuda_stream) for s in streams]))
print("rows/cols = default, pool 0..11; entry = time of the pair / time of one kernel (1 = concurrent, 2 = serialised)")
for i, a in enumerate(streams):
    print(" ".join("%.1f" % (min(pair(a, b) for _ in range(2)) / single) if j != i else " - " for j, b in enumerate(streams)))
