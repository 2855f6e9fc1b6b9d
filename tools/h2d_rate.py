"""Host -> device copy rate of this box for a buffer the size of a bench BAM file (pinned memory, one stream and three)."""
import time
import torch

n = 26_300_000
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(3)]
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
s = [torch.cuda.Stream() for _ in range(3)]
for k in (1, 3):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60):
            with torch.cuda.stream(s[i % k]):
                d[i % k].copy_(h[i % k], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("streams", k, "GB/s %.1f" % (60 * n / dt / 1e9), "ms per file %.3f" % (dt / 60 * 1e3))
