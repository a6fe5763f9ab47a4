"""Pinned H2D / D2H bandwidth on this box, alone and concurrently (sizing of the host pipeline)."""
import time
import torch

n = 24883200 // 4  # one 1080p RGB fp32 frame
host_a = [torch.empty(n, pin_memory=True) for _ in range(8)]
host_b = [torch.empty(n, pin_memory=True) for _ in range(8)]
dev_a = [torch.empty(n, device="cuda") for _ in range(8)]
dev_b = [torch.empty(n, device="cuda") for _ in range(8)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(8):
            if h2d:
                with torch.cuda.stream(s1):
                    dev_a[i].copy_(host_a[i], non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    host_b[i].copy_(dev_b[i], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = reps * 8 * n * 4
    return nbytes / dt / 1e9


for _ in range(2):
    print(f"H2D alone {run(True, False):.1f} GB/s; D2H alone {run(False, True):.1f} GB/s; both: {run(True, True):.1f} GB/s each direction", flush=True)
