"""How fast is hipHostRegister on the caller's pageable clip / on a fresh output tensor, per frame-sized slice?  Decides whether the
node can DMA straight from / into the caller's memory instead of staging through a pinned ring."""
import time
from concurrent.futures import ThreadPoolExecutor

import torch

rt = torch.cuda.cudart()
N, H, W = 33, 1080, 1920
frames = torch.rand(N, H, W, 3)          # pageable, touched
nb = frames[0].numel() * 4
dev = torch.empty_like(frames, device="cuda")
torch.cuda.synchronize()


def reg(t):
    return rt.cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)


def unreg(t):
    return rt.cudaHostUnregister(t.data_ptr())


for rep in range(3):
    t0 = time.perf_counter()
    rc = [reg(frames[i]) for i in range(N)]
    t1 = time.perf_counter()
    for i in range(N):
        dev[i].copy_(frames[i], non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(N):
        unreg(frames[i])
    t3 = time.perf_counter()
    print(f"input, serial: register {N} x {nb / 1e6:.1f} MB {1e3 * (t1 - t0):.1f} ms ({N * nb / (t1 - t0) / 1e9:.1f} GB/s), H2D {1e3 * (t2 - t1):.1f} ms "
          f"({N * nb / (t2 - t1) / 1e9:.1f} GB/s), unregister {1e3 * (t3 - t2):.1f} ms, rc {set(map(int, rc))}", flush=True)

for workers in (4, 8):
    with ThreadPoolExecutor(workers) as ex:
        t0 = time.perf_counter()
        list(ex.map(reg, [frames[i] for i in range(N)]))
        t1 = time.perf_counter()
        list(ex.map(unreg, [frames[i] for i in range(N)]))
        t2 = time.perf_counter()
    print(f"input, {workers} threads: register {1e3 * (t1 - t0):.1f} ms ({N * nb / (t1 - t0) / 1e9:.1f} GB/s), unregister {1e3 * (t2 - t1):.1f} ms", flush=True)

# whole clip in one call
t0 = time.perf_counter()
rc = reg(frames)
t1 = time.perf_counter()
unreg(frames)
t2 = time.perf_counter()
print(f"input, one call: register {1e3 * (t1 - t0):.1f} ms ({N * nb / (t1 - t0) / 1e9:.1f} GB/s), unregister {1e3 * (t2 - t1):.1f} ms rc {int(rc)}", flush=True)

# output: fresh (unfaulted) tensor
for workers in (1, 8):
    out = torch.empty(2 * N - 1, H, W, 3)
    rows = [out[i] for i in range(out.shape[0])]
    with ThreadPoolExecutor(workers) as ex:
        t0 = time.perf_counter()
        list(ex.map(reg, rows))
        t1 = time.perf_counter()
    src = torch.rand(8, H, W, 3, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(out.shape[0]):
        out[i].copy_(src[i % 8], non_blocking=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(unreg, rows))
    t4 = time.perf_counter()
    print(f"output (fresh, {workers} threads): register {out.shape[0]} rows {1e3 * (t1 - t0):.1f} ms ({out.numel() * 4 / (t1 - t0) / 1e9:.1f} GB/s), D2H "
          f"{1e3 * (t3 - t2):.1f} ms ({out.numel() * 4 / (t3 - t2) / 1e9:.1f} GB/s), unregister {1e3 * (t4 - t3):.1f} ms", flush=True)
    del out
