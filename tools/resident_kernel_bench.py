"""RIFE step time beside a resident kernel on another stream — a stand-in for an overlapped RCCL all-gather's kernel — for
vfi_set_reserved_cus = 0 / 16 / 32 (profiles/r04_reserved_cus.txt).  Build the stand-in first:
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/resident_kernel.hip -o tools/micro/libresident_kernel.so"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.load_package()
from cfi_amd import _lib, synth
from cfi_amd.rife import RifeEngine
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libresident_kernel.so"))
hog.launch_hog.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
B, H, W = 32, 1080, 1920
eng = RifeEngine(synth.rife47_synth_state_dict(1234), "4.7")
eng.configure(H, W, B, B + 1, 1.0)
g = torch.Generator().manual_seed(0)
raw = torch.rand((B + 1, H, W, 3), generator=g).cuda()
out = torch.empty((B, H, W, 3), device="cuda")
sink = torch.zeros(1, dtype=torch.int32, device="cuda")
s0, s1, ts = list(range(B)), list(range(1, B + 1)), [0.5] * B
side = torch.cuda.Stream()
def step():
    eng.load_frames(list(range(B + 1)), [raw[j] for j in range(B + 1)])
    eng.interpolate(s0, s1, ts, out)
K = 5
# how long does the stand-in really stay resident?
torch.cuda.synchronize(); t0 = time.perf_counter()
hog.launch_hog(32, int(25e-3 * 100e6), ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(sink.data_ptr())); torch.cuda.synchronize()
print(f"hog alone (25 ms asked): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
lib = _lib.load()
ref = None
for reserve in (0, 16, 32):
    assert lib.vfi_set_reserved_cus(reserve) == 0
    for grid, ms in [(0, 0), (16, 25), (32, 25)]:
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            if grid:
                assert hog.launch_hog(grid, int(ms * 1e-3 * 100e6), ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(sink.data_ptr())) == 0
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        cur = out[:3].cpu()
        if ref is None:
            ref = cur
        print(f"reserved {reserve:2d} CUs | stand-in collective {grid:3d} workgroups x {ms:2d} ms per step: {dt * 1e3:8.3f} ms/step  {B / dt:7.1f} frames/s   "
              f"bit-identical to the first run: {bool(torch.equal(cur, ref))}", flush=True)
lib.vfi_set_reserved_cus(0)
