"""-m gpu: the device side of the single-process multi-device path (csrc/comm.hip, multidev.py) on the ONE GPU of the test box:
the RCCL clique in its degenerate one-device form (same calls, real RCCL), the weight-arena clone, and two host threads with
their own engines / streams / pipelines sharing one device.  More than one device cannot be had here; the N-device shard /
assemble logic is covered by tests/test_multidev_cpu.py (N = 8, uneven blocks)."""
import ctypes as C

import pytest
import torch

from cfi_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    return synth.rife47_synth_state_dict(1234)


def test_rccl_clique_of_one_device(hip_lib):
    from cfi_amd import multidev

    torch.cuda.set_device(0)
    comm = multidev.Comm([0])
    try:
        assert hip_lib.vfi_comm_size(comm.handle) == 1
        x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
        want = x.clone()
        torch.cuda.synchronize()
        comm.broadcast([x.data_ptr()], x.numel(), root=0)
        comm.all_gather_v([x.data_ptr()], [x.numel()])
        comm.synchronize()
        assert torch.equal(x, want)
    finally:
        comm.close()
    with pytest.raises(RuntimeError, match="listed twice|not visible"):
        multidev.Comm([0, 0])


def _clone_with_weights(engine):
    """second engine on the same device, arena filled by a device-to-device copy (what the RCCL broadcast does across devices)"""
    from cfi_amd.rife import RifeEngine

    clone = RifeEngine.clone_empty(engine, engine.device)
    (src, n), (dst, m) = engine.weights(), clone.weights()
    assert n == m and n * 4 > 21_000_000 and src != dst          # 5.3 M parameters, packed / padded
    a = (C.c_float * n).from_address  # noqa: F841  (device addresses: not dereferenced on the host)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), n * 4, 3) == 0       # hipMemcpyDeviceToDevice
    return clone


def test_cloned_network_is_bit_identical(hip_lib, sd):
    from cfi_amd.rife import RifeEngine, run_tasks

    torch.cuda.set_device(0)
    e0 = RifeEngine(sd, "4.7")
    e1 = _clone_with_weights(e0)
    try:
        frames = synth.smooth_frames(3, 136, 200, seed=4, shift=3.0)
        tasks = [(0, 0.5), (1, 0.25), (1, 0.75)]
        a = run_tasks(e0, frames, tasks, batch_size=2)
        b = run_tasks(e1, frames, tasks, batch_size=2)
        assert torch.equal(a, b)
    finally:
        e0.close()
        e1.close()


def test_two_host_threads_share_one_device(hip_lib, sd):
    """RifeDeviceGroup.run with two members on the same GPU: two host threads, each with its own engine, streams, upload /
    download pipelines and ring sets, filling disjoint rows of one host tensor — equal to the single-engine result."""
    from cfi_amd import multidev
    from cfi_amd.rife import RifeEngine, run_tasks
    from cfi_amd.schedule import rife_output_plan, rife_task_list

    torch.cuda.set_device(0)
    e0 = RifeEngine(sd, "4.7")
    e1 = _clone_with_weights(e0)
    try:
        frames = synth.smooth_frames(9, 136, 200, seed=5, shift=3.0)
        _, tasks = rife_task_list(9, 3, None)                    # 16 tasks -> 8 + 8
        plan = rife_output_plan(9, tasks)
        rows = [0] * len(tasks)
        for i, (kind, idx) in enumerate(plan):
            if kind == "new":
                rows[idx] = i
        want = torch.zeros(len(plan), 136, 200, 3)
        run_tasks(e0, frames, tasks, 4, out=want, out_rows=rows)
        group = multidev.RifeDeviceGroup.__new__(multidev.RifeDeviceGroup)
        group.devices, group.engines, group.comm, group.owns_primary = [0, 0], [e0, e1], None, True
        got = torch.zeros(len(plan), 136, 200, 3)
        bounds = group.run(frames, tasks, 4, 1.0, got, rows)
        assert bounds == [(0, 8), (8, 16)]
        assert torch.equal(got, want)
    finally:
        e0.close()
        e1.close()


def test_node_with_device_selection(hip_lib, sd, tmp_path, monkeypatch):
    """VFI_DEVICES=all on a one-GPU box degenerates to the plain single-device path (no clique, no threads)."""
    import cfi_amd.rife as R

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    frames = synth.smooth_frames(3, 72, 104, seed=6, shift=2.0)
    try:
        (ref,) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=2)
        monkeypatch.setenv("VFI_DEVICES", "all")
        (out,) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=2)
        assert torch.equal(out, ref)
    finally:
        for e in R._model_cache.values():
            e.close()
        R._model_cache.clear()


_DIRECT_SNIPPET = """
import sys
sys.path.insert(0, {root!r})
import torch
from pkgload import load_package
load_package()
from cfi_amd import _lib, multidev
lib = _lib.load()
assert lib.vfi_comm_all_gather_mode() == 0, "VFI_ALLGATHER=direct not honoured"
n = torch.cuda.device_count()
devs = list(range(n))
comm = multidev.Comm(devs)
per = 1 << 18
counts = [per + 1000 * r for r in range(n)]
total = sum(counts)
bufs, off = [], 0
for r, d in enumerate(devs):
    torch.cuda.set_device(d)
    b = torch.full((total,), -1.0, device=f"cuda:{{d}}")
    b[off:off + counts[r]] = float(r + 1)
    bufs.append(b)
    off += counts[r]
for d in devs:
    torch.cuda.synchronize(d)
torch.cuda.set_device(0)
for rep in range(2):
    comm.all_gather_v([b.data_ptr() for b in bufs], counts)
    # a library kernel launch on THIS thread right after the collective: a sticky HIP error left behind by the mesh set-up
    # (hipErrorPeerAccessAlreadyEnabled) would surface in its hipGetLastError check
    x = torch.rand(1, 8, 8, 4, device="cuda:0")
    y = torch.empty(1, 16, 16, 4, device="cuda:0")
    assert torch.cuda.current_device() == 0, "the collective moved the thread's current device"
    _lib.check(lib.vfi_upsample_nearest(x.data_ptr(), 4, y.data_ptr(), 4, 1, 8, 8, 16, 16, 4, None), "kernel after all_gather_v")
    comm.synchronize()
want = torch.cat([torch.full((c,), float(r + 1)) for r, c in enumerate(counts)])
for b in bufs:
    assert torch.equal(b.cpu(), want)
comm.close()
print("direct all-gather ok on", n, "device(s)")
"""


def test_direct_all_gather_then_kernel(hip_lib):
    """The direct full-mesh all-gather (VFI_ALLGATHER=direct; opt-in until this test has passed on two or more devices) on EVERY
    visible device, unequal blocks, twice, with a library kernel launch right behind it on the issuing thread (the mesh
    set-up must not leave hipErrorPeerAccessAlreadyEnabled in the thread's sticky error slot, nor move its current device).  On the
    one-GPU test box the clique has one member (no peer pair: only the plumbing runs); the driver's 8-GPU node runs the real thing."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DIRECT_SNIPPET.format(root=root)], env=dict(os.environ, VFI_ALLGATHER="direct"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "direct all-gather ok" in r.stdout
