"""CPU: the host logic of the single-process multi-device path (comfyui-frame-interpolation_amd/multidev.py): block partition of
the task list over N devices, one host thread per device writing ITS shard into the shared output tensor, error propagation.
The device side (RCCL clique, weight broadcast, per-device engines) is covered on the GPU box by tests/test_gpu_multidev.py."""
import threading

import pytest
import torch

from cfi_amd import multidev
from cfi_amd.schedule import rife_output_plan, rife_task_list


@pytest.mark.parametrize("n_dev,n_frames,mult", [(8, 14, 2), (8, 5, 4), (8, 3, 2), (3, 10, 3), (1, 6, 2), (8, 2, 2)])
def test_shards_cover_the_task_list_and_assemble_in_order(n_dev, n_frames, mult):
    """N = 8 with uneven blocks (13 tasks), more devices than tasks (2 tasks), one device: every task exactly once, blocks
    contiguous and in device order, every new frame lands in its final output row."""
    _, tasks = rife_task_list(n_frames, mult, None)
    plan = rife_output_plan(n_frames, tasks)
    new_rows = [0] * len(tasks)
    for i, (kind, idx) in enumerate(plan):
        if kind == "new":
            new_rows[idx] = i
    out = torch.full((len(plan), 2), -1.0)
    seen = []
    lock = threading.Lock()

    def fn(rank, dev, lo, hi):
        assert dev == 100 + rank
        for i in range(lo, hi):                       # what RifeDeviceGroup.run does with run_tasks(..., out=out, out_rows=...)
            pair, t = tasks[i]
            out[new_rows[i]] = torch.tensor([float(pair), float(t)])
        with lock:
            seen.append((rank, lo, hi, threading.current_thread().name))

    bounds = multidev.run_sharded([100 + r for r in range(n_dev)], len(tasks), fn, set_device=False)
    assert [b for b in bounds if b[1] > b[0]] == sorted((lo, hi) for _, lo, hi, _ in seen)
    sizes = [hi - lo for lo, hi in bounds]
    assert sum(sizes) == len(tasks) and max(sizes) - min(sizes) <= 1
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(n_dev - 1)) and bounds[0][0] == 0 and bounds[-1][1] == len(tasks)
    if n_dev > 1 and len(tasks) >= 2:
        assert len({name for *_, name in seen}) == len(seen), "every non-empty block on its own host thread"
    for i, (kind, idx) in enumerate(plan):
        if kind == "new":
            assert out[i].tolist() == [float(tasks[idx][0]), pytest.approx(tasks[idx][1])]
        else:
            assert out[i].tolist() == [-1.0, -1.0]      # pass-through rows are not the devices' business


def test_an_error_on_one_device_surfaces_after_all_threads_finished():
    done = []

    def fn(rank, dev, lo, hi):
        if rank == 5:
            raise RuntimeError("device 5 failed")
        done.append(rank)

    with pytest.raises(RuntimeError, match="device 5 failed"):
        multidev.run_sharded(list(range(8)), 16, fn, set_device=False)
    assert sorted(done) == [0, 1, 2, 3, 4, 6, 7]


def test_device_selection(monkeypatch):
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 2)
    monkeypatch.delenv("VFI_DEVICES", raising=False)
    assert multidev.selected_devices() == [2]                                   # default: ComfyUI's device only
    monkeypatch.setenv("VFI_DEVICES", "all")
    assert multidev.selected_devices() == [2, 0, 1, 3, 4, 5, 6, 7]             # the caller's device stays the primary
    monkeypatch.setenv("VFI_DEVICES", "4,2,5")
    assert multidev.selected_devices() == [2, 4, 5]      # the engine's device is the primary (it holds the weight arena = broadcast root)
    monkeypatch.setenv("VFI_DEVICES", "4,5")
    with pytest.raises(ValueError, match="leaves out device 2"):      # never silently add a GPU the user excluded
        multidev.selected_devices()
    monkeypatch.setenv("VFI_DEVICES", "1,1")
    with pytest.raises(ValueError):
        multidev.selected_devices()
    monkeypatch.setenv("VFI_DEVICES", "9")
    with pytest.raises(ValueError):
        multidev.selected_devices()


def test_all_gather_plan_offsets(hip_lib):
    """vfi_comm_plan_all_gather (the copy list the direct full-mesh all-gather executes): applied to host arrays with unequal and
    empty shards it reproduces the in-place all-gather, and every ordered pair appears at most once (one xGMI link each)."""
    import ctypes as C

    import numpy as np

    for counts in ([5, 3, 0, 7], [4, 4], [0, 0, 9], [1], [2, 0, 0, 0, 0, 0, 0, 6]):
        n, tot = len(counts), sum(counts)
        carr = (C.c_int64 * n)(*counts)
        k = hip_lib.vfi_comm_plan_all_gather(n, carr, None, 0)
        assert k == sum(n - 1 for c in counts if c > 0)
        plan = (C.c_int64 * (4 * max(k, 1)))()
        assert hip_lib.vfi_comm_plan_all_gather(n, carr, plan, 4 * max(k, 1)) == k
        bufs = [np.full(tot, -1.0, np.float32) for _ in range(n)]
        off = 0
        for r, c in enumerate(counts):               # every rank holds its own block
            bufs[r][off:off + c] = 100 * r + np.arange(c)
            off += c
        want = np.concatenate([100 * r + np.arange(c, dtype=np.float32) for r, c in enumerate(counts)]) if tot else np.zeros(0, np.float32)
        pairs = set()
        for j in range(k):
            src, dst, o, c = (int(plan[4 * j + i]) for i in range(4))
            assert src != dst and (src, dst) not in pairs
            pairs.add((src, dst))
            bufs[dst][o:o + c] = bufs[src][o:o + c]
        for b in bufs:
            assert np.array_equal(b, want)
    bad = (C.c_int64 * 2)(3, -1)
    assert hip_lib.vfi_comm_plan_all_gather(2, bad, None, 0) < 0
