"""Host-side pieces of the node pipeline that need no GPU (comfyui-frame-interpolation_amd/hostpipe.py): raw host copies,
background first-touch of the output tensor, pass-through row copies (contiguous and strided / RGBA sources)."""
import torch

from cfi_amd import hostpipe


def test_host_copy_contiguous_and_strided():
    g = torch.Generator().manual_seed(0)
    src = torch.rand(6, 20, 30, 4, generator=g)
    dst = torch.empty(20, 30, 4)
    hostpipe.host_copy(dst, src[2])                       # memmove path
    assert torch.equal(dst, src[2])
    dst3 = torch.empty(20, 30, 3)
    hostpipe.host_copy(dst3, src[3][..., :3])             # strided source (alpha drop) -> torch copy path
    assert torch.equal(dst3, src[3][..., :3])
    dst64 = torch.empty(20, 30, 3, dtype=torch.float64)
    hostpipe.host_copy(dst64, src[1][..., :3])            # dtype change -> torch copy path
    assert torch.equal(dst64, src[1][..., :3].double())


def test_prefault_then_fill():
    out = torch.empty(9, 64, 96, 3)                       # ~ 0.66 MB = 11 chunks of 64 KiB, swept by ONE task per worker
    futs = hostpipe.prefault_async(out, chunk=1 << 16, workers=3)
    assert len(futs) == 3 and len(hostpipe.prefault_async(out, chunk=1 << 20, workers=3)) == 1
    for f in futs:
        f.result()
    out.fill_(1.5)                                        # pages are usable afterwards
    assert float(out.sum()) == 1.5 * out.numel()
    assert hostpipe.prefault_async(out[:, ::2]) == []     # non-contiguous: nothing to do
    assert hostpipe.prefault_async(out, workers=0) == []


def test_copy_rows_async_matches_index_copy():
    g = torch.Generator().manual_seed(1)
    frames = torch.rand(5, 16, 24, 4, generator=g)
    out = torch.zeros(9, 16, 24, 3)
    rows, idx = [0, 2, 4, 6, 8], [0, 1, 2, 3, 4]
    for f in hostpipe.copy_rows_async(out, rows, frames, idx):
        f.result()
    for r, j in zip(rows, idx):
        assert torch.equal(out[r], frames[j][..., :3])
    assert float(out[1::2].abs().sum()) == 0.0


def test_short_switch_interval_is_reentrant_across_threads():
    """rife._short_switch_interval: the first thread in saves the interpreter's switch interval, the last one out restores it."""
    import sys
    import threading
    from cfi_amd import rife

    before = sys.getswitchinterval()
    inside, leave = threading.Barrier(3, timeout=30), threading.Event()
    seen = []

    def user():
        with rife._short_switch_interval(1e-4):
            seen.append(sys.getswitchinterval())
            inside.wait()
            leave.wait(30)

    ts = [threading.Thread(target=user, daemon=True) for _ in range(2)]
    for t in ts:
        t.start()
    try:
        inside.wait()
        during = sys.getswitchinterval()
    finally:
        leave.set()
    for t in ts:
        t.join(30)
    assert len(seen) == 2 and all(abs(x - 1e-4) < 1e-6 for x in seen + [during])
    assert sys.getswitchinterval() == before
