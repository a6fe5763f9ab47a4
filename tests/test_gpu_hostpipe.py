"""Host pipeline on the device (comfyui-frame-interpolation_amd/hostpipe.py): uploads ahead of use, copy-back through pinned slots, and the copy
streams being the SAME objects in every call (a stream made per call alternates between hardware queues: profiles/r04_e2e_stream_reuse.txt)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_upload_download_round_trip_and_stream_reuse():
    from cfi_amd import hostpipe

    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    g = torch.Generator().manual_seed(3)
    frames = torch.rand(7, 40, 56, 4, generator=g)          # RGBA clip: the uploader drops alpha
    seen = []
    for rep in range(2):
        order = [3, 0, 6, 1, 5]
        fired = []
        # second repetition: the first two items go through the split (row-band) staging path and announce themselves
        up = hostpipe.Uploader(frames, order, dev, main, depth=2, on_staged=(lambda: fired.append(1)) if rep else None, staged_after=2 if rep else 0)
        assert up._split_n == (2 if rep else 0)
        down = hostpipe.Downloader(dev, (40, 56, 3), main, depth=3)
        seen.append((up.stream, down.stream))
        out = torch.zeros(len(order), 40, 56, 3)
        try:
            for i in range(len(order)):
                src = up.get(i)
                dst = (src * 2.0).contiguous()                # stands in for the compute stream's work
                up.release(i)
                done = torch.cuda.Event()
                done.record(main)
                main.wait_event(down.push(done, dst[None], [out[i]]))
        finally:
            up.close()
            down.close()
        for i, f in enumerate(order):
            assert torch.equal(out[i], frames[f][..., :3] * 2.0)
        assert fired == ([1] if rep else [])
    assert seen[0][0] is seen[1][0] and seen[0][1] is seen[1][1] and seen[0][0] is not seen[0][1]
    assert hostpipe._stream(dev, "up") is seen[0][0]
