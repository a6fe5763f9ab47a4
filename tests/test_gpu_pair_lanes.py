"""-m gpu: pair lanes (comfyui-frame-interpolation_amd/lanes.py) — K engines on K HIP streams, pairs of a clip round robin over them.
The clip's frames must be BIT-IDENTICAL to the single-stream loop's for every model that takes the generic node loop (the lanes
share nothing writable: per-engine workspaces, library scratch keyed by (device, stream)), for clips that keep every lane busy,
for skipped pairs and list multipliers, and when engines of different lanes splat at the same time."""
import pytest
import torch

from cfi_amd import synth
from cfi_amd.lanes import LaneSet, lanes_for, lanes_of
from cfi_amd.schedule import InterpolationStateList, generic_output_plan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    from cfi_amd import _lib

    _lib.check(hip_lib.vfi_init(0), "vfi_init")
    return hip_lib


def _factory(model):
    if model == "m2m":
        from cfi_amd.m2m import M2MEngine
        sd = synth.m2m_synth_state_dict(1234)
        return lambda: M2MEngine(sd)
    if model == "gmfss":
        from cfi_amd.gmfss import GMFSSEngine
        # the matched ("coherent") checkpoint: with random weights GMFlow's flow is noise of hundreds of pixels, the splats take their
        # atomic spill path and ONE stream already differs from itself by 1.5e-6 from run to run
        sds = synth.gmfss_coherent_state_dicts(3, "union")
        return lambda: GMFSSEngine(sds)
    if model == "ifunet":
        from cfi_amd.ifunet import IFUNetEngine
        sd = synth.ifunet_synth_state_dict(1234)
        return lambda: IFUNetEngine(sd)
    from cfi_amd.ifrnet import IFRNetEngine
    sd = synth.ifrnet_synth_state_dict("S", 1234)
    return lambda: IFRNetEngine(sd, "S")


@pytest.mark.parametrize("model,multiplier,states", [
    ("m2m", 2, None), ("m2m", [3, 0, 2, 1, 2, 4], None), ("m2m", 3, InterpolationStateList([1, 4], True)),
    ("ifrnet", 2, None), ("ifunet", 2, None), ("gmfss", 2, None), ("gmfss", 3, InterpolationStateList([2], True))])
def test_lanes_bit_identical_to_single_stream(lib, model, multiplier, states):
    from cfi_amd.m2m import run_plan

    H, W = (128, 192) if model != "gmfss" else (192, 256)
    # 6 pairs: two rounds over three lanes; RGBA clip
    fr = synth.smooth_frames(7, H, W, seed=11, shift=2.5, c=4) if model != "gmfss" else synth.texture_frames(7, H, W, seed=5)
    plan, tasks = generic_output_plan(len(fr), multiplier, states)
    build = _factory(model)
    single = build()
    try:
        want = run_plan(single, fr, plan, tasks)
        torch.cuda.synchronize()
    finally:
        if hasattr(single, "close"):
            single.close()
    lanes = LaneSet(build, 3)
    try:
        assert lanes_of(lanes, 6)[1] == 3 and lanes_of(lanes, 1)[1] == 1 and lanes_of(lanes, 3)[1] == 1 and lanes_of(lanes, 4)[1] == 2
        for rep in range(2):      # the second call finds every lane built and its workspace warm
            got = run_plan(lanes, fr, plan, tasks)
            torch.cuda.synchronize()
            assert got.shape == want.shape
            # (GMFSS: its splats' atomic spill pass sums in hardware order — one stream already differs from itself by ~1.5e-6 now and then)
            same = torch.equal(got, want) if model != "gmfss" else (got - want).abs().max().item() <= 5e-6
            assert same, f"{model}: lanes differ from the single-stream loop by {(got - want).abs().max().item():.3e} (call {rep})"
        assert len(lanes.engines) == min(3, max(1, len(tasks) // 2))
    finally:
        lanes.close()


def test_film_node_lanes_bit_identical(lib, tmp_path, monkeypatch):
    import cfi_amd.film as FM
    from cfi_amd import ckpt

    sd = synth.film_synth_state_dict(1234)
    pth = tmp_path / "film_net_fp32.pt"
    torch.save(sd, pth)
    monkeypatch.setattr(FM, "load_file_from_github_release", lambda model_type, ck: str(pth))
    import cfi_amd.lanes as LN

    monkeypatch.setitem(LN.PAIRS_PER_LANE, "film", 1)      # (the node opens a FILM lane per 24 pairs; this clip has 4 kept ones)
    frames = synth.smooth_frames(6, 64, 80, seed=2, shift=1.5, c=4)
    outs = {}
    for k in ("1", "3"):
        monkeypatch.setenv("VFI_PAIR_LANES", k)
        ckpt.clear_engine_cache()
        assert lanes_for("film") == int(k)
        (outs[k],) = FM.FILM_VFI().vfi("film_net_fp32.pt", frames, multiplier=[3, 2, 1, 4, 2],
                                       optional_interpolation_states=InterpolationStateList([1], True))
    ckpt.clear_engine_cache()
    assert outs["1"].shape == outs["3"].shape and torch.equal(outs["1"], outs["3"])


def test_concurrent_splats_do_not_share_scratch(lib):
    """vfi_softsplat_sum on three streams at once (what GMFSS lanes do): the block ranges, spill list and control words are per
    (device, stream).  With one set per device the lanes overwrote each other's tile ranges (round 6 first probe)."""
    import ctypes as C

    from cfi_amd import _lib

    H, W, Cc = 270, 480, 4
    g = torch.Generator().manual_seed(3)
    ins = [torch.rand(1, H, W, Cc, generator=g).cuda() for _ in range(3)]
    flows = [(torch.randn(1, H, W, 2, generator=g) * s).cuda() for s in (1.0, 6.0, 20.0)]
    want = []
    for x, f in zip(ins, flows):
        o = torch.zeros(1, H, W, Cc, device="cuda")
        _lib.check(lib.vfi_softsplat_sum(x.data_ptr(), f.data_ptr(), o.data_ptr(), 1, H, W, Cc, _lib.stream_ptr()), "splat")
        want.append(o)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    for rep in range(8):
        got = [torch.zeros(1, H, W, Cc, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        for k in range(3):
            with torch.cuda.stream(streams[k]):
                _lib.check(lib.vfi_softsplat_sum(ins[k].data_ptr(), flows[k].data_ptr(), got[k].data_ptr(), 1, H, W, Cc, _lib.stream_ptr()), "splat")
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(got[k], want[k]), f"stream {k}, repetition {rep}"


def test_lane_streams_sit_on_different_hardware_queues(lib):
    """The HIP runtime binds streams to a handful of hardware queues; two lanes on one queue run in turn (M2M: 6.75 instead of 6.17 ms per
    pair).  The probe (vfi_stream_spin on both streams) must call a stream a queue-mate of itself, and own_streams_apart must hand out
    lanes that are pairwise on different queues."""
    from cfi_amd import _lib

    dev = torch.device("cuda", 0)
    one = _lib.OwnStream(dev)
    try:
        assert _lib.streams_share_queue(one.stream, one.stream)
    finally:
        one.release()
    got = _lib.own_streams_apart(dev, 3)
    try:
        assert len({o.ptr for o in got}) == 3
        for i in range(3):
            for j in range(i + 1, 3):
                assert not _lib.streams_share_queue(got[i].stream, got[j].stream), (i, j)
    finally:
        for o in got:
            o.release()
