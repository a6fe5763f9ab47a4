"""-m gpu: vfi_film_run / vfi_m2m_run — the whole FILM / M2M node call behind ONE C entry point each (host clip in, host clip out)
— against the Python node classes (which are pinned against the reference nodes' goldens): int and list multipliers, skipped
pairs (FILM drops them, M2M keeps the frame), the m == 0 / local-skip-index quirks of generic_frame_loop, RGBA input."""
import ctypes as C

import pytest
import torch

from gpu_util import describe_diff
from cfi_amd import _lib, synth
from cfi_amd.schedule import InterpolationStateList

pytestmark = pytest.mark.gpu


def _call(fn, handle, frames, multiplier, mult_list, skip):
    from cfi_amd import _lib

    n, h, w, c = frames.shape
    ml = (C.c_int * len(mult_list))(*mult_list) if mult_list is not None else None
    sk = (C.c_uint8 * (n - 1))(*skip) if skip is not None else None
    n_out = C.c_int64(0)
    _lib.check(fn(handle, None, n, h, w, c, multiplier, ml, len(mult_list) if mult_list is not None else 0, sk, None, C.byref(n_out)), "size query")
    out = torch.full((n_out.value, h, w, 3), float("nan"))
    _lib.check(fn(handle, frames.data_ptr(), n, h, w, c, multiplier, ml, len(mult_list) if mult_list is not None else 0, sk, out.data_ptr(),
                  C.byref(n_out)), "run")
    return out


@pytest.mark.parametrize("kw,mult,mlist,skip", [
    (dict(multiplier=2), 2, None, None),
    (dict(multiplier=3, optional_interpolation_states=InterpolationStateList([1], True)), 3, None, [0, 1, 0]),
    (dict(multiplier=[4, 2]), 0, [4, 2], None),
    (dict(multiplier=9), 9, None, None),          # 9 frames per pair: the first multiplier whose bisection has exact ties (fp32 linspace bits decide)
    # list multipliers below 2: inference(..., inter_frames = m - 1) runs no iteration for m in {-1, 0, 1} and the reference node
    # emits frame_i alone (film/__init__.py:12-41,98-99) -> rows 1 + 3 + 1 + last frame
    (dict(multiplier=[0, 3, -1]), 0, [0, 3, -1], None),
])
def test_film_run_equals_the_node(hip_lib, tmp_path, monkeypatch, kw, mult, mlist, skip):
    import cfi_amd.film as FM
    from cfi_amd import ckpt

    sd = synth.film_synth_state_dict(1234)
    pth = tmp_path / "film_net_fp32.pt"
    torch.save(sd, pth)
    monkeypatch.setattr(FM, "load_file_from_github_release", lambda model_type, ckpt_: str(pth))
    ckpt.clear_engine_cache()
    frames = synth.smooth_frames(4, 64, 96, seed=3, shift=1.5, c=4).contiguous()
    (want,) = FM.FILM_VFI().vfi("film_net_fp32.pt", frames, **kw)
    ckpt.clear_engine_cache()
    eng = FM.FilmEngine(sd)
    try:
        out = _call(hip_lib.vfi_film_run, eng.handle, frames, mult, mlist, skip)
    finally:
        eng.close()
    assert out.shape == want.shape and not torch.isnan(out).any()
    assert (out - want).abs().max().item() <= 2e-5, describe_diff(out, want, "vfi_film_run vs the node")
    if mlist == [0, 3, -1]:
        assert out.shape[0] == 6 and torch.equal(out[0], frames[0, ..., :3]) and torch.equal(out[1], frames[1, ..., :3]) \
            and torch.equal(out[4], frames[2, ..., :3]) and torch.equal(out[5], frames[3, ..., :3])


def test_multipliers_the_reference_fails_on(hip_lib, tmp_path, monkeypatch):
    """FILM: torch.linspace(0, 1, m + 1) raises for m <= -2 (film/__init__.py:22); generic_frame_loop allocates
    torch.zeros(m * 2, ...) per pair of a multiplier LIST and raises for m < 0 (vfi_utils.py:178,364-371).  Node classes and the
    whole-clip C entries refuse the same inputs."""
    import cfi_amd.film as FM
    from cfi_amd import ckpt, m2m

    frames = synth.smooth_frames(3, 64, 64, seed=5, shift=1.0).contiguous()
    n_out = C.c_int64(0)
    sdf = synth.film_synth_state_dict(1234)
    pth = tmp_path / "film_net_fp32.pt"
    torch.save(sdf, pth)
    monkeypatch.setattr(FM, "load_file_from_github_release", lambda model_type, ckpt_: str(pth))
    ckpt.clear_engine_cache()
    with pytest.raises(RuntimeError, match="linspace"):
        FM.FILM_VFI().vfi("film_net_fp32.pt", frames, multiplier=[2, -2])
    ckpt.clear_engine_cache()
    eng = FM.FilmEngine(sdf)
    try:
        ml = (C.c_int * 2)(2, -2)
        assert hip_lib.vfi_film_run(eng.handle, None, 3, 64, 64, 3, 0, ml, 2, None, None, C.byref(n_out)) != 0
        assert "linspace" in _lib.last_error()
    finally:
        eng.close()
    sdm = synth.m2m_synth_state_dict(1234)
    pm = tmp_path / "M2M.pth"
    torch.save(sdm, pm)
    monkeypatch.setattr(m2m, "load_file_from_github_release", lambda model_type, ckpt_: str(pm))
    ckpt.clear_engine_cache()
    with pytest.raises(ValueError, match="negative"):
        m2m.M2M_VFI().vfi("M2M.pth", frames, multiplier=[2, -1])
    ckpt.clear_engine_cache()
    eng = m2m.M2MEngine(sdm)
    try:
        ml = (C.c_int * 2)(2, -1)
        assert hip_lib.vfi_m2m_run(eng.handle, None, 3, 64, 64, 3, 0, ml, 2, None, None, C.byref(n_out)) != 0
        assert "negative" in _lib.last_error()
    finally:
        eng.close()


@pytest.mark.parametrize("kw,mult,mlist,skip", [
    (dict(multiplier=2), 2, None, None),
    (dict(multiplier=3, optional_interpolation_states=InterpolationStateList([1], True)), 3, None, [0, 1, 0]),
    (dict(multiplier=[2, 0, 3]), 0, [2, 0, 3], None),
    (dict(multiplier=[1, 2, 0]), 0, [1, 2, 0], None),
    (dict(multiplier=[3], optional_interpolation_states=InterpolationStateList([0], False)), 0, [3], [0, 1, 1]),   # keep-list [0]: frame 0 not skipped
    (dict(multiplier=[3, 2], optional_interpolation_states=InterpolationStateList([0], True)), 0, [3, 2], [1, 0, 0]),  # local index 0 skipped: every pair
])
def test_m2m_run_equals_the_node(hip_lib, tmp_path, monkeypatch, kw, mult, mlist, skip):
    from cfi_amd import ckpt, m2m

    sd = synth.m2m_synth_state_dict(1234)
    pth = tmp_path / "M2M.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(m2m, "load_file_from_github_release", lambda model_type, ckpt_: str(pth))
    ckpt.clear_engine_cache()
    frames = synth.smooth_frames(4, 64, 64, seed=13, shift=2.0, c=4).contiguous()
    (want,) = m2m.M2M_VFI().vfi("M2M.pth", frames, **kw)
    ckpt.clear_engine_cache()
    eng = m2m.M2MEngine(sd)
    try:
        out = _call(hip_lib.vfi_m2m_run, eng.handle, frames, mult, mlist, skip)
    finally:
        eng.close()
    assert out.shape == want.shape and not torch.isnan(out).any()
    assert (out - want).abs().max().item() <= 2e-5, describe_diff(out, want, "vfi_m2m_run vs the node")
