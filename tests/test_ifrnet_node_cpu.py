"""IFRNet node on CPU: the real IFRNet_VFI.vfi (comfyui-frame-interpolation_amd/ifrnet.py) with the device engine swapped
for a stand-in that calls the oracle, against the oracle's restatement of the reference node (ifrnet/__init__.py:32-57 +
generic_frame_loop) — checks the host logic: plan, pass-through frames, and that the loop's timestep reaches the network
as its working-resolution factor while the ``scale_factor`` widget becomes the time embedding."""
import pytest
import torch

from cfi_amd import synth
from cfi_amd.schedule import InterpolationStateList


class OracleEngine:
    """prepare/render interface of IFRNetEngine on the CPU, backed by the oracle (test infrastructure only)."""

    def __init__(self, sd):
        self.sd, self.device, self.embt, self.calls = sd, torch.device("cpu"), None, []

    def prepare(self, f0, f1):
        self.pair = (f0.permute(2, 0, 1)[None], f1.permute(2, 0, 1)[None])

    def render(self, t, out=None):
        from oracle import ifrnet_oracle

        self.calls.append((t, self.embt))
        with torch.inference_mode():
            y = ifrnet_oracle.ifrnet_forward(self.sd, self.pair[0], self.pair[1], t, self.embt)[0].permute(1, 2, 0)
        out.copy_(y)
        return out

    def release_workspace(self):
        pass

    def close(self):
        pass


@pytest.mark.parametrize("kw", [dict(multiplier=2), dict(multiplier=2, scale_factor=0.5),
                                dict(multiplier=4, optional_interpolation_states=InterpolationStateList([0], True))])
def test_node_host_logic_matches_oracle_node(tmp_path, monkeypatch, kw):
    import cfi_amd.ifrnet as I
    from oracle import ifrnet_oracle

    sd = synth.ifrnet_synth_state_dict("S", 7)
    pth = tmp_path / "IFRNet_S_Vimeo90K.pth"
    torch.save(sd, pth)
    eng = OracleEngine(sd)
    monkeypatch.setattr(I, "load_file_from_github_release", lambda model_type, ckpt_name: str(pth))
    monkeypatch.setattr(I, "cached_engine", lambda model_type, path, build: (eng, True))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    frames = synth.smooth_frames(3, 64, 64, seed=4, shift=2.0, c=4)      # RGBA clip: alpha is dropped
    (out,) = I.IFRNet_VFI().vfi(pth.name, frames, clear_cache_after_n_frames=10, **kw)
    okw = dict(kw)
    states = okw.pop("optional_interpolation_states", None)
    want = ifrnet_oracle.ifrnet_vfi(sd, frames, states=states, **okw)
    assert out.shape == want.shape and out.dtype == torch.float32
    assert (out - want).abs().max().item() <= 2e-5
    m = kw["multiplier"]
    n_pairs = 2 - (1 if states is not None else 0)
    assert eng.calls == [(k / m, float(kw.get("scale_factor", 1.0))) for _ in range(n_pairs) for k in range(1, m)]


def test_node_input_types_match_reference_surface():
    import cfi_amd.ifrnet as I

    it = I.IFRNet_VFI.INPUT_TYPES()
    assert list(it["required"]) == ["ckpt_name", "frames", "clear_cache_after_n_frames", "multiplier", "scale_factor"]
    assert it["required"]["ckpt_name"][0] == ["IFRNet_S_Vimeo90K.pth", "IFRNet_L_Vimeo90K.pth", "IFRNet_S_GoPro.pth", "IFRNet_L_GoPro.pth"]
    assert it["required"]["scale_factor"][0] == [0.25, 0.5, 1.0, 2.0, 4.0] and "optional_interpolation_states" in it["optional"]
    assert I.IFRNet_VFI.RETURN_TYPES == ("IMAGE",) and I.IFRNet_VFI.FUNCTION == "vfi"
    with pytest.raises(AssertionError):
        I.IFRNet_VFI().vfi("IFRNet_S_Vimeo90K.pth", torch.zeros(1, 8, 8, 3))
