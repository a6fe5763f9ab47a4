"""FILM's checkpoint is a TorchScript file (``torch.jit.load(model_path)``, film/__init__.py:74); the real ``film_net_fp32.pt`` is
absent offline.  What CAN be pinned here: a TorchScript module saved from the reference's own source mirror
``film_arch.Interpolator`` (the module the artifact was exported from) goes through this package's loader
(``film._load_state_dict`` -> ``film_spec.check_state_dict``) with every key, shape and value intact — the path a real artifact
takes, which the GPU tests (plain ``torch.save`` state dicts) never exercise.  Needs /root/reference (build container only)."""
import os
import warnings

import pytest
import torch

from cfi_amd import film_spec, synth


@pytest.mark.skipif(not os.path.isdir("/root/reference/vfi_models/film"), reason="needs the reference checkout (build container)")
def test_torchscript_artifact_goes_through_the_loader(tmp_path):
    from cfi_amd import film
    from oracle.validate_film_vs_reference import load_film_arch

    fa = load_film_arch()
    sd = synth.film_synth_state_dict(1234)
    net = fa.Interpolator()
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.rand(1, 3, 64, 64)
    with warnings.catch_warnings(), torch.inference_mode():
        warnings.simplefilter("ignore")
        ts = torch.jit.trace(net, (x, x.flip(3), torch.full((1, 1), 0.5)), check_trace=False)
    pt = tmp_path / "film_net_fp32.pt"
    ts.save(str(pt))
    got = film._load_state_dict(str(pt))
    film_spec.check_state_dict(got)                       # = load_state_dict(strict=True): no missing / unexpected key, shapes equal
    assert list(got.keys()) == list(film_spec.film_shapes().keys())
    for k, v in sd.items():
        assert torch.equal(got[k], v), k


def test_plain_state_dict_still_loads(tmp_path):
    from cfi_amd import film

    sd = synth.film_synth_state_dict(7)
    p = tmp_path / "film.pt"
    torch.save(sd, p)
    got = film._load_state_dict(str(p))
    film_spec.check_state_dict(got)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
