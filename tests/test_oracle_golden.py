"""Pin the oracle: oracle/rife_oracle.py must reproduce outputs of the REAL reference
(tests/golden/*.npz, written by oracle/make_golden.py from /root/reference) — CPU only."""
import os

import numpy as np
import pytest
import torch

from cfi_amd import synth
from cfi_amd.schedule import InterpolationStateList
from oracle import ref_import, rife_oracle

# torch-CPU results depend (at the 1-ulp level) on the host's SIMD width / oneDNN kernels, so
# "same machine" is bit-exact (oracle/VALIDATION.log) while across machines we allow 2e-5.
TOL = 2e-5


def test_warp_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "rife_warp.npz"))
    y = rife_oracle.warp(torch.from_numpy(g["x"]), torch.from_numpy(g["flow"]))
    assert np.abs(y.numpy() - g["y"]).max() <= 1e-6


def test_ifnet47_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "rife47_net_anime.npz"))
    sd = synth.rife47_synth_state_dict(1234)
    fr = torch.from_numpy(g["frames"])
    ts = torch.from_numpy(g["timesteps"]).view(-1, 1, 1, 1)
    b = ts.shape[0]
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    with torch.inference_mode():
        out = rife_oracle.ifnet47_forward(sd, i0, i1, ts).permute(0, 2, 3, 1)
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() <= TOL


CASES = {
    "m2": dict(multiplier=2),
    "m3_bs2": dict(multiplier=3, batch_size=2),
    "mlist": dict(multiplier=[3, 0, 1]),
    "m2_skip12": dict(multiplier=2, states=InterpolationStateList([1, 2], True)),
    "m2_keep12": dict(multiplier=2, states=InterpolationStateList([1, 2], False)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_node_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "rife47_node.npz"))
    sd = synth.rife47_synth_state_dict(1234)
    out = rife_oracle.rife_vfi(sd, torch.from_numpy(g["frames"]), **CASES[name])
    assert out.shape == g[name].shape
    assert np.abs(out.numpy() - g[name]).max() <= TOL


@pytest.mark.skipif(not ref_import.available(), reason="reference checkout not mounted")
def test_oracle_bit_exact_vs_live_reference():
    """Where /root/reference exists, run the real IFNet next to the oracle: must be identical."""
    ref = ref_import.rife_arch()
    sd = synth.rife47_synth_state_dict(7)
    net = ref.IFNet("4.7")
    net.load_state_dict(sd, strict=True)
    net.eval()
    fr = synth.smooth_frames(2, 70, 90, seed=9, shift=3.0)
    i0 = fr[0:1].permute(0, 3, 1, 2).contiguous()
    i1 = fr[1:2].permute(0, 3, 1, 2).contiguous()
    ts = torch.tensor([0.3]).view(1, 1, 1, 1)
    with torch.inference_mode():
        a = net(i0, i1, ts, [8, 4, 2, 1], False, False)
        b = rife_oracle.ifnet47_forward(sd, i0, i1, ts)
    assert torch.equal(a, b)


# ---- FILM / M2M: goldens written by oracle/make_golden_film_m2m.py from the reference's film_arch.Interpolator,
# M2M_arch.M2M_PWC and the real M2M_VFI node (the two cupy ops through their C restatement) ------------------------------

def test_film_interpolator_matches_reference_golden(golden_dir):
    from oracle import film_oracle

    g = np.load(os.path.join(golden_dir, "film_net.npz"))
    x = torch.from_numpy(g["frames"]).permute(0, 3, 1, 2)
    with torch.inference_mode():
        out = film_oracle.film_forward(synth.film_synth_state_dict(1234), x[0:1], x[1:2]).permute(0, 2, 3, 1)
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() <= TOL * max(1.0, np.abs(g["out"]).max())


def test_m2m_model_matches_reference_golden(golden_dir):
    from oracle import m2m_model_oracle as mo

    g = np.load(os.path.join(golden_dir, "m2m_net.npz"))
    x = torch.from_numpy(g["frames"]).permute(0, 3, 1, 2)
    ts = [torch.tensor([float(t)]).view(1, 1, 1, 1) for t in g["times"]]
    with torch.inference_mode():
        outs = mo.m2m_forward(synth.m2m_synth_state_dict(1234), x[0:1], x[1:2], ts)
    out = torch.cat(outs, 0).permute(0, 2, 3, 1)
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() <= 5 * TOL


M2M_NODE_CASES = {
    "m2": dict(multiplier=2),
    "m3_skip1": dict(multiplier=3, states=InterpolationStateList([1], True)),
    "mlist_203": dict(multiplier=[2, 0, 3]),
    "mlist_120": dict(multiplier=[1, 2, 0]),
    "mlist_3_keep0": dict(multiplier=[3], states=InterpolationStateList([0], False)),
}


@pytest.mark.parametrize("name", list(M2M_NODE_CASES))
def test_m2m_node_matches_reference_golden(golden_dir, name):
    """The oracle's restatement of generic_frame_loop (incl. the m == 0 and local-skip-index quirks) vs the real node"""
    from oracle import m2m_model_oracle as mo

    g = np.load(os.path.join(golden_dir, "m2m_node.npz"))
    out = mo.m2m_vfi(synth.m2m_synth_state_dict(1234), torch.from_numpy(g["frames"]), **M2M_NODE_CASES[name])
    assert out.shape == g[name].shape
    assert np.abs(out.numpy() - g[name]).max() <= 5 * TOL


@pytest.mark.parametrize("name", list(M2M_NODE_CASES))
def test_m2m_plan_matches_reference_golden(golden_dir, name):
    """Host logic of the product node: the output plan must reproduce the reference's frame count and pass-through slots"""
    from cfi_amd.schedule import generic_output_plan

    g = np.load(os.path.join(golden_dir, "m2m_node.npz"))
    kw = M2M_NODE_CASES[name]
    plan, tasks = generic_output_plan(len(g["frames"]), kw["multiplier"], kw.get("states"))
    assert len(plan) == len(g[name])
    for i, (kind, idx) in enumerate(plan):
        if kind == "src":
            assert np.array_equal(g[name][i], g["frames"][idx][..., :3])


def test_config0_anime_pair_matches_reference_node(golden_dir):
    """BASELINE.json configs[0] on the CPU: oracle node loop vs the reference node's output on the full demo pair"""
    g = np.load(os.path.join(golden_dir, "rife47_node_anime540.npz"))
    frames = torch.from_numpy(g["frames_u8"].astype(np.float32) / 255.0)
    out = rife_oracle.rife_vfi(synth.rife47_synth_state_dict(1234), frames, multiplier=2)
    assert out.shape == (3, 540, 960, 3) and torch.equal(out[0], frames[0]) and torch.equal(out[2], frames[1])
    assert np.abs(out[1].numpy() - g["mid"]).max() <= TOL


# ---- arch 4.17 (rife417.pth): goldens written by oracle/validate_rife417_vs_reference.py from the reference's IFNet("4.17")

def test_ifnet417_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "rife417_net_anime.npz"))
    sd = synth.rife417_synth_state_dict(1234)
    fr = torch.from_numpy(g["frames"])
    ts = torch.from_numpy(g["timesteps"]).view(-1, 1, 1, 1)
    b = ts.shape[0]
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    with torch.inference_mode():
        out = rife_oracle.ifnet47_forward(sd, i0, i1, ts, arch="4.17").permute(0, 2, 3, 1)
    assert out.shape == g["out"].shape and np.abs(out.numpy() - g["out"]).max() <= TOL


@pytest.mark.parametrize("name,kw", [("m2", dict(multiplier=2)), ("mlist_bs2", dict(multiplier=[3, 1], batch_size=2))])
def test_node417_matches_reference_golden(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, "rife417_node.npz"))
    out = rife_oracle.rife_vfi(synth.rife417_synth_state_dict(1234), torch.from_numpy(g["frames"]), arch="4.17", **kw)
    assert out.shape == g[name].shape and np.abs(out.numpy() - g[name]).max() <= TOL


# ---- arch 4.26 (rife426.pth): goldens written by oracle/validate_rife426_vs_reference.py from the reference's IFNet("4.26")

def test_ifnet426_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "rife426_net_anime.npz"))
    sd = synth.rife426_synth_state_dict(1234)
    fr = torch.from_numpy(g["frames"])
    ts = torch.from_numpy(g["timesteps"]).view(-1, 1, 1, 1)
    b = ts.shape[0]
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    with torch.inference_mode():
        out = rife_oracle.ifnet47_forward(sd, i0, i1, ts, (16, 8, 4, 2, 1), arch="4.26").permute(0, 2, 3, 1)
    assert out.shape == g["out"].shape and np.abs(out.numpy() - g["out"]).max() <= TOL


@pytest.mark.parametrize("name,kw", [("m2", dict(multiplier=2)), ("mlist_bs2", dict(multiplier=[3, 1], batch_size=2))])
def test_node426_matches_reference_golden(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, "rife426_node.npz"))
    out = rife_oracle.rife_vfi(synth.rife426_synth_state_dict(1234), torch.from_numpy(g["frames"]), arch="4.26", **kw)
    assert out.shape == g[name].shape and np.abs(out.numpy() - g[name]).max() <= TOL


# ---- arch 4.0 (sudo_rife4 checkpoint): goldens written by oracle/validate_rife40_vs_reference.py ----------------------------

@pytest.mark.parametrize("key,training,fastmode", [("out_fast", True, True), ("out_full", False, False)])
def test_ifnet40_matches_reference_golden(golden_dir, key, training, fastmode):
    g = np.load(os.path.join(golden_dir, "rife40_net_anime.npz"))
    sd = synth.rife40_synth_state_dict(1234)
    fr = torch.from_numpy(g["frames"])
    ts = torch.from_numpy(g["timesteps"]).view(-1, 1, 1, 1)
    b = ts.shape[0]
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(b, 1, 1, 1)
    with torch.inference_mode():
        out = rife_oracle.ifnet40_forward(sd, i0, i1, ts, [8.0, 4.0, 2.0, 1.0], training, fastmode).permute(0, 2, 3, 1)
    assert out.shape == g[key].shape and np.abs(out.numpy() - g[key]).max() <= TOL


@pytest.mark.parametrize("name,kw", [("default", dict(multiplier=2, fast_mode=True, ensemble=True)),
                                     ("refine_bs2", dict(multiplier=[3, 1], batch_size=2, fast_mode=False, ensemble=False))])
def test_node40_matches_reference_golden(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, "rife40_node.npz"))
    out = rife_oracle.rife_vfi(synth.rife40_synth_state_dict(1234), torch.from_numpy(g["frames"]), arch="4.0", **kw)
    assert out.shape == g[name].shape and np.abs(out.numpy() - g[name]).max() <= TOL


# ---- IFRNet: goldens written by oracle/validate_ifrnet_vs_reference.py from the reference's IFRNet_VFI node ----------
IFRNET_NODE_CASES = {
    "x2": dict(multiplier=2),
    "x2_t05": dict(multiplier=2, scale_factor=0.5),
    "x4_skip1": dict(multiplier=4, states=InterpolationStateList([1], True)),
}


@pytest.mark.parametrize("kind", ["L", "S"])
@pytest.mark.parametrize("name", list(IFRNET_NODE_CASES))
def test_ifrnet_node_matches_reference_golden(golden_dir, kind, name):
    """oracle/ifrnet_oracle.py (incl. the node's timestep/scale_factor mis-binding: multiplier 4 runs the network at
    working resolutions 0.25, 0.5 and 0.75) vs outputs of the reference node"""
    from oracle import ifrnet_oracle

    g = np.load(os.path.join(golden_dir, "ifrnet_node.npz"))
    out = ifrnet_oracle.ifrnet_vfi(synth.ifrnet_synth_state_dict(kind, 1234), torch.from_numpy(g[f"{kind}_frames"]),
                                   **IFRNET_NODE_CASES[name])
    want = g[f"{kind}_{name}"]
    assert out.shape == want.shape
    assert np.abs(out.numpy() - want).max() <= 5 * TOL


def test_ifrnet_spec_and_geometry():
    from cfi_amd import ifrnet_spec

    for kind, n_tensors in (("L", 104), ("S", 104)):
        sh = ifrnet_spec.ifrnet_shapes(kind)
        assert len(sh) == n_tensors
        ifrnet_spec.check_state_dict(synth.ifrnet_synth_state_dict(kind, 1), kind)
    assert ifrnet_spec.decoder_io("L") == [(4, 385, 384, 148), (3, 436, 432, 100), (2, 292, 288, 68), (1, 196, 192, 8)]
    assert ifrnet_spec.decoder_io("S") == [(4, 145, 144, 58), (3, 166, 162, 40), (2, 112, 108, 28), (1, 76, 72, 8)]
    assert [ifrnet_spec.kind_of(n) for n in ifrnet_spec.CKPT_NAMES] == ["S", "L", "S", "L"]
    from cfi_amd.ifrnet import IFRNetEngine

    assert IFRNetEngine.geometry(72, 100, 0.75) == (128, 128, 96, 96, 128, 128)
    assert IFRNetEngine.geometry(64, 64, 1.0 / 3.0) == (64, 64, 21, 21, 63, 63)      # the reference fails in torch.cat here
    assert IFRNetEngine.geometry(1080, 1920, 0.5) == (1088, 1920, 544, 960, 1088, 1920)
    bad = synth.ifrnet_synth_state_dict("S", 1)
    bad.pop("decoder1.convblock.2.bias")
    with pytest.raises(KeyError):
        ifrnet_spec.check_state_dict(bad, "S")


# ---- GMFSS Fortuna (union), SURVEY 8f rank 3 groundwork: golden written by oracle/validate_gmfss_vs_reference.py from
# the reference's CommonModelInference (summation splat through the C restatement, as for M2M) -------------------------
def test_gmfss_union_matches_reference_golden(golden_dir):
    from oracle import gmfss_oracle

    g = np.load(os.path.join(golden_dir, "gmfss_union.npz"))
    x = torch.from_numpy(g["frames"]).permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        out = gmfss_oracle.gmfss_forward(synth.gmfss_synth_state_dicts(1234), x[0:1], x[1:2], float(g["t"])).permute(0, 2, 3, 1)
        base = gmfss_oracle.gmfss_forward(synth.gmfss_synth_state_dicts(1234, "base"), x[0:1], x[1:2], float(g["t"])).permute(0, 2, 3, 1)
    assert out.shape == g["out"].shape
    # the matching softmax over ~200 random-feature candidates amplifies last-bit differences of other CPUs' matmul
    # kernels; on the build container the agreement is bit-exact (oracle/VALIDATION_GMFSS.log)
    assert np.abs(out.numpy() - g["out"]).mean() <= 1e-4
    assert np.abs(base.numpy() - g["base_out"]).mean() <= 1e-4


def test_gmfss_spec_tables():
    from cfi_amd import gmfss_spec

    sh = gmfss_spec.gmfss_union_shapes()
    assert [len(sh[p]) for p in gmfss_spec.PARTS] == [120, 124, 14, 18, 133]
    base = gmfss_spec.gmfss_base_shapes()
    assert "ifnet" not in base and base["fusionnet"]["residual_model_head.1.weight"] == (64, 12, 3, 3)
    sds = synth.gmfss_synth_state_dicts(3)
    assert all(tuple(sds[p][k].shape) == tuple(v) for p in gmfss_spec.PARTS for k, v in sh[p].items())


# ---- IFUNet (second half): goldens written by oracle/validate_ifunet_vs_reference.py from the reference node --
IFUNET_NODE_CASES = {
    "x2": dict(multiplier=2),
    "x2_noens_s05": dict(multiplier=2, scale_factor=0.5, ensemble=False),
    "x3_skip0": dict(multiplier=3, states=InterpolationStateList([0], True)),
}


@pytest.mark.parametrize("name", list(IFUNET_NODE_CASES))
def test_ifunet_node_matches_reference_golden(golden_dir, name):
    from oracle import ifunet_oracle

    g = np.load(os.path.join(golden_dir, "ifunet_node.npz"))
    out = ifunet_oracle.ifunet_vfi(synth.ifunet_synth_state_dict(1234), torch.from_numpy(g["frames"]), **IFUNET_NODE_CASES[name])
    assert out.shape == g[name].shape
    assert np.abs(out.numpy() - g[name]).max() <= 5 * TOL


def test_ifunet_spec_table():
    from cfi_amd import ifunet_spec

    sh = ifunet_spec.ifunet_shapes()
    assert len(sh) == 608 and sh["flownet.block0.maskconvx16.weight"] == (2304, 256, 1, 1) and sh["refinenet.block1.conv0.0.0.weight"] == (64, 12, 3, 3)
    sd = synth.ifunet_synth_state_dict(2)
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in sh.items())


# ---- real-image 1080p goldens (oracle/make_golden_bocchi.py): the oracle against the REAL reference node's fingerprints ------
def _bocchi(golden_dir):
    import numpy as np

    u8 = np.load(os.path.join(golden_dir, "bocchi_pair_u8.npz"))["frames_u8"]
    return torch.from_numpy(u8.astype(np.float32) / 255.0)


@pytest.mark.parametrize("tag,m,k", [("default", 2, 1), ("hot", 4, 1)])
def test_rife_oracle_vs_reference_node_bocchi_1080p(golden_dir, tag, m, k):
    import numpy as np

    from oracle import golden_stats, rife_oracle

    sd = synth.rife47_synth_state_dict(1234) if tag == "default" else synth.rife47_hot_state_dict(1234)
    out = rife_oracle.rife_vfi(sd, _bocchi(golden_dir), multiplier=m)
    npz = np.load(os.path.join(golden_dir, "rife47_bocchi1080.npz"))
    fp = {f: npz[f"{tag}_x{m}_{k}/{f}"] for f in ("crops", "crop_pos", "pool_mean", "pool_max")}
    # bit-exact on the host that wrote the golden (signature beside it), cross-host spread of torch-CPU elsewhere (golden_stats.golden_tol)
    tol = golden_stats.golden_tol(os.path.join(golden_dir, "bocchi1080_host.json"))
    d = golden_stats.check(out[k].numpy(), fp, tol=tol, name=f"oracle vs reference node, RIFE 4.7 {tag} x{m}")
    assert d[0] <= tol and d[2] <= tol


def test_m2m_oracle_vs_reference_node_bocchi_1080p(golden_dir):
    import numpy as np

    from oracle import golden_stats, m2m_model_oracle

    out = m2m_model_oracle.m2m_vfi(synth.m2m_synth_state_dict(1234), _bocchi(golden_dir), multiplier=2)
    npz = np.load(os.path.join(golden_dir, "m2m_bocchi1080.npz"))
    fp = {f: npz[f"default_x2_1/{f}"] for f in ("crops", "crop_pos", "pool_mean", "pool_max")}
    tol = golden_stats.golden_tol(os.path.join(golden_dir, "bocchi1080_host.json"))
    d = golden_stats.check(out[1].numpy(), fp, tol=tol, name="oracle vs reference node, M2M default x2")
    assert d[0] <= tol and d[2] <= tol
