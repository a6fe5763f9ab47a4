"""ORACLE tooling — pin oracle/gmfss_oracle.py against the reference's own GMFSS Fortuna (union) modules and the
GMFSS_Fortuna_VFI node, here, on CPU, with seeded synthetic checkpoints; write tests/golden/gmfss_union.npz (outputs of
the REFERENCE).  The reference imports ``vfi_models.ops.softsplat``: it is the reference's own CuPy op package running on the host
(oracle/stubs/cupy compiles the kernel text the reference specialises with g++ behind a serial shim, see
oracle/ref_import.reference_ops), so the reference side runs its own splat kernel and wrapper; the oracle side runs
oracle/m2m_ops.c.
Bit-exact agreement is required.  Writes oracle/VALIDATION_GMFSS.log."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import gmfss_spec, synth  # noqa: E402
from oracle import gmfss_oracle as G, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    ref_import.reference_ops()     # the reference's own vfi_models.ops (cupy_ops) on the host shim, oracle/stubs/cupy
    import vfi_models.gmfss_fortuna as N
    from vfi_models.gmfss_fortuna import GMFSS_Fortuna_union_arch as A

    sds = synth.gmfss_synth_state_dicts(1234)
    model = A.Model()
    model.eval()
    nets = {"ifnet": model.ifnet, "flownet": model.flownet, "metricnet": model.metricnet, "feat_ext": model.feat_ext,
            "fusionnet": model.fusionnet}
    shapes = gmfss_spec.gmfss_union_shapes()
    for part, net in nets.items():
        ref_keys = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert list(ref_keys) == list(shapes[part]), f"{part}: key order differs"
        assert ref_keys == {k: tuple(v) for k, v in shapes[part].items()}, f"{part}: shapes differ"
        net.load_state_dict(sds[part], strict=True)
        log(f"{part}: reference module loaded the synthetic state_dict strictly: {len(sds[part])} tensors, "
            f"{sum(v.numel() for v in sds[part].values())} params")
    ok = True
    golden = {}
    with torch.inference_mode():
        # ---- sub-networks
        fr = synth.smooth_frames(2, 128, 192, seed=5, shift=3.0)
        x = fr.permute(0, 3, 1, 2).contiguous()
        i0, i1 = x[0:1], x[1:2]
        for (h, w) in ((64, 96), (128, 128)):
            a, b = i0[:, :, :h, :w].contiguous(), i1[:, :, :h, :w].contiguous()
            r = model.flownet(a, b, return_flow=True)
            o = G.gmflow(sds["flownet"], a, b)
            d = (r - o).abs().max().item()
            log(f"GMFlow {h}x{w}: max|ref-oracle| = {d:.3e}   max|flow| = {r.abs().max().item():.2f} px")
            ok &= d == 0.0
        f01 = model.flownet(i0, i1, return_flow=True)
        f10 = model.flownet(i1, i0, return_flow=True)
        r0, r1 = model.metricnet(i0, i1, f01, f10)
        o0, o1 = G.metricnet(sds["metricnet"], i0, i1, f01, f10)
        d = max((r0 - o0).abs().max().item(), (r1 - o1).abs().max().item())
        log(f"MetricNet 128x192: max|ref-oracle| = {d:.3e}   metric range [{r0.min().item():.2f}, {r0.max().item():.2f}]")
        ok &= d == 0.0
        rf, of = model.feat_ext(i0), G.featurenet(sds["feat_ext"], i0)
        d = max((a - b).abs().max().item() for a, b in zip(rf, of))
        log(f"FeatureNet 128x192: max|ref-oracle| = {d:.3e}")
        ok &= d == 0.0
        hh = torch.nn.functional.interpolate(i0, scale_factor=0.5, mode="bilinear", align_corners=False)
        h1 = torch.nn.functional.interpolate(i1, scale_factor=0.5, mode="bilinear", align_corners=False)
        r = model.ifnet(hh, h1, 0.3, scale_list=[8, 4, 2, 1])
        o = G.ifnet46_forward(sds["ifnet"], hh, h1, 0.3)
        d = (r - o).abs().max().item()
        log(f"IFNet 4.6 64x96 t=0.3: max|ref-oracle| = {d:.3e}")
        ok &= d == 0.0
        torch.manual_seed(0)
        gx = [torch.randn(1, c, 64 >> k, 96 >> k) * 0.5 for k, c in ((0, 9), (0, 128), (1, 256), (2, 384))]
        r, o = model.fusionnet(*gx), G.gridnet(sds["fusionnet"], *gx)
        d = (r - o).abs().max().item()
        log(f"GridNet 64x96: max|ref-oracle| = {d:.3e}")
        ok &= d == 0.0
        # ---- the whole model, through CommonModelInference.forward
        cm = N.CommonModelInference.__new__(N.CommonModelInference)
        torch.nn.Module.__init__(cm)
        cm.model = model
        for (h, w, t) in ((100, 150, 0.5), (128, 192, 0.25), (64, 64, 0.75)):
            fr = synth.smooth_frames(2, h, w, seed=h, shift=2.5)
            x = fr.permute(0, 3, 1, 2).contiguous()
            r = cm(x[0:1], x[1:2], t, 1)
            o = G.gmfss_forward(sds, x[0:1], x[1:2], t)
            d = (r - o).abs().max().item()
            log(f"GMFSS union forward {h}x{w} t={t}: max|ref-oracle| = {d:.3e}   out range [{r.min().item():.3f}, {r.max().item():.3f}] "
                f"std {r.std().item():.3f}")
            ok &= d == 0.0
            if (h, w) == (100, 150):
                golden["frames"], golden["t"], golden["out"] = fr.numpy(), np.float32(t), r.permute(0, 2, 3, 1).contiguous().numpy()
        # ---- the base model ("GMFSS_fortuna": no IFNet, 12-channel GridNet head), GMFSS_Fortuna_arch.py
        from vfi_models.gmfss_fortuna import GMFSS_Fortuna_arch as B

        bsds = synth.gmfss_synth_state_dicts(1234, "base")
        bmodel = B.Model()
        bmodel.eval()
        bnets = {"flownet": bmodel.flownet, "metricnet": bmodel.metricnet, "feat_ext": bmodel.feat_ext, "fusionnet": bmodel.fusionnet}
        bshapes = gmfss_spec.gmfss_base_shapes()
        for part, net in bnets.items():
            assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in bshapes[part].items()}, part
            assert list(net.state_dict()) == list(bshapes[part]), f"{part}: key order differs"
            net.load_state_dict(bsds[part], strict=True)
        cmb = N.CommonModelInference.__new__(N.CommonModelInference)
        torch.nn.Module.__init__(cmb)
        cmb.model = bmodel
        for (h, w, t) in ((100, 150, 0.5), (64, 64, 0.3)):
            fr = synth.smooth_frames(2, h, w, seed=h, shift=2.5)
            x = fr.permute(0, 3, 1, 2).contiguous()
            r = cmb(x[0:1], x[1:2], t, 1)
            o = G.gmfss_forward(bsds, x[0:1], x[1:2], t)
            d = (r - o).abs().max().item()
            log(f"GMFSS base forward {h}x{w} t={t}: max|ref-oracle| = {d:.3e}   out std {r.std().item():.3f}")
            ok &= d == 0.0
            if (h, w) == (100, 150):
                golden["base_out"] = r.permute(0, 2, 3, 1).contiguous().numpy()
    log("RESULT: " + ("oracle == reference, bit-exact on every case (the reference side ran its own softsplat kernel text, host-compiled; the oracle side oracle/m2m_ops.c)" if ok
                      else "MISMATCH"))
    np.savez_compressed(os.path.join(OUT, "gmfss_union.npz"), **golden)
    log(f"wrote tests/golden/gmfss_union.npz ({os.path.getsize(os.path.join(OUT, 'gmfss_union.npz')) / 1e6:.2f} MB)")
    with open(os.path.join(ROOT, "oracle", "VALIDATION_GMFSS.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
