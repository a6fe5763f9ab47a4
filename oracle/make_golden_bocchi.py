"""ORACLE tooling — real-image 1080p goldens: the REAL reference nodes run on demo_frames/bocchi0.jpg + bocchi1.jpg.

    python oracle/make_golden_bocchi.py [rife] [film] [m2m]

Runs only in the build container (needs /root/reference; ~10 min on 8 cores).  SURVEY.md §2 row 25 / §8d config 2(ii).
What is executed, per model, on the full 1080x1920 pair decoded by PIL (stored as uint8 in tests/golden/bocchi_pair_u8.npz,
frames = u8 / 255 as ComfyUI's LoadImage does):
  * RIFE 4.7: ``vfi_models.rife.RIFE_VFI.vfi("rife47.pth", frames, multiplier=2)`` and ``multiplier=4`` (t = .25/.5/.75),
    default synthetic checkpoint AND the "hot" one (synth.rife47_hot_state_dict: 40-70 px flows);
  * FILM: ``vfi_models.film.FILM_VFI.vfi("film_net_fp32.pt", frames, multiplier=2)`` — the node ``torch.jit.load``s its model;
    the artifact is absent offline, so a TorchScript TRACE of the in-tree source mirror ``film_arch.Interpolator`` with the
    synthetic weights is saved to a temp file and loaded by the unmodified node (default + hot weights);
  * M2M: ``vfi_models.m2m.M2M_VFI.vfi("M2M.pth", frames, multiplier=2)`` on the reference's OWN ops package (kernel text
    compiled for the host, oracle/ref_import.reference_ops) (default + hot weights).
Each result is stored as a fingerprint (oracle/golden_stats.py: 12 full-precision 128x128 crops + 8x8 block mean / max of
the whole frame) and the in-repo oracle is checked against the reference's frame on the spot (bit-exact expected) —
appended to oracle/VALIDATION_BOCCHI.log.
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import synth  # noqa: E402
from oracle import golden_stats, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LOG = os.path.join(ROOT, "oracle", "VALIDATION_BOCCHI.log")
_lines = []


def log(s):
    print(s, flush=True)
    _lines.append(s)


def bocchi_u8():
    from PIL import Image

    return np.stack([np.asarray(Image.open(os.path.join(ref_import.REFERENCE, "demo_frames", n)).convert("RGB"))
                     for n in ("bocchi0.jpg", "bocchi1.jpg")])


def save_fp(name, frames_by_key):
    """frames_by_key: {key: [H,W,3] float32 numpy} -> tests/golden/<name>.npz with '<key>/<field>' arrays"""
    arrs = {}
    for key, fr in frames_by_key.items():
        for f, v in golden_stats.fingerprint(fr).items():
            arrs[f"{key}/{f}"] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    log(f"  wrote tests/golden/{name}.npz ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e6:.2f} MB): {sorted(frames_by_key)}")


def do_rife(fr):
    from oracle import rife_oracle

    out = {}
    for tag, sd in (("default", synth.rife47_synth_state_dict(1234)), ("hot", synth.rife47_hot_state_dict(1234))):
        with tempfile.TemporaryDirectory() as td:
            pth = os.path.join(td, "rife47.pth")
            torch.save(sd, pth)
            R = ref_import.rife_node(pth)
            for m in (2, 4):
                R._model_cache.clear()
                t0 = time.time()
                o = R.RIFE_VFI().vfi("rife47.pth", fr, multiplier=m)[0]
                assert o.shape[0] == m + 1 and torch.equal(o[0], fr[0]) and torch.equal(o[-1], fr[1])
                want = rife_oracle.rife_vfi(sd, fr, multiplier=m)
                d = (want - o).abs().max().item()
                x = fr.permute(0, 3, 1, 2)
                _, aux = rife_oracle.ifnet47_forward(sd, x[0:1], x[1:2], torch.tensor([0.5]).view(1, 1, 1, 1), return_aux=True)
                fmax = [round(a[0].abs().max().item(), 1) for a in aux]
                log(f"RIFE 4.7 {tag} x{m}: reference node {time.time() - t0:.1f} s; oracle vs reference max|d| = {d:.3e}; "
                    f"max |flow| per block (px) {fmax}; mid-frame range [{o[1:-1].min().item():.3f}, {o[1:-1].max().item():.3f}]")
                assert d == 0.0
                for k in range(1, m):
                    if (tag, m, k) in (("default", 2, 1), ("hot", 2, 1), ("hot", 4, 1)):      # kept small: 3 frames are committed
                        out[f"{tag}_x{m}_{k}"] = o[k].numpy()
    save_fp("rife47_bocchi1080", out)


def do_film(fr):
    import warnings

    from oracle import film_oracle
    from oracle.validate_film_vs_reference import load_film_arch

    fa = load_film_arch()
    ref_import.setup()
    import vfi_models.film as FM

    out = {}
    x = fr.permute(0, 3, 1, 2).contiguous()
    for tag, sd in (("default", synth.film_synth_state_dict(1234)), ("hot", synth.film_hot_state_dict(1234))):
        net = fa.Interpolator()
        net.load_state_dict(sd, strict=True)
        net.eval()
        with tempfile.TemporaryDirectory() as td, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pt = os.path.join(td, "film_net_fp32.pt")
            t0 = time.time()
            with torch.inference_mode():
                ts = torch.jit.trace(net, (x[0:1], x[1:2], torch.full((1, 1), 0.5)), check_trace=False)
            ts.save(pt)
            log(f"FILM {tag}: TorchScript trace of film_arch.Interpolator saved ({os.path.getsize(pt) / 1e6:.0f} MB, {time.time() - t0:.0f} s)")
            FM.load_file_from_github_release = lambda model_type, ckpt: pt
            t0 = time.time()
            o = FM.FILM_VFI().vfi("film_net_fp32.pt", fr, multiplier=2)[0]
            t_node = time.time() - t0
        assert o.shape[0] == 3 and torch.equal(o[0], fr[0]) and torch.equal(o[2], fr[1])
        with torch.inference_mode():
            want, aux = film_oracle.film_forward(sd, x[0:1], x[1:2], return_aux=True)
        want = want.clamp(0, 1).permute(0, 2, 3, 1)[0]
        d = (want - o[1]).abs().max().item()
        fl = [round(f.abs().max().item(), 1) for f in aux["fwd_flow"]]
        log(f"FILM {tag} x2: reference node (unmodified, torch.jit.load of the trace) {t_node:.0f} s; oracle vs reference max|d| = {d:.3e}; "
            f"max |flow| per pyramid level fine->coarse (px) {fl}")
        # The eager module is bit-exact with the oracle (oracle/VALIDATION_FILM.log); the TorchScript executor runs the same graph
        # with its own fusions / conv algorithm choices, so the unmodified node differs from eager by fp32 rounding noise.
        assert d <= 1e-4
        out[f"{tag}_x2_1"] = o[1].numpy()
    save_fp("film_bocchi1080", out)


def do_m2m(fr):
    from oracle import m2m_model_oracle
    from oracle.validate_m2m_vs_reference import load_m2m_arch

    load_m2m_arch()        # the reference's M2M_arch bound to the reference's own (host-compiled) ops
    import vfi_models.m2m as M

    out = {}
    x = fr.permute(0, 3, 1, 2).contiguous()
    for tag, sd in (("default", synth.m2m_synth_state_dict(1234)), ("hot", synth.m2m_hot_state_dict(1234))):
        with tempfile.TemporaryDirectory() as td:
            pth = os.path.join(td, "M2M.pth")
            torch.save(sd, pth)
            M.load_file_from_github_release = lambda model_type, ckpt: pth
            for m in (2, 3):
                t0 = time.time()
                o = M.M2M_VFI().vfi("M2M.pth", fr, multiplier=m)[0]
                t_node = time.time() - t0
                assert o.shape[0] == m + 1 and torch.equal(o[0], fr[0]) and torch.equal(o[-1], fr[1])
                want = m2m_model_oracle.m2m_vfi(sd, fr, multiplier=m)
                d = (want - o).abs().max().item()
                with torch.inference_mode():
                    _, aux = m2m_model_oracle.m2m_forward(sd, x[0:1], x[1:2], [torch.tensor([0.5]).view(1, 1, 1, 1)], return_aux=True)
                tf = aux["ten_fwd"].abs()
                log(f"M2M {tag} x{m}: reference node on its own ops {t_node:.0f} s; oracle vs reference max|d| = {d:.3e}; PWC flow max "
                    f"{aux['fwd'].abs().max().item():.1f} px (1/2 res), refined multi-branch flows max {tf.max().item():.1f} / mean {tf.mean().item():.1f} px; "
                    f"frame range [{o[1:-1].min().item():.3f}, {o[1:-1].max().item():.3f}]")
                assert d == 0.0
                for k in range(1, m):
                    if (tag, m, k) in (("default", 2, 1), ("hot", 2, 1), ("hot", 3, 1)):
                        out[f"{tag}_x{m}_{k}"] = o[k].numpy()
    save_fp("m2m_bocchi1080", out)


def main():
    which = sys.argv[1:] or ["rife", "film", "m2m"]
    os.makedirs(OUT, exist_ok=True)
    u8 = bocchi_u8()
    assert u8.shape == (2, 1080, 1920, 3)
    np.savez_compressed(os.path.join(OUT, "bocchi_pair_u8.npz"), frames_u8=u8)
    log(f"== make_golden_bocchi {' '.join(which)} (torch {torch.__version__}, {torch.get_num_threads()} threads)")
    log(f"bocchi pair decoded: {u8.shape} uint8, tests/golden/bocchi_pair_u8.npz {os.path.getsize(os.path.join(OUT, 'bocchi_pair_u8.npz')) / 1e6:.2f} MB; "
        f"mean |frame0 - frame1| = {np.abs(u8[0].astype(np.int32) - u8[1]).mean() / 255:.4f}")
    fr = torch.from_numpy(u8.astype(np.float32) / 255.0)
    try:
        with torch.inference_mode():
            if "rife" in which:
                do_rife(fr)
            if "film" in which:
                do_film(fr)
            if "m2m" in which:
                do_m2m(fr)
        # the goldens are bit-exact pins only on a host whose torch-CPU kernels are the ones that ran here (golden_stats.host_signature)
        sig = golden_stats.write_host_signature(os.path.join(OUT, "bocchi1080_host.json"))
        log(f"host signature {sig}: {golden_stats.host_signature()[1]}")
    finally:
        with open(LOG, "a") as f:
            f.write("\n".join(_lines) + "\n")


if __name__ == "__main__":
    main()
