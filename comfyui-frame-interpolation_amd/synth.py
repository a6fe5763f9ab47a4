"""Deterministic synthetic checkpoints and frames (no network, no checkpoints on disk).

Used by tests, ``bench.py`` and ``__graft_entry__.smoke()``.  Every tensor is
drawn from its own generator seeded by ``(seed, crc32(key))`` so the values do
not depend on module construction order: the same dict can be produced on the
GPU box (no reference there) and loaded into the reference ``IFNet`` here
(``load_state_dict(strict=True)``) when golden vectors are generated.

Magnitudes mirror torch's default Conv2d / ConvTranspose2d init
(U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fan_in taken over dims 1.. of the weight),
which the survey measured to give |flow| of a few pixels per stage — enough to
exercise the warp path.
"""
import zlib
import math
import torch

from .rife_spec import rife47_shapes


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
    return g


def synth_state_dict(shapes, seed=1234):
    sd = {}
    last_fan_in = 1
    for k, shp in shapes.items():
        g = _gen(seed, k)
        if k.endswith("beta"):
            t = 1.0 + 0.25 * (torch.rand(shp, generator=g) * 2 - 1)
        elif k.endswith("weight"):
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            last_fan_in = fan_in
            b = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shp, generator=g) * 2 - 1) * b
        else:  # bias
            b = 1.0 / math.sqrt(last_fan_in)
            t = (torch.rand(shp, generator=g) * 2 - 1) * b
        sd[k] = t.to(torch.float32).contiguous()
    return sd


def rife47_synth_state_dict(seed=1234):
    return synth_state_dict(rife47_shapes(), seed)


def rife417_synth_state_dict(seed=1234):
    from .rife_spec import rife417_shapes

    return synth_state_dict(rife417_shapes(), seed)


def rife426_synth_state_dict(seed=1234):
    from .rife_spec import rife426_shapes

    return synth_state_dict(rife426_shapes(), seed)


def rife40_synth_state_dict(seed=1234):
    """arch 4.0: conv weights / biases as torch's default init, PReLU slopes U(0.1, 0.4)"""
    from .rife_spec import rife40_shapes

    shapes = rife40_shapes()
    sd = synth_state_dict({k: v for k, v in shapes.items() if len(v) != 1 or k.endswith("bias")}, seed)
    out = {}
    for k, shp in shapes.items():
        if k in sd:
            out[k] = sd[k]
        else:   # PReLU slope vector
            out[k] = (0.1 + 0.3 * torch.rand(shp, generator=_gen(seed, k))).to(torch.float32)
    return out


def film_synth_state_dict(seed=1234, gain=1.2):
    """FILM: torch-default init shrinks the signal to a bias-dominated constant through ~20 LeakyReLU convs, which
    would make parity tests insensitive; weights are U(+-gain*sqrt(3/fan_in)) (roughly variance preserving), biases
    U(+-0.1): features O(1), finest-level flows of ~15 px, outputs O(1)."""
    from .film_spec import film_shapes

    sd = {}
    for k, shp in film_shapes().items():
        g = _gen(seed, k)
        if k.endswith("weight"):
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (torch.rand(shp, generator=g) * 2 - 1) * (gain * math.sqrt(3.0 / fan_in))
        else:
            t = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        sd[k] = t.to(torch.float32).contiguous()
    return sd


def m2m_synth_state_dict(seed=1234, gain=1.0):
    """M2M: conv weights U(+-gain*sqrt(3/fan_in)) (fan_in over dims 1.. of the weight, so ConvTranspose2d follows
    torch's convention), biases U(+-0.05), PReLU slopes U(0.1,0.4), paramAlpha 10 (its trained-from value)."""
    from .m2m_spec import m2m_shapes

    sd = {}
    for k, shp in m2m_shapes().items():
        g = _gen(seed, k)
        if k == "paramAlpha":
            t = torch.full(shp, 10.0)
        elif len(shp) == 4:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (torch.rand(shp, generator=g) * 2 - 1) * (gain * math.sqrt(3.0 / fan_in))
        elif k.endswith("bias"):
            t = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        else:  # PReLU slope(s)
            t = 0.1 + 0.3 * torch.rand(shp, generator=g)
        sd[k] = t.to(torch.float32).contiguous()
    return sd


def ifrnet_synth_state_dict(kind="L", seed=1234, gain=1.0):
    """IFRNet_L / IFRNet_S: conv weights U(+-gain*sqrt(3/fan_in)) (roughly variance preserving through the PReLU
    trunk, so the decoders emit flows of a few pixels), biases U(+-0.05), PReLU slopes U(0.1, 0.4)."""
    from .ifrnet_spec import ifrnet_shapes

    sd = {}
    for k, shp in ifrnet_shapes(kind).items():
        g = _gen(seed, kind + k)
        if len(shp) == 4:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (torch.rand(shp, generator=g) * 2 - 1) * (gain * math.sqrt(3.0 / fan_in))
        elif k.endswith("bias"):
            t = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        else:  # PReLU slopes
            t = 0.1 + 0.3 * torch.rand(shp, generator=g)
        sd[k] = t.to(torch.float32).contiguous()
    return sd


def gmfss_synth_state_dicts(seed=1234, variant="union"):
    """GMFSS Fortuna: the five ("union") or four ("base") state_dicts.  Conv / linear weights U(+-sqrt(3/fan_in)), biases U(+-0.05), PReLU
    slopes U(0.1, 0.4), LayerNorm weights 1 +- 0.1, ResConv betas 1 +- 0.25 (as rife47)."""
    from .gmfss_spec import gmfss_shapes

    out = {}
    for part, shapes in gmfss_shapes(variant).items():
        sd = {}
        for k, shp in shapes.items():
            g = _gen(seed, part + "/" + k)
            if k.endswith("beta"):
                t = 1.0 + 0.25 * (torch.rand(shp, generator=g) * 2 - 1)
            elif len(shp) >= 2:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                t = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in)
            elif "norm" in k and k.endswith("weight"):
                t = 1.0 + 0.1 * (torch.rand(shp, generator=g) * 2 - 1)
            elif k.endswith("bias"):
                t = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
            else:  # PReLU slope
                t = 0.1 + 0.3 * torch.rand(shp, generator=g)
            sd[k] = t.to(torch.float32).contiguous()
        out[part] = sd
    return out


def gmfss_coherent_state_dicts(seed=1234, variant="union", gain=4.0, ln=0.05):
    """GMFSS Fortuna checkpoints whose GMFlow MATCHES COHERENTLY on textured inputs (``texture_frames``): the test vector for
    the end-to-end 1e-3 gate.  With plain random weights (``gmfss_synth_state_dicts``) GMFlow's matching softmax is spread
    over unrelated pixels, flows are tens of pixels with no spatial structure, the soft splat divides by near-zero sums in
    the holes and rounding differences are amplified a thousand-fold — the reference model itself is ill-conditioned there.
    Changes to the flow network only (same keys, same shapes, still a valid checkpoint for the reference's loader):
      * transformer LayerNorm weights x ``ln``, biases 0: every block is near the identity (x + small message), as in a
        converged network where features are refined, not replaced;
      * backbone.conv2 rows zero-mean, no bias: no common-mode component in the matching features;
      * backbone.trident_conv x ``gain``: matching logits ~ gain^2 * 11 * cosine, so the global / local softmax locks onto
        the most similar patch instead of averaging over the image;
      * feature_flow_attn projections = identity: propagation attends to the pixel's own flow.
    On ``texture_frames`` (one pixel of motion per frame — below the 4 / 8 px resolution of the two matching levels) every
    patch locks onto itself: flows are ~0 almost everywhere with a few isolated neighbour matches of up to 4 px,
    forward/backward consistent, no near-ties in the matching softmax, and the whole model is well-conditioned (CPU test
    double: 1e-5 ... 1e-4 end to end).  Large and incoherent flows through the splats are what the teacher-forced render test on
    the random checkpoint covers."""
    sds = dict(gmfss_synth_state_dicts(seed, variant))
    sd = dict(sds["flownet"])
    for k in list(sd):
        if k.startswith("transformer.") and ".norm" in k:
            sd[k] = (sd[k] * (ln if k.endswith("weight") else 0.0)).contiguous()
    w = sd["backbone.conv2.weight"]
    sd["backbone.conv2.weight"] = (w - w.mean(dim=1, keepdim=True)).contiguous()
    sd["backbone.conv2.bias"] = torch.zeros_like(sd["backbone.conv2.bias"])
    sd["backbone.trident_conv.weight"] = (sd["backbone.trident_conv.weight"] * gain).contiguous()
    c = sd["feature_flow_attn.q_proj.weight"].shape[0]
    for n in ("q_proj", "k_proj"):
        sd[f"feature_flow_attn.{n}.weight"] = torch.eye(c)
        sd[f"feature_flow_attn.{n}.bias"] = torch.zeros(c)
    sds["flownet"] = sd
    return sds


def texture_frames(n, h, w, seed=0, shift=1.0, cell=32, c=3):
    """[n,h,w,c] f32 in [0,1]: a smooth random texture (bicubic interpolation of a ``cell``-pixel grid of random colours, no
    pixel noise) translating ``shift`` px/frame horizontally — locally unique patches, stable under the translation: the
    input on which patch matching by feature similarity is well-posed."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    pad = int(abs(shift) * n) + 8
    hh, ww = h + 2 * pad, w + 2 * pad
    base = torch.rand(1, c, hh // cell + 2, ww // cell + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(hh, ww), mode="bicubic", align_corners=False).clamp(0, 1)
    out = []
    for i in range(n):
        dx = shift * i
        ix = int(math.floor(dx))
        fx = dx - ix
        a = img[0, :, pad:pad + h, pad + ix:pad + ix + w]
        b = img[0, :, pad:pad + h, pad + ix + 1:pad + ix + 1 + w]
        out.append(((1 - fx) * a + fx * b).permute(1, 2, 0))
    return torch.stack(out).contiguous().to(torch.float32)


def ifunet_synth_state_dict(seed=1234):
    """IFUNet.pth: conv / linear weights U(+-sqrt(3/fan_in)) (the IFBlocks' flow convs x1.5, ResynNet's last convs x0.5: flows of a few
    pixels), biases U(+-0.05), PReLU slopes U(0.1, 0.4), BatchNorm weight 1 +- 0.1,
    bias +-0.05, running_mean +-0.1, running_var U(0.5, 1.5)."""
    from .ifunet_spec import ifunet_shapes

    sd = {}
    for k, shp in ifunet_shapes().items():
        g = _gen(seed, "ifunet/" + k)
        if k.endswith("num_batches_tracked"):
            t = torch.tensor(100, dtype=torch.int64)
        elif len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in) * (1.5 if "flowconv" in k else 0.5 if "lastconv" in k else 1.0)
        elif k.endswith("running_var"):
            t = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            t = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        elif ".bn." in k or k.split(".")[-2] == "1" and "refinenet.block" in k:   # BatchNorm affine
            t = 1.0 + 0.1 * (torch.rand(shp, generator=g) * 2 - 1) if k.endswith("weight") else (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        elif k.endswith("bias"):
            t = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        else:  # PReLU slopes
            t = 0.1 + 0.3 * torch.rand(shp, generator=g)
        sd[k] = t.contiguous() if t.dtype == torch.int64 else t.to(torch.float32).contiguous()
    return sd


def smooth_frames(n, h, w, seed=0, shift=3.0, c=3):
    """[n,h,w,c] f32 in [0,1]: low-pass noise drifting ``shift`` px/frame (ComfyUI IMAGE layout)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    pad = int(abs(shift) * n) + 8
    hh, ww = h + 2 * pad, w + 2 * pad
    base = torch.rand(1, c, hh // 8 + 2, ww // 8 + 2, generator=g)
    base = torch.nn.functional.interpolate(base, size=(hh, ww), mode="bicubic", align_corners=False)
    fine = torch.rand(1, c, hh, ww, generator=g) * 0.15
    img = (base * 0.85 + fine).clamp(0, 1)
    out = []
    for i in range(n):
        dx = shift * i
        ix = int(math.floor(dx))
        fx = dx - ix
        a = img[0, :, pad + ix // 2 : pad + ix // 2 + h, pad + ix : pad + ix + w]
        b = img[0, :, pad + ix // 2 : pad + ix // 2 + h, pad + ix + 1 : pad + ix + 1 + w]
        out.append(((1 - fx) * a + fx * b).permute(1, 2, 0))
    return torch.stack(out).contiguous().to(torch.float32)


def noise_frames(n, h, w, seed=0, c=3):
    """[n,h,w,c] f32 i.i.d. U[0,1) — BASELINE.md's worst-case-gradient input."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.rand(n, h, w, c, generator=g, dtype=torch.float32)


# ---- "hot" checkpoints: the same keys / shapes with gains that make the networks move pixels by tens of px (trained
# checkpoints do; torch-default init gives 3-5 px).  Goldens: oracle/make_golden_bocchi.py; tests: *_bocchi_* / *_hot_*.

def rife47_hot_state_dict(seed=1234, gain=8.0):
    """RIFE 4.7 with every ``lastconv`` (flow + mask + feature heads) x ``gain``: flows of 40-70 px per block on real
    1080p content, mask logits of tens of units, warps leaving the frame at the borders."""
    return {k: ((v * gain).contiguous() if "lastconv" in k else v) for k, v in rife47_synth_state_dict(seed).items()}


def film_hot_state_dict(seed=1234):
    """FILM: ``film_synth_state_dict``'s variance-preserving gain already yields 20+ px residual flows per pyramid level
    (80+ px accumulated at the finest level of a 1080p pair); gain 1.3 roughly doubles that without overflowing."""
    return film_synth_state_dict(seed, gain=1.3)


def m2m_hot_state_dict(seed=1234):
    """M2M with gain 1.3: PWC flows of ~10 px at quarter resolution, refined multi-branch flows of 30-100 px, i.e. the
    splat works on strongly convergent / divergent fields (occlusions, holes filled by the linear blend)."""
    return m2m_synth_state_dict(seed, gain=1.3)
