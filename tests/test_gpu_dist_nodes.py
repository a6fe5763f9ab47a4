"""-m gpu: the N>1 path of the three node classes, for real — 2 ranks (gloo rendezvous, both on the one GPU of the box)
shard the pairs / tasks, interpolate with the HIP engines and all-gather the new frames; every rank must return what a
single process returns (tasks are independent; only fp32 summation order may differ, 2e-5)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_nodes(td):
    """The three nodes on small clips -> dict of outputs (same call in the single- and the multi-process case)."""
    from cfi_amd import film, m2m, synth
    from cfi_amd import rife as R
    from cfi_amd.schedule import InterpolationStateList

    out = {}
    for name, mod, sd in (("rife47", R, synth.rife47_synth_state_dict(1234)), ("film", film, synth.film_synth_state_dict(1234)),
                          ("m2m", m2m, synth.m2m_synth_state_dict(1234))):
        pth = os.path.join(td, f"{name}_{os.getpid()}.pth")
        torch.save(sd, pth)
        mod.load_file_from_github_release = lambda model_type, ckpt, p=pth: p
    R._model_cache.clear()
    fr = synth.smooth_frames(5, 72, 104, seed=21, shift=2.0, c=4)
    out["rife_m3"] = R.RIFE_VFI().vfi("rife47.pth", fr, multiplier=3, batch_size=2)[0]                      # 8 tasks -> 4 + 4
    out["rife_list_skip"] = R.RIFE_VFI().vfi("rife47.pth", fr, multiplier=[2, 3, 1, 4],
                                             optional_interpolation_states=InterpolationStateList([1], True))[0]   # 1 + 3 tasks, uneven
    fr64 = synth.smooth_frames(4, 64, 96, seed=22, shift=2.0)
    out["film_m3"] = film.FILM_VFI().vfi("film_net_fp32.pt", fr64, multiplier=3)[0]                           # 3 pairs -> 2 + 1
    out["m2m_m3"] = m2m.M2M_VFI().vfi("M2M.pth", fr64, multiplier=3)[0]
    out["m2m_list"] = m2m.M2M_VFI().vfi("M2M.pth", fr64, multiplier=[2, 0, 3])[0]
    R._model_cache.clear()
    return out


def _worker(rank, world, port, td):
    sys.path.insert(0, ROOT)
    from pkgload import load_package

    load_package()
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _run_nodes(td)
        torch.save(out, os.path.join(td, f"out_rank{rank}.pt"))   # (tensors through an mp.Queue die with the child process)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_single_process(hip_lib, tmp_path):
    single = _run_nodes(str(tmp_path))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    got = {r: torch.load(os.path.join(str(tmp_path), f"out_rank{r}.pt")) for r in range(2)}
    for rank in (0, 1):
        for k, want in single.items():
            g = got[rank][k]
            assert g.shape == want.shape, (rank, k, g.shape, want.shape)
            # same arithmetic per task whatever the sharding (kernel and tile choice depend on the image, never on how many tasks
            # share a launch — batch invariance is asserted bit-exact elsewhere); the only order that is not fixed is the M2M splat's
            # global-atomic spill pass, hence a rounding-level tolerance instead of torch.equal
            tol = 2e-5
            assert (g - want).abs().max().item() <= tol, (rank, k, (g - want).abs().max().item())
