"""M2M node loop on CPU: generic_output_plan / run_plan (host logic of comfyui-frame-interpolation_amd/m2m.py) with a
stand-in engine that calls the oracle model, against the oracle's restatement of generic_frame_loop
(vfi_utils.py:149-389) — int and list multipliers, skip lists, the m == 0 quirks — single process and 2 ranks (gloo)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """prepare/render interface of M2MEngine on the CPU, backed by the oracle (test infrastructure only)."""

    def __init__(self, sd):
        self.sd, self.device = sd, torch.device("cpu")

    def prepare(self, f0, f1):
        self.pair = (f0.permute(2, 0, 1)[None], f1.permute(2, 0, 1)[None])

    def render(self, t, out=None):
        from oracle import m2m_model_oracle as mo

        with torch.inference_mode():
            y = mo.m2m_forward(self.sd, self.pair[0], self.pair[1], [torch.tensor([t]).view(1, 1, 1, 1)])[0][0].permute(1, 2, 0)
        if out is not None:
            out.copy_(y)
        return y


def _case():
    from cfi_amd import synth

    return synth.m2m_synth_state_dict(1234), synth.smooth_frames(4, 64, 64, seed=2, shift=2.0)


def _states(spec):
    from cfi_amd.schedule import InterpolationStateList

    return None if spec is None else InterpolationStateList(*spec)


CASES = [(2, None), (3, ([1], True)), (2, ([0, 2], False)), ([2, 0, 3], None), ([1, 2, 0], None), ([3], ([0], False)), ([0, 0, 0], None)]


def test_plan_known_answers():
    from cfi_amd.schedule import generic_output_plan

    plan, tasks = generic_output_plan(3, 3, None)
    assert plan == [("src", 0), ("new", 0), ("new", 1), ("src", 1), ("new", 2), ("new", 3), ("src", 2)]
    assert tasks == [(0, [1 / 3, 2 / 3]), (1, [1 / 3, 2 / 3])]
    plan, tasks = generic_output_plan(4, [2, 0], None)       # padded with 2 -> [2, 0, 2]; pair 1 dropped with its frame
    assert plan == [("src", 0), ("new", 0), ("src", 2), ("new", 1), ("src", 3)]
    plan, tasks = generic_output_plan(3, [2, 0], None)       # last pair dropped: the clip's last frame is never appended
    assert plan == [("src", 0), ("new", 0)]
    with pytest.raises(NotImplementedError):
        generic_output_plan(3, 2.0, None)
    with pytest.raises(ValueError, match="negative"):       # the reference's torch.zeros(multiplier * 2, ...) raises (vfi_utils.py:178)
        generic_output_plan(3, [2, -1], None)


@pytest.mark.parametrize("multiplier,spec", CASES)
def test_run_plan_matches_oracle_loop(multiplier, spec):
    from cfi_amd.m2m import run_plan
    from cfi_amd.schedule import generic_output_plan
    from oracle import m2m_model_oracle as mo

    sd, fr = _case()
    plan, tasks = generic_output_plan(len(fr), multiplier, _states(spec))
    if not plan:   # every pair dropped: the reference dies in torch.cat([]) (vfi_utils.py:386); the node raises as well
        with pytest.raises((RuntimeError, ValueError)):
            mo.m2m_vfi(sd, fr, multiplier, _states(spec))
        with pytest.raises(RuntimeError):
            run_plan(OracleEngine(sd), fr, plan, tasks)
        return
    got = run_plan(OracleEngine(sd), fr, plan, tasks)
    want = mo.m2m_vfi(sd, fr, multiplier, _states(spec))
    # not bit-equal: the oracle loop slices frames out of the clip tensor, run_plan copies them (different alignment ->
    # different vectorised summation order in torch's mean/std), ~1e-5
    assert got.shape == want.shape and (got - want).abs().max().item() <= 1e-4
    for i, (kind, idx) in enumerate(plan):
        if kind == "src":
            assert torch.equal(got[i], fr[idx])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pkgload import load_package

    load_package()
    import torch.distributed as dist
    from cfi_amd.m2m import run_plan
    from cfi_amd.schedule import generic_output_plan
    from test_m2m_schedule import OracleEngine, _case

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd, fr = _case()
    plan, tasks = generic_output_plan(len(fr), [3, 2, 2], None)   # 2 + 1 + 1 new frames over 3 pairs: uneven shards
    out = run_plan(OracleEngine(sd), fr, plan, tasks)
    if rank == 0:
        q.put(out.numpy())      # plain pickle: a torch tensor would travel through torch's shared-memory file descriptors
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_oracle_loop():
    from oracle import m2m_model_oracle as mo

    from mp_util import run_ranks

    got = run_ranks(_worker, 2, timeout=300)
    sd, fr = _case()
    want = mo.m2m_vfi(sd, fr, [3, 2, 2], None)
    assert got.shape == want.shape and (got - want).abs().max().item() <= 1e-4
