"""N>1 path on CPU: 2 ranks over gloo shard the task list, 'interpolate' with a stand-in blend and
all-gather the new frames; the assembled result must equal the single-process result."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blend(frames, tasks):
    out = [frames[p] * (1 - t) + frames[p + 1] * t for p, t in tasks]
    return torch.stack(out) if out else torch.empty((0,) + tuple(frames.shape[1:]))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pkgload import load_package

    load_package()
    import torch.distributed as dist
    from cfi_amd.dist import all_gather_frames
    from cfi_amd.schedule import rife_output_plan, rife_task_list, shard_tasks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    frames = torch.rand(6, 8, 10, 3, generator=g)
    _, tasks = rife_task_list(6, [3, 2, 1, 4], None)  # 2+1+0+3+1(padded 2) = 7 tasks: uneven shards
    lo, hi = shard_tasks(tasks, rank, world)
    counts = [shard_tasks(tasks, r, world)[1] - shard_tasks(tasks, r, world)[0] for r in range(world)]
    local = _blend(frames, tasks[lo:hi])
    new = all_gather_frames(local, counts)
    plan = rife_output_plan(6, tasks)
    out = torch.stack([frames[i] if k == "src" else new[i] for k, i in plan])
    if rank == 0:
        q.put(out.numpy())      # plain pickle: a torch tensor would travel through torch's shared-memory file descriptors
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    from cfi_amd.schedule import rife_output_plan, rife_task_list

    from mp_util import run_ranks

    got = run_ranks(_worker, 2, timeout=120)
    g = torch.Generator().manual_seed(3)
    frames = torch.rand(6, 8, 10, 3, generator=g)
    _, tasks = rife_task_list(6, [3, 2, 1, 4], None)
    new = _blend(frames, tasks)
    plan = rife_output_plan(6, tasks)
    want = torch.stack([frames[i] if k == "src" else new[i] for k, i in plan])
    assert torch.equal(got, want)
