"""Sharding of the interpolation task list over ranks (one process per GPU, RCCL over xGMI).

The reference has no distributed code; the semantics come from its loop structure: every
``(pair, timestep)`` task is independent (vfi_models/rife/__init__.py:164-207), so the flat task
list is block-partitioned over ranks and the only data-path exchange is gathering the new frames
(SURVEY.md 8e).  Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import torch


def world():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_frames(local, counts):
    """All-gather-v of new frames.

    local:  [counts[rank], H, W, 3] tensor on this rank's device (may have 0 rows)
    counts: per-rank frame counts (known to every rank from the shared task list)
    Returns the concatenation in rank order, [sum(counts), H, W, 3], on every rank.
    Uneven counts are handled by padding to max(counts): one collective, no size exchange.
    """
    import torch.distributed as dist

    rank, ws = world()
    if ws == 1:
        return local
    m = max(counts)
    if m == 0:
        return local
    shape = (m,) + tuple(local.shape[1:])
    send = local
    if local.shape[0] != m:
        send = torch.zeros(shape, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if local.is_cuda and dist.get_backend() == "gloo":
        # plumbing / test mode (several ranks on one GPU, no RCCL): stage through the host
        host = torch.empty((ws,) + shape, dtype=local.dtype)
        dist.all_gather_into_tensor(host.view(ws * m, *shape[1:]), send.contiguous().cpu())
        out = host.to(local.device)
    else:
        out = torch.empty((ws,) + shape, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(ws * m, *shape[1:]), send.contiguous())
    return torch.cat([out[r, : counts[r]] for r in range(ws)], 0)


def broadcast_state_dict(sd, keys, device):
    """Rank 0 holds the checkpoint; ONE broadcast of all tensors as a flat fp32 buffer (21.3 MB for RIFE 4.7), split by the
    shape table every rank knows.  (The single-process path, multidev.py, broadcasts the PACKED device arena instead.)"""
    import torch.distributed as dist

    rank, ws = world()
    if ws == 1:
        return sd
    numels = []
    for shape in keys.values():
        n = 1
        for d in shape:
            n *= int(d)
        numels.append(n)
    if rank == 0:
        flat = torch.cat([sd[k].detach().to(torch.float32).reshape(-1) for k in keys]).to(device)
        assert flat.numel() == sum(numels)
    else:
        flat = torch.empty(sum(numels), dtype=torch.float32, device=device)
    dist.broadcast(flat, src=0)
    flat = flat.cpu()
    out, off = {}, 0
    for (k, shape), n in zip(keys.items(), numels):
        out[k] = flat[off:off + n].view(tuple(shape)).clone()
        off += n
    return out
