"""Interpolation schedule semantics of the reference nodes (pure Python, no GPU).

Mirrors — by behaviour, not by text — the reference's
  * ``InterpolationStateList`` / ``MakeInterpolationStateList``   vfi_utils.py:49-81
  * the RIFE node's per-pair multiplier list and task list        vfi_models/rife/__init__.py:149-174
  * the RIFE node's output interleave                             vfi_models/rife/__init__.py:225-230
Known answers for these are in SURVEY.md Appendix A11 and tests/test_schedule.py.
"""
import typing


class InterpolationStateList:
    """Skip-list (``is_skip_list=True``) or keep-list of frame-pair indices."""

    def __init__(self, frame_indices: typing.List[int], is_skip_list: bool):
        self.frame_indices = frame_indices
        self.is_skip_list = is_skip_list

    def is_frame_skipped(self, frame_index):
        listed = frame_index in self.frame_indices
        return listed if self.is_skip_list else not listed


class MakeInterpolationStateList:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "frame_indices": ("STRING", {"multiline": True, "default": "1,2,3"}),
                "is_skip_list": ("BOOLEAN", {"default": True},),
            },
        }

    RETURN_TYPES = ("INTERPOLATION_STATES",)
    FUNCTION = "create_options"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def create_options(self, frame_indices: str, is_skip_list: bool):
        indices = [int(item) for item in frame_indices.split(",")]
        return (InterpolationStateList(frame_indices=indices, is_skip_list=is_skip_list),)


def rife_multipliers(n_pairs, multiplier):
    """int -> same for every pair; list -> per pair, missing entries padded with 2."""
    if isinstance(multiplier, int):
        return [int(multiplier)] * n_pairs
    ms = list(map(int, multiplier))
    ms += [2] * (n_pairs - len(ms))
    return ms


def rife_task_list(n_frames, multiplier, states=None):
    """Flat list of ``(pair_idx, timestep)``; one task = one new frame.

    A skipped pair, or a pair with multiplier <= 1, yields no task (its first frame is
    still passed through by :func:`rife_output_plan`)."""
    n_pairs = n_frames - 1
    ms = rife_multipliers(n_pairs, multiplier)
    tasks = []
    for pair in range(n_pairs):
        if states is not None and states.is_frame_skipped(pair):
            continue
        m = ms[pair]
        for step in range(1, m):
            tasks.append((pair, step / m))
    return ms, tasks


def rife_output_plan(n_frames, tasks):
    """Output order as a list of ``("src", frame_idx)`` / ``("new", task_idx)`` entries:
    frame_0, its new frames in task order, frame_1, ..., frame_last."""
    per_pair = {}
    for ti, (pair, _) in enumerate(tasks):
        per_pair.setdefault(pair, []).append(ti)
    plan = []
    for pair in range(n_frames - 1):
        plan.append(("src", pair))
        for ti in per_pair.get(pair, ()):
            plan.append(("new", ti))
    plan.append(("src", n_frames - 1))
    return plan


def generic_output_plan(n_frames, multiplier, states=None):
    """Output order of ``generic_frame_loop`` in timestep mode (vfi_utils.py:149-389), used by the M2M node.

    Returns ``(plan, tasks)``: ``plan`` = list of ``("src", frame_idx)`` / ``("new", k)`` entries in output order,
    ``tasks`` = list of ``(pair_idx, [timesteps])`` whose new frames are numbered k = 0.. in order.
    int multiplier: frame_i, its m-1 middle frames (none when the pair is skipped), ..., last frame.
    list multiplier (padded with 2): each pair runs as its own 2-frame loop — m == 0 drops the pair INCLUDING its
    first frame (and the clip's last frame when it is the last pair), and the skip list is consulted with the
    pair's LOCAL index 0 (vfi_utils.py:364-386 passes a 2-frame slice to the same loop)."""
    plan, tasks, n_new = [], [], 0

    def one_pair(pair, m, skipped):
        nonlocal n_new
        plan.append(("src", pair))
        if skipped or m <= 1:
            return
        ts = [k / m for k in range(1, m)]
        tasks.append((pair, ts))
        for _ in ts:
            plan.append(("new", n_new))
            n_new += 1

    if type(multiplier) == int:
        for pair in range(n_frames - 1):
            one_pair(pair, multiplier, states is not None and states.is_frame_skipped(pair))
        plan.append(("src", n_frames - 1))
    elif type(multiplier) == list:
        ms = list(map(int, multiplier))
        ms += [2] * (n_frames - len(ms) - 1)
        for pair in range(n_frames - 1):
            if ms[pair] == 0:
                continue
            if ms[pair] < 0:       # the reference allocates torch.zeros(multiplier * 2, ...) for the pair: RuntimeError (vfi_utils.py:178)
                raise ValueError(f"multiplier {ms[pair]} of pair {pair}: the reference's frame loop fails on negative multipliers")
            one_pair(pair, ms[pair], states is not None and states.is_frame_skipped(0))
            if pair == n_frames - 2:
                plan.append(("src", n_frames - 1))
    else:
        raise NotImplementedError(f"multipiler of {type(multiplier)}")
    return plan, tasks


def shard_tasks(tasks, rank, world):
    """Contiguous block partition of the task list over ``world`` ranks.

    Returns ``(lo, hi)`` so that rank r owns ``tasks[lo:hi]``; blocks differ in size by at
    most one task and concatenating them in rank order restores the original order."""
    n = len(tasks)
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi
