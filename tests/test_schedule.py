"""Scheduling semantics (multiplier lists, skip/keep lists, output order) against the known answers
obtained by running the reference node with a dummy blend model (tests/golden/rife_schedule_kat.json,
SURVEY.md Appendix A11)."""
import json
import os

import pytest

from cfi_amd.schedule import (InterpolationStateList, MakeInterpolationStateList, rife_multipliers, rife_output_plan,
                              rife_task_list, shard_tasks)

CASES = {
    "m2": dict(multiplier=2),
    "m3_bs2": dict(multiplier=3),
    "mlist": dict(multiplier=[3, 0, 1]),
    "m2_skip12": dict(multiplier=2, states=InterpolationStateList([1, 2], True)),
    "m2_keep12": dict(multiplier=2, states=InterpolationStateList([1, 2], False)),
}


def positions(n, multiplier, states=None):
    _, tasks = rife_task_list(n, multiplier, states)
    plan = rife_output_plan(n, tasks)
    return [float(i) if k == "src" else tasks[i][0] + tasks[i][1] for k, i in plan]


@pytest.mark.parametrize("name", list(CASES))
def test_known_answers(golden_dir, name):
    kat = json.load(open(os.path.join(golden_dir, "rife_schedule_kat.json")))
    got = positions(5, **CASES[name])
    assert len(got) == len(kat[name])
    assert all(abs(a - b) < 1e-4 for a, b in zip(got, kat[name])), (got, kat[name])


def test_multiplier_padding_and_edge_cases():
    assert rife_multipliers(4, [3]) == [3, 2, 2, 2]
    assert rife_multipliers(2, 5) == [5, 5]
    assert positions(1, 2) == [0.0]                      # single frame: passed through
    assert positions(3, 1) == [0.0, 1.0, 2.0]            # multiplier 1: nothing new
    assert positions(2, 4) == [0.0, 0.25, 0.5, 0.75, 1.0]


def test_state_list_node():
    (st,) = MakeInterpolationStateList().create_options("1,2,3", True)
    assert st.is_frame_skipped(2) and not st.is_frame_skipped(0)
    (st,) = MakeInterpolationStateList().create_options("0", False)
    assert st.is_frame_skipped(2) and not st.is_frame_skipped(0)


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (7, 2), (48, 8), (5, 8)])
def test_shard_tasks_partition(n, world):
    tasks = list(range(n))
    parts = [shard_tasks(tasks, r, world) for r in range(world)]
    assert parts[0][0] == 0 and parts[-1][1] == n
    assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in parts]
    assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_frames,mult,bs,skip,ready_mode", [(33, 2, 8, (), "always"), (33, 2, 8, (), "never"), (17, 4, 4, (), "alternate"),
                                                               (12, 3, 8, (2, 5, 6), "always"), (3, 2, 8, (), "never"), (40, 2, 16, (1, 3, 5, 7, 9, 11), "alternate")])
def test_frame_pack_run_ahead_policy(n_frames, mult, bs, skip, ready_mode):
    """rife.run_tasks' frame-pack policy without a device (rife._pick_loads + _FrameSlots): every launch finds its frames resident,
    every frame is packed exactly once and in upload order, frames packed AHEAD of their launch never exceed PACK_AHEAD nor the
    slot count, whatever the uploads' timing (``ready``: always / never / every other query)."""
    from cfi_amd import rife as R
    from cfi_amd.schedule import InterpolationStateList, rife_task_list

    states = InterpolationStateList(list(skip), True) if skip else None
    _, tasks = rife_task_list(n_frames, mult, states)
    batches = list(R._batches(tasks, bs))
    order, last_use = [], {}
    for bi, (_, _, need) in enumerate(batches):
        for f in need:
            if f not in last_use:
                order.append(f)
            last_use[f] = bi
    n_slots = 2 * bs + 2 + R.PACK_AHEAD
    depth = min(len(order), n_slots) or 1
    slots = R._FrameSlots(n_slots)
    calls = [0]

    def ready(i):
        calls[0] += 1
        return ready_mode == "always" or (ready_mode == "alternate" and calls[0] % 2 == 0)

    item, packed = 0, []
    for bi, (pos, bt, need) in enumerate(batches):
        slots.retire([f for f, lu in last_use.items() if lu < bi])
        need_set = set(need)
        while True:
            chunk = R._pick_loads(order, item, need_set, slots, ready, depth)
            if not chunk:
                break
            packed += [f for f, _ in chunk]
            item += len(chunk)
            assert len(set(slots.slot_of.values())) == len(slots.slot_of) <= n_slots
            assert sum(1 for g in slots.slot_of if g not in need_set) <= R.PACK_AHEAD
            if all(f in slots.slot_of for f in need_set):
                break
        assert all(f in slots.slot_of for f in need_set), (bi, need_set, slots.slot_of)
    assert packed == order[:len(packed)] and len(packed) == len(set(packed))
    assert set(packed) >= {f for _, _, need in batches for f in need}
    if ready_mode == "never":       # nothing is packed before the launch that needs it
        assert packed == order
