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
