"""Pair lanes, host logic (comfyui-frame-interpolation_amd/lanes.py): lane count policy, per-call settings reaching every lane,
ownership of a borrowed first engine, the environment override.  (Streams and engines on a device: tests/test_gpu_pair_lanes.py.)"""
import torch

from cfi_amd import lanes as LN


class FakeEngine:
    built = 0

    def __init__(self):
        FakeEngine.built += 1
        self.device = torch.device("cpu")
        self.closed = self.released = False
        self.embt = None

    def close(self):
        self.closed = True

    def release_workspace(self):
        self.released = True


def test_lane_count_follows_the_clip():
    ls = LN.LaneSet(FakeEngine, 3)
    assert [LN.lanes_of(ls, n)[1] for n in (1, 2, 3, 4, 5, 6, 7, 100)] == [1, 1, 1, 2, 2, 3, 3, 3]
    film = LN.LaneSet(FakeEngine, 2, pairs_per_lane=LN.PAIRS_PER_LANE["film"])
    assert [LN.lanes_of(film, n)[1] for n in (4, 24, 47, 48, 200)] == [1, 1, 1, 2, 2]
    one = LN.LaneSet(FakeEngine, 1)
    assert LN.lanes_of(one, 50)[1] == 1
    plain = FakeEngine()
    get, n = LN.lanes_of(plain, 50)
    assert n == 1 and get(0)[0] is plain and get(0)[1] is None      # a plain engine: one lane, the caller's stream


def test_settings_reach_every_lane_and_borrowed_engine_survives():
    first = FakeEngine()
    ls = LN.LaneSet(FakeEngine, 3, first=first)
    ls.engines.append(FakeEngine())      # (a second lane, as lane(1) would have built it on a device)
    LN.configure(ls, lambda e: setattr(e, "embt", 0.5))
    assert [e.embt for e in ls.engines] == [0.5, 0.5]
    LN.configure(first, lambda e: setattr(e, "embt", 0.25))      # a plain engine takes the same call
    assert first.embt == 0.25
    second = ls.engines[1]
    ls.release_workspace()
    assert first.released and second.released
    ls.close()
    assert second.closed and not first.closed and ls.engines == []


def test_environment_override(monkeypatch):
    monkeypatch.delenv("VFI_PAIR_LANES", raising=False)
    assert LN.lanes_for("m2m") == 3 and LN.lanes_for("film") == 2 and LN.lanes_for("rife") == 1
    monkeypatch.setenv("VFI_PAIR_LANES", "1")
    assert LN.lanes_for("m2m") == 1 and LN.lane_set("gmfss", FakeEngine).k == 1
    monkeypatch.setenv("VFI_PAIR_LANES", "5")
    assert LN.lane_set("film", FakeEngine).k == 5 and LN.lane_set("film", FakeEngine).pairs_per_lane == 24
