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


def test_hold_keeps_a_forked_stages_blocks_until_the_join():
    """OpsEngine._hold (the pool recycles by program order, i.e. for one stream): scopes that end while the hold is active keep their
    blocks — a stage issued afterwards on the current stream cannot be handed them — and everything returns when the hold ends."""
    from emu_backend import EmuBackend

    from cfi_amd.opsengine import OpsEngine

    eng = OpsEngine(_test_backend=EmuBackend(), pooled=True)
    with eng._scope():
        a = eng._t("warm", 1, 64, 64, 8)      # grows the pool's first chunk
    free_before = [list(map(list, holes)) for holes in eng._pool.free]
    with eng._hold() as hold:
        hold["active"] = True
        with eng._scope():
            side = eng._t("side_tmp", 1, 32, 32, 8)
        hold["active"] = False
        with eng._scope():
            main = eng._t("main_tmp", 1, 32, 32, 8)
            assert main.data_ptr() != side.data_ptr(), "the held block was recycled under the forked stage"
        assert len(hold["blocks"]) == 1
    assert eng._held is None and [list(map(list, holes)) for holes in eng._pool.free] == free_before
    with eng._scope():      # after the join the block is anybody's again
        again = eng._t("later", 1, 32, 32, 8)
        assert again.data_ptr() == side.data_ptr()
    assert eng._fork() == (None, None)      # nothing to fork onto without a device
