"""CPU: FILM bisection schedule (vfi_models/film/__init__.py:12-42) — known answers from SURVEY.md A10, which were
obtained by running the reference's inference() with a dummy midpoint model (re-checked against the live reference
by oracle/validate_film_vs_reference.py)."""
from cfi_amd import film_spec


def test_schedule_known_answers():
    from cfi_amd.film import film_schedule

    assert film_schedule(1) == [(0, 2, 1)]
    assert film_schedule(2) == [(0, 3, 2), (0, 2, 1)]            # x3: (0,1,.667) then (0,.5,.5)
    assert film_schedule(3) == [(0, 4, 2), (0, 2, 1), (2, 4, 3)]
    assert film_schedule(4) == [(0, 5, 2), (0, 2, 1), (2, 5, 4), (2, 4, 3)]
    assert [c[2] for c in film_schedule(7)] == [4, 2, 1, 3, 6, 5, 7]


def test_film_spec_matches_survey():
    shapes = film_spec.film_shapes()
    assert len(shapes) == 82
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    assert n == 34436667
    assert [film_spec.feat_channels(l) for l in range(5)] == [64, 192, 448, 960, 960]


def test_c_side_schedule_equals_python(hip_lib):
    """vfi_film_run's bisection order (csrc/clip_run.hip: torch.linspace and the fp32 distance matrix restated in C) equals the
    Python node's film_schedule — which is pinned against the reference's inference() — for 0..24 new frames per pair."""
    import ctypes as C

    from cfi_amd.film import film_schedule

    for inter in range(0, 25):
        buf = (C.c_int * (3 * max(inter, 1)))()
        n = hip_lib.vfi_test_film_schedule(inter, buf, 3 * max(inter, 1))
        assert n == inter
        got = [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]
        assert got == [tuple(int(v) for v in c) for c in film_schedule(inter)], inter


def test_c_side_linspace_equals_torch(hip_lib):
    import numpy as np
    import torch

    for n in range(2, 200):
        out = np.zeros(n, np.float32)
        assert hip_lib.vfi_test_linspace01(n, out.ctypes.data) == n
        assert np.array_equal(out, torch.linspace(0, 1, n).numpy()), n
