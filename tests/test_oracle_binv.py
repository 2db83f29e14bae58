"""The explicit-B^-1 / OpenMP variant of the oracle's primal loop (the "same algorithm on the host
cores" CPU baseline and the long-window checker) takes exactly the pivots of the oracle's
LU-per-iteration loop — on the reference's fixtures, on the netlib problems and on the synthetic
family, single- and multi-threaded, with and without periodic re-inversion."""
import numpy as np
import pytest

import os

from helpers import GOLDEN, known_answers, read_mps
from oracle import ellp_oracle as eo

KA = known_answers()


def _same_run(view_a, view_b, max_iter, **kw):
    st_a, it_a, msg_a = eo.primal_solve_with_initial(view_a, max_iter)
    st_b, it_b, msg_b, secs = eo.primal_binv_solve_with_initial(view_b, max_iter, **kw)
    assert st_a == st_b, (st_a, st_b, msg_a, msg_b)
    assert it_a == it_b
    np.testing.assert_array_equal(view_a.B, view_b.B)
    nN = view_a.nN
    np.testing.assert_array_equal(view_a.N[:nN], view_b.N[:nN])
    np.testing.assert_array_equal(view_a.Nb[:nN], view_b.Nb[:nN])
    np.testing.assert_allclose(view_a.x, view_b.x, rtol=0, atol=1e-8)
    assert secs >= 0.0


@pytest.mark.parametrize("fx", KA["problems"], ids=[p["name"] for p in KA["problems"]])
def test_fixture_phase1_same_pivots(fx):
    ph, err = eo.primal_phase1(eo.Problem.from_fixture(fx))
    if not ph or err:
        pytest.skip("phase 1 does not exist (infeasible at setup)")
    a, b = ph.view().copy(), ph.view().copy()
    if a.m == 0:
        pytest.skip("m == 0 goes to the trivial solver before the seam")
    _same_run(a, b, eo.MAX_ITER_NONE, threads=2)


@pytest.mark.parametrize("fx", KA["netlib"], ids=[p["name"] for p in KA["netlib"]])
def test_netlib_two_phases_same_pivots(fx):
    prob = eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"])))
    ph1, err = eo.primal_phase1(prob)
    assert ph1 and not err
    a, b = ph1.view().copy(), ph1.view().copy()
    _same_run(a, b, eo.MAX_ITER_NONE, threads=3)
    ph1.store_point(a)
    ph2 = eo.primal_phase2(ph1)
    a2, b2 = ph2.view().copy(), ph2.view().copy()
    _same_run(a2, b2, eo.MAX_ITER_NONE, threads=3, refresh=25)


@pytest.mark.parametrize("m,n,threads,refresh", [(20, 50, 1, 0), (50, 120, 4, 0), (50, 120, 2, 7), (100, 250, 4, 64)])
def test_synthetic_same_pivots(m, n, threads, refresh):
    from ellp_amd import synth
    f = synth.primal_phase1_flat(20260301, m, n)

    class V:
        pass

    def view():
        v = V()
        for k, val in f.items():
            setattr(v, k, val.copy() if hasattr(val, "copy") else val)
        v.nB, v.nN = len(f["B"]), len(f["N"])
        return v
    _same_run(view(), view(), eo.MAX_ITER_NONE, threads=threads, refresh=refresh)


def test_singular_basis_is_reported():
    from ellp_amd import synth
    f = synth.primal_phase1_flat(5, 6, 9)

    class V:
        pass
    v = V()
    for k, val in f.items():
        setattr(v, k, val.copy() if hasattr(val, "copy") else val)
    v.nB, v.nN = len(f["B"]), len(f["N"])
    cols = v.A.reshape(v.n, v.m)  # flat column-major: row k of this view is column k
    cols[v.B[1], :] = cols[v.B[0], :]  # two equal basic columns
    st, it, msg, _ = eo.primal_binv_solve_with_initial(v, 10, threads=1)
    assert st == eo.ERR_SINGULAR and "not invertible" in msg


def _same_dual_run(view_a, view_b, max_iter, **kw):
    st_a, it_a, msg_a = eo.dual_solve_with_initial(view_a, max_iter)
    st_b, it_b, msg_b, secs = eo.dual_binv_solve_with_initial(view_b, max_iter, **kw)
    assert st_a == st_b, (st_a, st_b, msg_a, msg_b)
    if st_a < 0:
        return
    assert it_a == it_b
    np.testing.assert_array_equal(view_a.B, view_b.B)
    nN = view_a.nN
    np.testing.assert_array_equal(view_a.N[:nN], view_b.N[:nN])
    np.testing.assert_array_equal(view_a.Nb[:nN], view_b.Nb[:nN])
    np.testing.assert_allclose(view_a.x, view_b.x, rtol=0, atol=1e-8)
    np.testing.assert_allclose(view_a.d, view_b.d, rtol=0, atol=1e-8)


@pytest.mark.parametrize("fx", KA["problems"], ids=[p["name"] for p in KA["problems"]])
def test_fixture_dual_phase1_same_pivots(fx):
    ph, err = eo.dual_phase1(eo.Problem.from_fixture(fx))
    if not ph or err:
        pytest.skip("dual phase 1 does not exist")
    a, b = ph.view().copy(), ph.view().copy()
    if a.m == 0:
        pytest.skip("m == 0 goes to the trivial solver before the seam")
    _same_dual_run(a, b, eo.MAX_ITER_NONE, threads=2)


@pytest.mark.parametrize("m,n,threads,refresh", [(20, 50, 1, 0), (50, 120, 4, 0), (50, 120, 2, 9)])
def test_synthetic_dual_same_pivots(m, n, threads, refresh):
    from ellp_amd import synth
    f = synth.dual_start_flat(20260301, m, n)

    class V:
        pass

    def view():
        v = V()
        for k, val in f.items():
            setattr(v, k, val.copy() if hasattr(val, "copy") else val)
        v.nB, v.nN = len(f["B"]), len(f["N"])
        return v
    _same_dual_run(view(), view(), eo.MAX_ITER_NONE, threads=threads, refresh=refresh)
