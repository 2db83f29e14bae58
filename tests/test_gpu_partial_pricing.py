"""Partial pricing (ellp_opts.partial_segments; SURVEY.md §8 f4 — an opt-in extension, the reference prices every
column every iteration): the nonbasic positions are cut into P segments, an iteration prices one of them with the
reference's entering rule, an empty pass moves on, P empty passes in a row are the optimality test.  Checked
against the same rule restated in the oracle (oracle/ellp_oracle.c, eo_set_partial_segments): the same pivots,
pass for pass; and the optimum is the full-pricing optimum."""
import numpy as np
import pytest

from helpers import known_answers
from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu
KA = known_answers()


def _E():
    from ellp_amd import _engine as E
    return E


def _flat(v):
    return _E().FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN])


def _both(view, P, max_iter=eo.MAX_ITER_NONE):
    E = _E()
    ov = view.copy()
    eo.set_partial_segments(P)
    try:
        st_o, it_o, err_o = eo.primal_solve_with_initial(ov, max_iter)
    finally:
        eo.set_partial_segments(0)
    fp = _flat(view)
    st_g, stats, err_g = E.primal_solve_with_initial(fp, E.default_opts(max_iter=None, partial_segments=P, refactor_period=1 << 30))
    return ov, st_o, it_o, fp, st_g, stats, err_g


@pytest.mark.parametrize("P", [2, 3, 7, 1000])
@pytest.mark.parametrize("m,n", [(40, 100), (150, 400)])
def test_same_pivots_as_the_restated_rule(m, n, P):
    p1, err = eo.primal_phase1(eo.synth_problem(20260301 + m, m, n))
    assert p1 is not None and not err
    v1 = p1.view()
    ov, st_o, it_o, fp, st_g, stats, err_g = _both(v1, P)
    assert st_g == st_o == eo.OPTIMAL, err_g
    assert stats.iters == it_o                      # empty passes included, on both sides
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N[:ov.nN])
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))
    # phase 2 from the oracle's phase-1 end point, both ways again
    p1.store_point(ov)
    v2 = eo.primal_phase2(p1).view()
    ov2, st_o2, it_o2, fp2, st_g2, stats2, err_g2 = _both(v2, P)
    assert st_g2 == st_o2, err_g2
    assert stats2.iters == it_o2
    np.testing.assert_array_equal(fp2.B, ov2.B)
    # the optimum does not depend on the pricing rule
    full = v2.copy()
    st_f, it_f, _ = eo.primal_solve_with_initial(full, eo.MAX_ITER_NONE)
    assert st_f == st_o2
    if st_f == eo.OPTIMAL:
        assert abs(float(np.dot(v2.c, fp2.x)) - float(np.dot(v2.c, full.x))) < 1e-8 * (1 + abs(float(np.dot(v2.c, full.x))))


@pytest.mark.parametrize("fx", KA["problems"], ids=[p["name"] for p in KA["problems"]])
def test_known_answers_with_partial_pricing(fx):
    """the reference's small fixtures: every bound kind, infeasible and unbounded outcomes"""
    prob = eo.Problem.from_fixture(fx)
    p1, err = eo.primal_phase1(prob)
    if p1 is None or err:
        pytest.skip("infeasible at setup")
    v1 = p1.view()
    if v1.m == 0 or v1.nN == 0:
        pytest.skip("never reaches the device")
    ov, st_o, it_o, fp, st_g, stats, err_g = _both(v1, 3)
    assert st_g == st_o, err_g
    assert stats.iters == it_o
    if st_o >= 0:
        np.testing.assert_array_equal(fp.B, ov.B)


def test_partial_pricing_prices_a_fraction_of_the_columns():
    """what it is for: the pricing pass reads one segment, so its time falls with P (the 8 ld |N| bytes per pivot of
    SURVEY.md §8d become 8 ld |N| / P) while the iteration count rises"""
    E = _E()
    from ellp_amd import synth
    flat = synth.primal_phase1_flat(20260301, 1000, 6000)
    out = {}
    for P in (1, 8):
        fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"], flat["lb"], flat["ub"],
                           flat["x"], flat["B"], flat["N"], flat["Nb"])
        # flags=1: every column is streamed (the unit-column shortcut already takes the slack columns' 14 % off the full pass)
        eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, partial_segments=P, pipeline=1, profile=1, flags=1))
        st, stats, msg = eng.run(1500)
        out[P] = stats.kernel_ms[0] / max(1, stats.kernel_calls[0])       # ELLP_K_PRICE = 0 (include/ellp_hip.h)
        eng.close()
    assert out[8] < 0.5 * out[1], out
