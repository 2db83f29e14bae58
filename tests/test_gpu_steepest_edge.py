"""Steepest-edge pricing for the primal loop, an opt-in EXTENSION (ellp_opts.flags =
ELLP_FLAG_PRIMAL_STEEPEST_EDGE; ellp_amd/csrc/engine/ellp_se.inc).  Not the reference's rule (its pivot() is Dantzig's,
primal_simplex_solver.rs:253-287), so the checker is the same rule restated in the oracle first (eo_set_primal_rule(1)):
the engine must take the restated rule's pivots, and the optimum must be the reference rule's (HiGHS's)."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu
SE = 4
HIGHS = {(100, 250): -127.83583703722091, (200, 500): -251.6515333670212}


def _E():
    from ellp_amd import _engine as E
    return E


@pytest.fixture(autouse=True)
def _rule():
    eo.set_primal_rule(1)
    yield
    eo.set_primal_rule(0)


@pytest.mark.parametrize("m,n,flags", [(100, 250, SE), (200, 500, SE), (200, 500, SE | 1), (150, 2000, SE)])
def test_pivots_of_the_restated_rule_and_the_known_optimum(m, n, flags):
    """Both phases at the seam (phase 1 starts from a signed permutation: exact weights from the columns; phase 2 from a
    general basis: exact weights 1 + |B^-1 a_j|^2, the oracle from one LU, the engine from the inverse it has just built);
    flags | 1: every column streamed (no unit-column shortcut in the SE kernel).
    The weights are sums with cancellation: oracle (fresh LU) and engine (explicit inverse) hold them to 1e-12, not to the
    bit, so after some hundreds of pivots a near-tie of two keys r_j^2 / gamma_j can fall the other way and the paths part
    (both are the rule's paths).  Pinned: the same pivots over phase 1 and over the first 150 iterations of phase 2; from
    there the same optimum and an iteration count within 10 %."""
    E = _E()
    p1, err = eo.primal_phase1(eo.synth_problem(20260301, m, n))
    ph = p1
    for phase in (1, 2):
        v = ph.view()
        for budget in ((200000,) if phase == 1 else (150, 200000)):
            ov = v.copy()
            st_o, it_o, _ = eo.primal_solve_with_initial(ov, budget)
            fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN])
            st_g, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=budget, flags=flags))
            assert st_g == st_o, (phase, budget, st_g, st_o, msg)
            if budget == 200000 and phase == 2:
                assert st_o == E.OPTIMAL and abs(stats.iters - it_o) <= 0.1 * it_o, (stats.iters, it_o)
                assert abs(fp.obj() - ov.obj()) < 1e-8 * (1 + abs(ov.obj()))
            else:
                assert stats.iters == it_o, (phase, budget, stats.iters, it_o)
                np.testing.assert_array_equal(fp.B, ov.B)
                np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))
        if phase == 1:
            assert st_o == E.OPTIMAL
            ph.store_point(ov)
            ph = eo.primal_phase2(ph)
    if (m, n) in HIGHS:
        assert abs(ov.obj() - HIGHS[(m, n)]) < 1e-8 * abs(HIGHS[(m, n)])


def test_fewer_iterations_than_dantzig_through_the_user_api():
    """PrimalSimplexSolver::new(None).solve with and without the extension on 400 x 1000: the same optimum, a third of the
    iterations (the weights survive the phase hand-off on the resident engine)"""
    from ellp_amd import Bound, ConstraintOp, PrimalSimplexSolver, Problem, synth
    m, n = 400, 1000
    A, b, c = synth.dense_lp(20260301, m, n)
    res = {}
    for flags in (0, SE):
        p = Problem()
        ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
        for i in range(m):
            p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
        r = PrimalSimplexSolver.new(None, flags=flags, pipeline=1 if flags == 0 else 0).solve(p)  # both on the explicit inverse
        assert r.kind == "optimal"
        res[flags] = (r.solution.obj(), sum(r.iters))
    assert abs(res[0][0] - res[SE][0]) < 1e-8 * abs(res[0][0])
    assert res[SE][1] < res[0][1] // 2, res
