"""DualPhase2::from(phase_1) on the device (ellp_engine_dual_rephase; dual_problem.rs:258-404, SURVEY.md §8
f2): after phase 1 the engine keeps its matrix, basis and B^-1 in HBM; y, d, the nonbasic values and
labels, x_B, N in variable order and the dual objective are rebuilt there.  Checked against the arrays
the oracle's own construction produces from the same phase-1 end point, then phase 2 is run from both."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def _flat(v):
    return _E().FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)


def _case(prob):
    E = _E()
    d1, err = eo.dual_phase1(prob)
    assert d1 is not None and not err
    v1 = d1.view()
    # phase 1 on the engine (the reference's dual phase 1 needs ~10^5 pivots at this size: seconds here,
    # minutes for the LU-per-iteration oracle); its END POINT is handed to the oracle's own phase-2
    # construction, so both constructions start from the same basis
    fp = _flat(v1)
    eng = E.Engine(E.ENGINE_DUAL, fp, E.default_opts(max_iter=None, pipeline=1))
    st_g, stats, msg = eng.run(1 << 40)
    assert st_g == E.OPTIMAL, msg
    eng.read_point()
    ov1 = v1.copy()
    ov1.x[:] = fp.x
    ov1.B[:] = fp.B
    ov1.N[:fp.nN] = fp.N
    ov1.Nb[:fp.nN] = fp.Nb
    ov1.y[:] = fp.y
    ov1.d[:] = fp.d
    d1.store_point(ov1)
    assert d1.dual_obj() > -1e-8                           # phase 1 ended dual feasible
    d2, err2 = eo.dual_phase2(d1)
    assert d2 is not None and not err2
    v2 = d2.view()
    assert v2.m == v1.m and v2.n == v1.n and np.array_equal(np.asarray(v2.A), np.asarray(v1.A))  # same matrix in both phases
    eng.dual_rephase(v2.c, v2.b, v2.kind, v2.lb, v2.ub)
    eng.read_point()
    np.testing.assert_array_equal(fp.B, v2.B)
    np.testing.assert_array_equal(fp.N, v2.N[:v2.nN])     # variable order
    np.testing.assert_array_equal(fp.Nb, v2.Nb[:v2.nN])
    sc = 1 + max(np.abs(v2.x).max(), np.abs(v2.y).max(), np.abs(v2.d).max())
    np.testing.assert_allclose(fp.y, v2.y, rtol=0, atol=1e-9 * sc)
    np.testing.assert_allclose(fp.d, v2.d, rtol=0, atol=1e-9 * sc)
    np.testing.assert_allclose(fp.x, v2.x, rtol=0, atol=1e-9 * sc)
    # phase 2 from both
    ov2 = v2.copy()
    st_o2, it_o2, _ = eo.dual_solve_with_initial(ov2, eo.MAX_ITER_NONE)
    st_g2, stats2, msg2 = eng.run(1 << 40)
    eng.read_point()
    eng.close()
    assert st_g2 == st_o2, msg2
    if st_o2 == eo.OPTIMAL:
        assert abs(float(np.dot(v2.c, fp.x)) - float(np.dot(v2.c, ov2.x))) < 1e-8 * (1 + abs(float(np.dot(v2.c, ov2.x))))
    return st_o2, int(stats2.iters), it_o2


@pytest.mark.parametrize("m,n", [(150, 300), (200, 500)])
def test_dual_rephase_on_the_synthetic_family(m, n):
    st, it_g, it_o = _case(eo.synth_problem(20260301, m, n))
    assert st == eo.OPTIMAL


def test_dual_rephase_with_upper_bounds_and_free_variables():
    """Lower / Upper / Free variables (none TwoSided or Fixed, so that the box problem keeps every column),
    >= and <= rows: labels Upper and Free and the assertions on the sign of d are exercised"""
    rng = np.random.default_rng(5)
    m, n = 140, 260
    A = rng.uniform(0.1, 1.1, size=(m, n))
    x0 = rng.uniform(0.2, 1.0, size=n)
    vars_, cons = [], []
    for j in range(n):
        u = rng.random()
        if u < 0.6:
            vars_.append([float(rng.uniform(0.1, 1.0)), ["Lower", 0.0, 0.0]])
        elif u < 0.9:
            vars_.append([float(-rng.uniform(0.1, 1.0)), ["Upper", 0.0, 2.0]])
        else:
            vars_.append([0.0, ["Free", 0.0, 0.0]])
    for i in range(m):
        ax = float(A[i] @ x0)
        if i % 3 == 0:
            cons.append([[[j, float(A[i, j])] for j in range(n)], "Gte", ax * 0.5])
        else:
            cons.append([[[j, float(A[i, j])] for j in range(n)], "Lte", ax * 1.5])
    _case(eo.Problem.from_fixture({"vars": vars_, "constraints": cons}))


@pytest.mark.parametrize("engine", [{"pipeline": 1}, {}], ids=["explicit-inverse", "default"])
def test_dual_solver_end_to_end_on_one_resident_engine(engine):
    """DualSimplexSolver::new(None).solve(problem) through the C++ host mirror.  On an engine of the explicit-inverse
    kind (m > 512 by default; asked for here) with no dropped variables both phases are slices of ONE resident engine
    (matrix uploaded once, DualPhase2::from done on the device from a freshly rebuilt inverse).  With default options
    this size runs the LU-per-iteration kernel, which keeps no inverse: the hand-off is then done on the host, as the
    reference does it (DualPhase2::point_on_host).  Either way the optimum must be HiGHS's"""
    from scipy.optimize import linprog
    from ellp_amd import DualSimplexSolver, Problem
    m, n = 150, 300
    A, b, c = eo.synth_dense_lp(20260301, m, n)
    fx = {"vars": [[float(c[j]), ["Lower", 0.0, 0.0]] for j in range(n)],
          "constraints": [[[[j, float(A[i, j])] for j in range(n)], "Lte", float(b[i])] for i in range(m)]}
    r = DualSimplexSolver.new(None, **engine).solve(Problem.from_fixture(fx))
    assert r.kind == "optimal"
    h = linprog(c, A_ub=A, b_ub=b, bounds=(0, None), method="highs")
    assert abs(r.solution.obj() - h.fun) < 1e-8 * (1 + abs(h.fun))
    x = np.asarray(r.solution.x())
    assert np.all(A @ x[:n] <= b + 1e-8) and x[:n].min() > -1e-9
