"""Phase-1 -> phase-2 hand-off on the device (ellp_engine_rephase): one resident
engine runs phase 1, gets the phase-2 costs and bounds (primal_problem.rs:263-291) and runs phase 2
from the basis, point and B^-1 it already holds.  Checked against the ORACLE's phase-2 loop started
from the oracle's own phase-1 end point: same status, same pivots, same point."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, known_answers, read_mps
from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu
KA = known_answers()


def _two_phases_resident(prob, max_iter=1000):
    from ellp_amd import _engine as E
    p1, err = eo.primal_phase1(prob)
    assert p1 is not None and not err
    v1 = p1.view()
    if v1.m == 0 or v1.nN == 0:
        pytest.skip("never reaches the device (trivial problem)")
    fp = E.FlatProblem(v1.m, v1.n, v1.n_c, v1.A, v1.c, v1.b, v1.kind, v1.lb, v1.ub, v1.x, v1.B, v1.N[:v1.nN],
                       v1.Nb[:v1.nN])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None))
    st1, stats1, msg = eng.run(max_iter)
    eng.read_point()
    ov = v1.copy()
    st_o, it_o, _ = eo.primal_solve_with_initial(ov, max_iter)
    assert st1 == st_o == eo.OPTIMAL, msg
    assert stats1.iters == it_o
    np.testing.assert_array_equal(fp.B, ov.B)
    if not abs(ov.obj()) < 1e-10:
        eng.close()
        return "infeasible", None, None
    p1.store_point(ov)
    p2 = eo.primal_phase2(p1)
    v2 = p2.view()
    eng.rephase(v2.c, v2.kind, v2.lb, v2.ub)
    st2, stats2, msg2 = eng.run(max_iter)
    eng.read_point()
    eng.close()
    ov2 = v2.copy()
    st_o2, it_o2, _ = eo.primal_solve_with_initial(ov2, max_iter)
    assert st2 == st_o2, msg2
    return st2, (fp, stats2), (ov2, it_o2)


def _same(fp_stats, oracle, exact=True):
    (fp, stats), (ov, it_o) = fp_stats, oracle
    assert abs(fp.obj() - ov.obj()) <= 1e-9 * (1.0 + abs(ov.obj()))
    if exact:
        assert stats.iters == it_o  # counters restart with the phase, like a new solve_with_initial
        np.testing.assert_array_equal(fp.B, ov.B)
        np.testing.assert_array_equal(fp.N[:fp.nN], ov.N[:ov.nN])
        np.testing.assert_array_equal(fp.Nb[:fp.nN], ov.Nb[:ov.nN])
        np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))


@pytest.mark.parametrize("fx", [p for p in KA["problems"] if p["constraints"]],
                         ids=[p["name"] for p in KA["problems"] if p["constraints"]])
def test_rephase_known_answers(fx):
    p1, err = eo.primal_phase1(eo.Problem.from_fixture(fx))
    if p1 is None or err:
        pytest.skip("infeasible at setup")
    st, g, o = _two_phases_resident(eo.Problem.from_fixture(fx))
    if st == "infeasible":
        assert fx["check"] == "infeasible"
        return
    if st == eo.OPTIMAL:
        _same(g, o)
        if fx["check"] in ("optimal", "optimal_obj"):
            assert abs(g[0].obj() - fx["obj"]) < 1e-8


@pytest.mark.parametrize("fx", KA["netlib"], ids=[p["name"] for p in KA["netlib"]])
def test_rephase_netlib(fx):
    st, g, o = _two_phases_resident(eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"]))))
    assert st == eo.OPTIMAL
    _same(g, o)
    assert abs(g[0].obj() / fx["obj"] - 1.0) < 1e-6


def test_rephase_synthetic_with_free_variable_labels():
    """A free structural variable is Fixed(0) in phase 1 and Free in phase 2: its nonbasic label
    must become Free on the device (primal_problem.rs:285-289)."""
    rng = np.random.default_rng(5)
    m, n = 12, 20
    A = rng.uniform(0.1, 1.1, size=(m, n))
    b = A @ rng.uniform(0, 1, size=n)
    c = -rng.uniform(0.1, 1.1, size=n)
    bounds = [["Lower", 0.0, 0.0]] * n
    bounds[3] = ["Free", 0.0, 0.0]
    bounds[7] = ["TwoSided", -1.0, 2.0]
    fx = {"vars": [[float(c[j]), bounds[j]] for j in range(n)],
          "constraints": [[[[j, float(A[i, j])] for j in range(n)], "Lte", float(b[i])] for i in range(m)]}
    st, g, o = _two_phases_resident(eo.Problem.from_fixture(fx), max_iter=10000)
    assert st in (eo.OPTIMAL, eo.UNBOUNDED)
    if st == eo.OPTIMAL:
        _same(g, o)


def test_rephase_rejects_dual_engines():
    from ellp_amd import _engine as E
    p1, err = eo.dual_phase1(eo.synth_problem(20260301, 10, 20))
    v = p1.view()
    fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN],
                       v.y, v.d)
    eng = E.Engine(E.ENGINE_DUAL, fp, E.default_opts(max_iter=None))
    with pytest.raises(E.EllpHipError):
        eng.rephase(v.c, v.kind, v.lb, v.ub)
    eng.close()
