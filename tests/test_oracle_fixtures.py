"""Pins the CPU oracle against EVERY known answer the reference's tests hold
(tests/problems/mod.rs:130-674 via tests/integration_tests.rs:51-127; README.md:88-106),
for both solvers, exactly as `generate_tests!` does: solver.solve(prob) then check_result."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, check_result, known_answers, read_mps
from oracle import ellp_oracle as eo

KA = known_answers()


@pytest.mark.parametrize("solver", ["primal", "dual"])
@pytest.mark.parametrize("fx", KA["problems"], ids=[p["name"] for p in KA["problems"]])
def test_known_answer(fx, solver):
    prob = eo.Problem.from_fixture(fx)
    res = eo.solve(prob, solver)  # Default::default(): max_iter 1000
    assert res.status >= 0, res.err
    check_result(fx, eo.STATUS_NAME[res.status], res.obj, res.x, KA["abs_eps"], KA["rel_eps"])


@pytest.mark.parametrize("solver", ["primal", "dual"])
@pytest.mark.parametrize("fx", KA["netlib"], ids=[p["name"] for p in KA["netlib"]])
def test_netlib(fx, solver):
    prob = eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"])))
    res = eo.solve(prob, solver)
    assert res.status >= 0, res.err
    check_result(fx, eo.STATUS_NAME[res.status], res.obj, res.x, KA["abs_eps"], KA["rel_eps"])


@pytest.mark.parametrize("solver", ["primal", "dual"])
def test_readme_output(solver):
    """README.md:88-106 prints 17 significant digits for both solvers."""
    fx = next(p for p in KA["problems"] if p["name"] == KA["readme"]["problem"])
    res = eo.solve(eo.Problem.from_fixture(fx), solver)
    assert res.status == eo.OPTIMAL
    assert abs(res.obj - KA["readme"]["obj"]) < 1e-13
    np.testing.assert_allclose(res.x, KA["readme"]["x"], rtol=0, atol=1e-14)


def test_partial_pricing_restatement_reaches_the_same_optimum():
    """eo_set_partial_segments (the checker of the engine's opt-in partial pricing, an extension: SURVEY.md §8 f4):
    whatever the number of segments, phase 1 ends feasible and phase 2 at the full-pricing optimum; one segment
    per column and more segments than columns are legal"""
    import numpy as np
    prob = eo.synth_problem(20260301, 30, 70)
    objs = {}
    for P in (0, 2, 5, 70, 1000):
        eo.set_partial_segments(P)
        try:
            p1, err = eo.primal_phase1(prob)
            v1 = p1.view()
            st1, it1, _ = eo.primal_solve_with_initial(v1, eo.MAX_ITER_NONE)
            assert st1 == eo.OPTIMAL and abs(v1.obj()) < 1e-9
            p1.store_point(v1)
            v2 = eo.primal_phase2(p1).view()
            st2, it2, _ = eo.primal_solve_with_initial(v2, eo.MAX_ITER_NONE)
            assert st2 == eo.OPTIMAL
            objs[P] = (float(np.dot(v2.c, v2.x)), it1 + it2)
        finally:
            eo.set_partial_segments(0)
    ref = objs[0][0]
    for P, (o, it) in objs.items():
        assert abs(o - ref) < 1e-8 * (1 + abs(ref)), (P, o, ref)
    assert objs[5][1] != objs[0][1]      # the rule is really another one: it takes another path
