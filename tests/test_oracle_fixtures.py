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
