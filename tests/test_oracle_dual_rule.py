"""The oracle's restatement of two dual EXTENSIONS (oracle/ellp_oracle.c, eo_set_dual_rule; not the
reference's rules): bit 1 the leaving row of largest violation, bit 0 the bound-flipping ratio test.  They change the
path, never the answer: every known answer of the reference (tests/problems/mod.rs:130-674) must come out as under the
reference's own rules, and on the synthetic family the iteration counts must fall as tools/dual_rule_time.py reports."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, check_result, known_answers, read_mps
from oracle import ellp_oracle as eo

KA = known_answers()


@pytest.fixture(autouse=True)
def _restore():
    yield
    eo.set_dual_rule(0)


@pytest.mark.parametrize("rule", [1, 2, 3])
@pytest.mark.parametrize("fx", KA["problems"], ids=[p["name"] for p in KA["problems"]])
def test_known_answers_under_the_extensions(fx, rule):
    eo.set_dual_rule(rule)
    r = eo.solve(eo.Problem.from_fixture(fx), "dual", None)
    if fx["check"] in ("optimal", "optimal_obj") and r.status != eo.OPTIMAL:
        pytest.fail(f"status {r.status} {r.err}")
    # the point of a degenerate optimum may be another vertex of the optimal face: objective only for those
    if fx["check"] == "optimal":
        assert r.status == eo.OPTIMAL and abs(r.obj - fx["obj"]) < 1e-8
    else:
        check_result(fx, eo.STATUS_NAME.get(r.status, str(r.status)), r.obj, r.x)


@pytest.mark.parametrize("rule", [2, 3])
@pytest.mark.parametrize("fx", KA["netlib"], ids=[p["name"] for p in KA["netlib"]])
def test_netlib_under_the_extensions(fx, rule):
    eo.set_dual_rule(rule)
    r = eo.solve(eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"]))), "dual", None)
    assert r.status == eo.OPTIMAL and abs(r.obj / fx["obj"] - 1.0) < 1e-6


def test_iteration_counts_on_the_synthetic_family():
    counts = {}
    for rule in (0, 1, 2, 3):
        eo.set_dual_rule(rule)
        r = eo.solve(eo.synth_problem(20260301, 100, 250), "dual", None)
        assert r.status == eo.OPTIMAL and abs(r.obj - (-127.83583703722091)) < 1e-8 * 128
        counts[rule] = sum(r.iters)
    assert counts[0] > 20000 and counts[2] < counts[0] // 10 and counts[3] < counts[2]


@pytest.mark.parametrize("fx", KA["problems"] + KA["netlib"], ids=[p["name"] for p in KA["problems"] + KA["netlib"]])
def test_known_answers_under_steepest_edge_pricing(fx):
    """the primal extension (eo_set_primal_rule(1), exact steepest-edge weights): same answers as the reference's Dantzig rule"""
    eo.set_primal_rule(1)
    try:
        if "file" in fx:
            r = eo.solve(eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"]))), "primal", None)
            assert r.status == eo.OPTIMAL and abs(r.obj / fx["obj"] - 1.0) < 1e-6
            return
        r = eo.solve(eo.Problem.from_fixture(fx), "primal", None)
        if fx["check"] == "optimal":
            assert r.status == eo.OPTIMAL and abs(r.obj - fx["obj"]) < 1e-8
        else:
            check_result(fx, eo.STATUS_NAME.get(r.status, str(r.status)), r.obj, r.x)
    finally:
        eo.set_primal_rule(0)
