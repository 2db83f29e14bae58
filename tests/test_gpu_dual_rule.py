"""An opt-in EXTENSION of the dual loop (ellp_opts.flags = ELLP_FLAG_DUAL_MAX_VIOLATION): the leaving
row is the basic position with the LARGEST bound violation (the first of equals) instead of the reference's first
violated one (dual_simplex_solver.rs:200-236).  Why: under the reference's rule dual phase 1 needs 1.6e5 iterations at
200 x 500 and grows 20-fold per doubling of m; under this one 6.4e3, growing 4-fold (tools/dual_rule_time.py).  It is not
the reference's behaviour, so the checker is the same rule restated in the oracle first (eo_set_dual_rule(2)): the
LU-per-iteration kernels must reproduce it bit for bit, the explicit-inverse engine pivot for pivot, and the optimum
must be the reference rule's optimum."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
from oracle import ellp_oracle as eo
from test_gpu_small import assert_identical, flat

pytestmark = pytest.mark.gpu
MAXVIOL = 2


def _E():
    from ellp_amd import _engine as E
    return E


@pytest.fixture(autouse=True)
def _rule():
    eo.set_dual_rule(2)
    yield
    eo.set_dual_rule(0)


def _both_phases(prob, pipeline, exact, max_iter=400000):
    """dual phase 1 and 2 at the seam, oracle (restated rule) against engine (the flag); returns the final objective"""
    E = _E()
    d1, err = eo.dual_phase1(prob)
    assert d1 is not None and not err
    ph, obj, total = d1, None, 0
    for phase in (1, 2):
        v = ph.view()
        ov = v.copy()
        st_o, it_o, err_o = eo.dual_solve_with_initial(ov, max_iter)
        fp = flat(v)
        st_g, stats, err_g = E.dual_solve_with_initial(fp, E.default_opts(max_iter=max_iter, pipeline=pipeline, flags=MAXVIOL))
        if exact:
            assert_identical((pipeline, phase), ov, st_o, it_o, err_o, fp, st_g, stats, err_g, "dual")
        else:
            assert st_g == st_o and stats.iters == it_o, (pipeline, phase, st_g, st_o, stats.iters, it_o, err_g)
            np.testing.assert_array_equal(fp.B, ov.B)
            np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))
        assert st_o == eo.OPTIMAL
        total += it_o
        if phase == 1:
            ph.store_point(ov)
            ph, e2 = eo.dual_phase2(ph)
            assert ph is not None and not e2
        else:
            obj = ov.obj()
    return obj, total


@pytest.mark.parametrize("name", ["afiro", "adlittle", "blend"])
def test_netlib_bit_for_bit_on_the_persistent_kernel(name):
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    prob = eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, ka["file"])))
    obj, _ = _both_phases(prob, 3, True)
    assert abs(obj / ka["obj"] - 1.0) < 1e-6  # tests/problems/mod.rs:661-673: the rule changes the path, not the optimum


def test_replicated_netlib_bit_for_bit_on_the_mid_kernel():
    ka = next(p for p in known_answers()["netlib"] if p["name"] == "adlittle")
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), 3)
    prob = eo.Problem.from_fixture(permuted_fixture(base, np.random.default_rng(3)))
    obj, _ = _both_phases(prob, 3, True)
    assert abs(obj / (3 * ka["obj"]) - 1.0) < 1e-6


@pytest.mark.parametrize("m,n,pipeline", [(100, 250, 1), (100, 250, 2), (150, 380, 3), (200, 500, 2)])
def test_synthetic_family_same_pivots_and_the_known_optimum(m, n, pipeline):
    """three launches (find_leaving), the fused dual iteration (row-block records) and the mid kernel; the optimum is
    SURVEY.md §8d's independent HiGHS value"""
    highs = {(100, 250): -127.83583703722091, (200, 500): -251.6515333670212, (150, 380): None}[(m, n)]
    obj, total = _both_phases(eo.synth_problem(20260301, m, n), pipeline, pipeline == 3)
    if highs is not None:
        assert abs(obj - highs) < 1e-8 * abs(highs)
    assert total < {100: 3000, 150: 8000, 200: 12000}[m]  # the reference's rule: 27,391 / - / 160,562


def test_closing_work_folded_into_the_next_pricing_launch():
    """m > 512: no tiny-pivot maintenance, the next pricing launch picks the leaving row from the records (dual_fold);
    a window against the explicit-inverse CPU loop is not available for this rule, so: the engine's three forms agree"""
    E = _E()
    from ellp_amd import synth
    f = synth.dual_start_flat(20260301, 600, 1400)
    outs = []
    for pipeline in (1, 2):
        fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"],
                           f["Nb"], f["y"], f["d"])
        st, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=2500, pipeline=pipeline, flags=MAXVIOL))
        outs.append((st, int(stats.iters), fp.B.copy()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
