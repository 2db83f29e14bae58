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
BFLIP = 16  # ELLP_FLAG_DUAL_BOUND_FLIPPING (oracle: eo_set_dual_rule bit 0)


def _E():
    from ellp_amd import _engine as E
    return E


@pytest.fixture(autouse=True)
def _rule():
    eo.set_dual_rule(2)
    yield
    eo.set_dual_rule(0)


def _both_phases(prob, pipeline, exact, max_iter=400000, flags=MAXVIOL):
    """dual phase 1 and 2 at the seam, oracle (restated rule) against engine (the flag); returns the final objective"""
    E = _E()
    eo.set_dual_rule((2 if flags & MAXVIOL else 0) | (1 if flags & BFLIP else 0))
    d1, err = eo.dual_phase1(prob)
    assert d1 is not None and not err
    ph, obj, total = d1, None, 0
    for phase in (1, 2):
        v = ph.view()
        ov = v.copy()
        st_o, it_o, err_o = eo.dual_solve_with_initial(ov, max_iter)
        fp = flat(v)
        st_g, stats, err_g = E.dual_solve_with_initial(fp, E.default_opts(max_iter=max_iter, pipeline=pipeline, flags=flags))
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


# ---- the long-step ("bound flipping") ratio test, ELLP_FLAG_DUAL_BOUND_FLIPPING (ellp's README.md:114-116; oracle: bit 0) ----

@pytest.mark.parametrize("flags", [BFLIP, BFLIP | MAXVIOL], ids=["flipping", "flipping+largest-violation"])
@pytest.mark.parametrize("name", ["afiro", "adlittle", "blend"])
def test_bound_flipping_bit_for_bit_on_the_persistent_kernel(name, flags):
    """k_small (m <= 128): the walk over the breakpoints in (ratio, position) order, the flips of the passed boxed variables,
    x_B after one more solve with the iteration's LU, the step from the violation AFTER the flips — status, iteration count,
    index sets and every bit of x, y, d are the oracle's in both phases (phase 1 is a box problem: every variable can flip)"""
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    prob = eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, ka["file"])))
    obj, total = _both_phases(prob, 0 if name == "adlittle" else 3, True, flags=flags)  # with the flag, pipeline 0 selects the exact kernels too
    assert abs(obj / ka["obj"] - 1.0) < 1e-6  # the rule changes the path, not the optimum


@pytest.mark.parametrize("flags", [BFLIP, BFLIP | MAXVIOL], ids=["flipping", "flipping+largest-violation"])
def test_bound_flipping_bit_for_bit_on_the_mid_kernel(flags):
    """k_mid (ADLITTLE x 3, 168 rows; with the flag pipeline 0 runs the LU-per-iteration kernel alone, not the hybrid)"""
    E = _E()
    ka = next(p for p in known_answers()["netlib"] if p["name"] == "adlittle")
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), 3)
    prob = eo.Problem.from_fixture(permuted_fixture(base, np.random.default_rng(3)))
    obj, total = _both_phases(prob, 0, True, flags=flags)
    assert abs(obj / (3 * ka["obj"]) - 1.0) < 1e-6
    # fewer iterations than without the long step (the oracle: 752 / 313 against 1,034 / 344 under the plain ratio test)
    assert total < (800 if flags == BFLIP else 330), total


def test_bound_flipping_on_the_synthetic_family():
    """150 x 380, both extensions, k_mid: bit for bit, and about half the iterations of the largest-violation rule alone
    (3,839 there; SURVEY.md §8d's family has boxed variables in phase 1 only)"""
    obj, total = _both_phases(eo.synth_problem(20260301, 150, 380), 3, True, flags=BFLIP | MAXVIOL)
    assert abs(obj - (-192.0045391718814)) < 1e-8 * 192
    assert total < 2500, total


def test_bound_flipping_where_it_does_not_exist_is_an_error():
    """the caller must be able to tell which rule ran: the explicit-inverse pipelines and LPs above 1,024 rows refuse the flag"""
    E = _E()
    from ellp_amd import synth
    f = synth.dual_start_flat(9, 60, 100)
    def fp_of(f):
        return E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"],
                             f["y"], f["d"])
    for pipeline in (1, 2):
        with pytest.raises(E.EllpHipError) as ei:
            E.Engine(E.ENGINE_DUAL, fp_of(f), E.default_opts(max_iter=None, pipeline=pipeline, flags=BFLIP))
        assert "BOUND_FLIPPING" in str(ei.value)
    with pytest.raises(E.EllpHipError) as ei:
        E.Engine(E.ENGINE_DUAL, fp_of(synth.dual_start_flat(9, 1100, 40)), E.default_opts(max_iter=None, flags=BFLIP))
    assert "1,024" in str(ei.value)
    eng = E.Engine(E.ENGINE_DUAL, fp_of(f), E.default_opts(max_iter=None, flags=BFLIP))
    try:
        with pytest.raises(E.EllpHipError):
            eng.step(0)  # the stepped loop is the explicit-inverse engine's
    finally:
        eng.close()


@pytest.mark.parametrize("name,copies", [("blend", 1), ("adlittle", 3)])
def test_bound_flipping_through_the_user_api(name, copies):
    """DualSimplexSolver::new(None, flags).solve — both phases, the hand-off on the host (the LU-per-iteration engines keep no
    inverse to re-phase from): the pinned optimum (tests/problems/mod.rs:661-673), a feasible point, no more iterations than the
    plain ratio test under the same leaving rule"""
    import ellp_amd
    from helpers import fixture_violation
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    fx = read_mps(os.path.join(GOLDEN, ka["file"]))
    if copies > 1:
        fx = permuted_fixture(blockdiag(fx, copies), np.random.default_rng(3))
    iters = {}
    for flags in (MAXVIOL, MAXVIOL | BFLIP):
        r = ellp_amd.DualSimplexSolver.new(400000, flags=flags).solve(ellp_amd.Problem.from_fixture(fx))
        assert r.kind == ellp_amd.SolverResult.Optimal, (flags, r.kind)
        assert abs(r.solution.obj() / (copies * ka["obj"]) - 1.0) < 1e-9
        v = fixture_violation(fx, r.solution.x())
        assert v[0] < 1e-8 and v[1] < 1e-8, v
        iters[flags] = sum(r.iters)
    assert iters[MAXVIOL | BFLIP] <= iters[MAXVIOL], iters  # the oracle: 47 / 47 (BLEND), 313 / 344 (ADLITTLE x 3)


@pytest.mark.parametrize("flags", [BFLIP, BFLIP | MAXVIOL], ids=["flipping", "flipping+largest-violation"])
def test_bound_flipping_random_lps_bit_for_bit(flags):
    """300 random small LPs with every bound kind and integer data (exact ties between breakpoints, free and fixed variables,
    the last-breakpoint case), 150 feasible box-bounded ones (phase 2 runs; every variable can flip) and 6 wide ones (30-80 rows,
    thousands of columns: the walk's block-wide minima over many chunks): both dual phases on the exact kernel, every bit the
    oracle's under the same rule"""
    import test_gpu_random as R
    E = _E()
    eo.set_dual_rule((2 if flags & MAXVIOL else 0) | 1)
    cases = [R.random_fixture(np.random.default_rng(s)) for s in range(21000, 21300)]
    cases += [R.feasible_fixture(np.random.default_rng(s)) for s in range(27000, 27150)]
    cases += [R.wide_fixture(np.random.default_rng(s)) for s in range(100, 106)]
    ran = flipped_somewhere = 0
    for k, fx in enumerate(cases):
        prob = eo.Problem.from_fixture(fx)
        d1, err = eo.dual_phase1(prob)
        if d1 is None or err:
            continue
        ph = d1
        for phase in (1, 2):
            v = ph.view()
            if v.m == 0:
                break
            ov = v.copy()
            st_o, it_o, err_o = eo.dual_solve_with_initial(ov, 3000)
            fp = flat(v)
            st_g, stats, err_g = E.dual_solve_with_initial(fp, E.default_opts(max_iter=3000, pipeline=0, flags=flags))
            assert_identical((k, phase), ov, st_o, it_o, err_o, fp, st_g, stats, err_g, "dual")
            ran += 1
            if phase == 1 and st_o == eo.OPTIMAL:
                # the plain ratio test from the same arrays: a different iteration count means the walk passed breakpoints
                eo.set_dual_rule(2 if flags & MAXVIOL else 0)
                pv = v.copy()
                _, it_p, _ = eo.dual_solve_with_initial(pv, 3000)
                eo.set_dual_rule((2 if flags & MAXVIOL else 0) | 1)
                flipped_somewhere += int(it_p != it_o)
            if phase == 2 or st_o != eo.OPTIMAL:
                break
            ph.store_point(ov)
            ph, e2 = eo.dual_phase2(ph)
            if ph is None or e2:
                break
    assert ran > 300, ran
    assert flipped_somewhere > 20, flipped_somewhere  # the long step really is taken in this population
