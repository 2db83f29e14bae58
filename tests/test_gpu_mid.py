"""The persistent exact kernel for LPs whose basis no longer fits LDS (ellp_amd/csrc/engine/ellp_mid.inc,
128 < m <= 1024; the default up to m = 512): the oracle's floating-point operations in the oracle's order,
with the factors in global memory and the LU blocked by panels — so everything must be EQUAL to the oracle:
iteration counts, index sets and the bits of x, y, d, for the primal and the dual loop, on dense synthetic
LPs (every workgroup size), on block-diagonal replications of the netlib LPs and on random LPs of every
bound kind."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
from oracle import ellp_oracle as eo
from test_gpu_small import assert_identical, both

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def two_phases_mid(fx, which, max_iter):
    """phase 1 and phase 2 at the seam through pipeline 3, each from the oracle's arrays, equal bits"""
    prob = eo.Problem.from_fixture(fx)
    p1, err = (eo.primal_phase1 if which == "primal" else eo.dual_phase1)(prob)
    assert p1 is not None and not err
    v1 = p1.view()
    assert 128 < v1.m <= 1024
    r = both(v1, which, max_iter)
    assert_identical((which, 1), *r, which)
    ov = r[0]
    if r[1] != eo.OPTIMAL:
        return 1, r[1]
    p1.store_point(ov)
    if which == "primal":
        if not abs(ov.obj()) < 1e-10:
            return 1, r[1]
        v2 = eo.primal_phase2(p1).view()
    else:
        p2, err2 = eo.dual_phase2(p1)
        if p2 is None or err2:
            return 1, r[1]
        v2 = p2.view()
    r2 = both(v2, which, max_iter)
    assert_identical((which, 2), *r2, which)
    return 2, r2[1]


@pytest.mark.parametrize("m,n,iters", [(129, 300, 400), (150, 380, 100000), (256, 500, 300), (300, 700, 250),
                                       (513, 1100, 60), (700, 1500, 40)])
def test_dense_synthetic_primal_bit_for_bit(m, n, iters):
    """every workgroup size (256, 512, 1024 threads), partial last panels, several trailing tiles"""
    E = _E()
    p1, err = eo.primal_phase1(eo.synth_problem(20260301 + m, m, n))
    v = p1.view()
    r = both(v, "primal", iters)
    assert_identical(("primal", m), *r, "primal")
    assert r[6].iters >= min(iters, 40)


@pytest.mark.parametrize("m,n,iters", [(160, 300, 300), (300, 520, 120), (600, 1000, 40)])
def test_dense_synthetic_dual_bit_for_bit(m, n, iters):
    p1, err = eo.dual_phase1(eo.synth_problem(20260301 + m, m, n))
    v = p1.view()
    r = both(v, "dual", iters)
    assert_identical(("dual", m), *r, "dual")
    assert r[6].iters >= min(iters, 40)


@pytest.mark.parametrize("name,copies", [("afiro", 5), ("blend", 2), ("adlittle", 3)])
@pytest.mark.parametrize("which", ["primal", "dual"])
def test_netlib_replications_bit_for_bit(name, copies, which):
    """real, sparse, degenerate bases (block-diagonal copies of a netlib LP, variables and rows shuffled): both
    phases to the end, equal bits, and the optimum copies x the pinned one (tests/problems/mod.rs:657-674)"""
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
    fx = permuted_fixture(base, np.random.default_rng(7 + copies))
    phases, st = two_phases_mid(fx, which, 100000)
    assert phases == 2 and st == eo.OPTIMAL


def test_random_bound_kinds_bit_for_bit():
    """every bound kind, bound flips, free variables, infeasible / unbounded endings — blown up past 128 rows by
    replication"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_random import feasible_fixture, random_fixture
    ran = 0
    for s in range(30000, 30012):
        fx = random_fixture(np.random.default_rng(s))
        mrows = max(1, len(fx["constraints"]))
        big = blockdiag(fx, 130 // mrows + 1)
        for which in ("primal", "dual"):
            prob = eo.Problem.from_fixture(big)
            p1, err = (eo.primal_phase1 if which == "primal" else eo.dual_phase1)(prob)
            if p1 is None or err:
                continue
            v = p1.view()
            if not (128 < v.m <= 1024) or (which == "primal" and v.nN == 0):
                continue
            r = both(v, which, 3000)
            assert_identical((s, which), *r, which)
            ran += 1
    assert ran >= 8


def test_defaults_by_size():
    """pipeline = 0: up to 128 rows the exact kernel with its factors in LDS (0 launches per iteration: whole iterations
    inside one persistent launch); 129-1024 rows the CERTIFIED HYBRID on the three-launch pipeline (round 3: this
    kernel alone up to 512 rows); above, two launches per iteration; pipeline = 3 still selects this kernel up to 1024"""
    E = _E()
    from ellp_amd import synth
    for m, n, want, hybrid in ((100, 260, 0, False), (200, 400, 3, True), (512, 900, 3, True), (1024, 1500, 3, True), (1025, 1500, 2, False)):
        f = synth.primal_phase1_flat(5, m, n)
        fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                           f["B"], f["N"], f["Nb"])
        eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None))
        c = eng.counters()
        assert c["launches_per_iteration"] == want and c["hybrid"] == hybrid, (m, c)
        eng.close()
        if 128 < m <= 1024:
            eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=3))
            c = eng.counters()
            assert c["launches_per_iteration"] == 0 and not c["hybrid"], (m, c)
            eng.close()


def test_slices_and_hand_over_to_the_explicit_inverse():
    """run in slices through the resident API (the slices compose to the oracle's run), then hand the engine to the
    explicit-inverse engine (a refresh builds B^-1 from the current basis) and finish there"""
    E = _E()
    from ellp_amd import synth
    f = synth.primal_phase1_flat(11, 200, 420)

    class V:
        pass
    ov = V()
    for k, val in f.items():
        setattr(ov, k, val.copy() if hasattr(val, "copy") else val)
    ov.nB, ov.nN = len(f["B"]), len(f["N"])
    st_o, it_o, _ = eo.primal_solve_with_initial(ov, 150)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=3))
    total = 0
    for k in (1, 2, 47, 100):
        st, stats, msg = eng.run(k)
        total += k
        assert stats.iters == min(total, it_o), msg
    eng.read_point()
    np.testing.assert_array_equal(fp.B, ov.B)
    assert fp.x.tobytes() == ov.x.tobytes()
    assert eng.inverse_residual() < 1e-11
    assert eng.counters()["launches_per_iteration"] == 3
    st, stats, msg = eng.run(1 << 40)
    eng.read_point()
    eng.close()
    assert st == E.OPTIMAL and abs(fp.obj()) < 1e-9, msg
