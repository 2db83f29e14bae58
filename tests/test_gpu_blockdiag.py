"""Real, sparse, degenerate LPs with MORE than 128 rows: block-diagonal replications of the reference's own netlib
fixtures (ADLITTLE x 3: m = 168, BLEND x 2: m = 148, ADLITTLE x 6: m = 336) in random variable / constraint orders
(the reference builds its problems by iterating HashMaps, tests/problems/mod.rs:657-674, so every order occurs),
through both solvers on the LU-per-iteration kernel (ellp_opts.pipeline = 3, ellp_mid.inc), phase by phase at the
seam.  Every order must end as the oracle does: same status, same iteration count, same basis, same bits of the point —
including the orders on which the reference's own rule ends wrongly (its absolute EPS tests on ill-conditioned
bases: about one ADLITTLE x 6 dual run in fifteen).  Round 3 ran whole solves on this kernel by default up to 512 rows;
since round 4 the default there is the certified hybrid (tests/test_gpu_hybrid.py: all 60 orders of each problem, on
results), which calls this kernel for its certificates and as its fall-back — so its bit-equality with the oracle is
what that design rests on, and it stays pinned here on a subset of the orders.  The plain explicit-inverse engine
(pipeline = 1 / 2) ends differently from the oracle on 10-40 % of these orders (tests/campaign/blockdiag_cpu.py) and is measured
at the end of this file."""
import os
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
from oracle import ellp_oracle as eo
from test_gpu_small import assert_identical, flat

pytestmark = pytest.mark.gpu

# (name, copies, orders in the routine suite): the first 12 / 12 / 6 of each problem's 60 orders (ADLITTLE x 6 takes 4 s per
# order on this kernel: 1.1 ms per iteration on the device, 0.4 on the host); tests/campaign/blockdiag_orders.py runs all
# 60 of each (profiles/r03_blockdiag_orders.json, 180 / 180)
CASES = [("adlittle", 3, 12), ("blend", 2, 12), ("adlittle", 6, 6)]


def _E():
    from ellp_amd import _engine as E
    return E


def seam_default(view, which, max_iter=200000):
    """one phase at the seam: oracle and engine (the LU-per-iteration kernel) from the same arrays; equal everything"""
    E = _E()
    ov = view.copy()
    fo = eo.primal_solve_with_initial if which == "primal" else eo.dual_solve_with_initial
    st_o, it_o, err_o = fo(ov, max_iter)
    fp = flat(view)
    fg = E.primal_solve_with_initial if which == "primal" else E.dual_solve_with_initial
    st_g, stats, err_g = fg(fp, E.default_opts(max_iter=max_iter, pipeline=3))
    assert_identical(which, ov, st_o, it_o, err_o, fp, st_g, stats, err_g, which)
    return ov, st_o


def order_case(fx, want_obj, tally):
    """both solvers on one order; tally[...] counts how the ORACLE ended (the engine ended the same way)"""
    prob = eo.Problem.from_fixture(fx)
    # primal (primal_simplex_solver.rs:32-93)
    p1, err = eo.primal_phase1(prob)
    assert p1 is not None and not err and 128 < p1.view().m <= 512
    ov, st = seam_default(p1.view(), "primal")
    ok = False
    if st == eo.OPTIMAL and abs(ov.obj()) < 1e-9:
        p1.store_point(ov)
        ov2, st2 = seam_default(eo.primal_phase2(p1).view(), "primal")
        ok = st2 == eo.OPTIMAL and abs(ov2.obj() / want_obj - 1.0) < 1e-6
    tally["primal_ok" if ok else "primal_reference_fails"] += 1
    # dual (dual_simplex_solver.rs:33-108)
    d1, err = eo.dual_phase1(prob)
    assert d1 is not None and not err
    ov, st = seam_default(d1.view(), "dual")
    ok = False
    if st == eo.OPTIMAL:
        d1.store_point(ov)
        d2, err2 = eo.dual_phase2(d1)
        if d2 is not None and not err2:
            ov2, st2 = seam_default(d2.view(), "dual")
            ok = st2 == eo.OPTIMAL and abs(ov2.obj() / want_obj - 1.0) < 1e-6
    tally["dual_ok" if ok else "dual_reference_fails"] += 1


@pytest.mark.parametrize("name,copies,orders", CASES, ids=[f"{n}x{c}" for n, c, _ in CASES])
def test_every_order_ends_as_the_oracle_does(name, copies, orders):
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
    rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
    tally = {"primal_ok": 0, "primal_reference_fails": 0, "dual_ok": 0, "dual_reference_fails": 0}
    for trial in range(orders):
        order_case(permuted_fixture(base, rng), copies * ka["obj"], tally)
    # the engine equalled the oracle on every one of them (asserted above); the oracle itself reaches the pinned
    # optimum (tests/problems/mod.rs:661-673, times the number of copies) on nearly all
    assert tally["primal_ok"] + tally["primal_reference_fails"] == orders
    assert tally["primal_ok"] >= orders - 2, tally
    assert tally["dual_ok"] >= orders - max(2, orders // 6), tally


def test_explicit_inverse_engine_on_the_same_orders_for_the_record():
    """pipeline = 1 (explicit inverse, three launches) on 12 orders of ADLITTLE x 3 through the dual: it ends at the
    optimum on most and differently from the oracle on some — the reason it is no longer the default below 513 rows.
    Nothing is asserted about how many; the test pins that the path still runs and returns a status."""
    E = _E()
    ka = next(p for p in known_answers()["netlib"] if p["name"] == "adlittle")
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), 3)
    rng = np.random.default_rng(zlib.crc32(b"adlittlex3"))
    ended = []
    for trial in range(12):
        prob = eo.Problem.from_fixture(permuted_fixture(base, rng))
        d1, err = eo.dual_phase1(prob)
        fp = flat(d1.view())
        st, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=200000, pipeline=1))
        ended.append(st)
    assert all(s in (E.OPTIMAL, E.INFEASIBLE, E.UNBOUNDED, E.MAXITER, E.ERR_SINGULAR, E.ERR_PANIC, E.ERR_NAN) for s in ended)
    print("explicit inverse, dual phase 1 of ADLITTLE x 3, 12 orders:", ended)
