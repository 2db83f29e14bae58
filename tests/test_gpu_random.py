"""Randomized parity campaign at the seam: random small LPs with every bound kind (Lower, Upper,
TwoSided, Free, Fixed), every constraint operator, integer data (many exact ties, degenerate
pivots, bound flips, redundant rows) and real data; oracle (CPU) against engine (GPU), phase by
phase.  Primal: pivot for pivot (same iteration count, basis, labels, point).  Dual: the
reference's ratio test is an exact first-minimum (`min_by`, dual…:263-279), so a tie that differs
in the last bit of alpha_j may legitimately be broken the other way; there the status and the
objective must agree and such cases must stay rare."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def random_fixture(rng):
    m = int(rng.integers(1, 14)); n = int(rng.integers(1, 18))
    dens = rng.choice([0.3, 0.6, 1.0])
    integer = rng.random() < 0.5
    vars_ = []
    for j in range(n):
        c = float(rng.integers(-4, 5)) if integer else float(rng.normal())
        k = rng.choice(["Lower", "Lower", "Upper", "TwoSided", "Free", "Fixed"], p=[0.35, 0.15, 0.15, 0.2, 0.1, 0.05])
        lo = float(rng.integers(-3, 3)) if integer else float(rng.normal())
        hi = lo + (float(rng.integers(1, 5)) if integer else float(abs(rng.normal()) + 0.1))
        b = {"Lower": ["Lower", lo, 0.0], "Upper": ["Upper", 0.0, hi], "TwoSided": ["TwoSided", lo, hi],
             "Free": ["Free", 0.0, 0.0], "Fixed": ["Fixed", lo, lo]}[k]
        vars_.append([c, b])
    cons = []
    for i in range(m):
        coeffs = []
        for j in range(n):
            if rng.random() < dens:
                a = float(rng.integers(-3, 4)) if integer else float(rng.normal())
                if a != 0.0:
                    coeffs.append([j, a])
        op = str(rng.choice(["Lte", "Gte", "Eq"], p=[0.45, 0.35, 0.2]))
        rhs = float(rng.integers(-5, 8)) if integer else float(rng.normal() * 2)
        cons.append([coeffs, op, rhs])
    return {"vars": vars_, "constraints": cons}

def feasible_fixture(rng):
    """Feasible and bounded by construction (so both phases run to an optimum): box-bounded or fixed
    variables, right-hand sides built from a point x0 inside the bounds, with exact-tie-prone integer
    data half of the time and a few redundant (duplicated / combined) rows."""
    m = int(rng.integers(2, 16)); n = int(rng.integers(2, 20))
    integer = rng.random() < 0.5
    vars_, x0 = [], []
    for j in range(n):
        c = float(rng.integers(-4, 5)) if integer else float(rng.normal())
        lo = float(rng.integers(-3, 3)) if integer else float(rng.normal())
        w = float(rng.integers(1, 5)) if integer else float(abs(rng.normal()) + 0.1)
        if rng.random() < 0.1:
            vars_.append([c, ["Fixed", lo, lo]]); x0.append(lo)
        else:
            vars_.append([c, ["TwoSided", lo, lo + w]])
            x0.append(lo + (float(rng.integers(0, int(w) + 1)) if integer else float(rng.random() * w)))
    rows = []
    for i in range(m):
        a = np.where(rng.random(n) < 0.7, rng.integers(-3, 4, size=n).astype(float) if integer else rng.normal(size=n), 0.0)
        rows.append(a)
    if m >= 3 and rng.random() < 0.4:
        rows[m - 1] = rows[0] + rows[1]            # a redundant equality candidate
    cons = []
    for i, a in enumerate(rows):
        ax = float(np.dot(a, x0))
        op = str(rng.choice(["Lte", "Gte", "Eq"], p=[0.4, 0.3, 0.3]))
        slack = float(rng.integers(0, 4)) if integer else float(abs(rng.normal()))
        rhs = ax + slack if op == "Lte" else (ax - slack if op == "Gte" else ax)
        cons.append([[[j, float(a[j])] for j in range(n) if a[j] != 0.0], op, rhs])
    return {"vars": vars_, "constraints": cons}


def wide_fixture(rng):
    """feasible, bounded, mixed bound kinds, few rows and thousands of columns (several columns per pricing block)"""
    m = int(rng.integers(30, 80)); n = int(rng.integers(1200, 3200))
    integer = rng.random() < 0.4
    vars_, x0 = [], []
    for j in range(n):
        c = float(rng.integers(-4, 5)) if integer else float(rng.normal())
        lo = float(rng.integers(-3, 3)) if integer else float(rng.normal())
        w = float(rng.integers(1, 5)) if integer else float(abs(rng.normal()) + 0.1)
        u = rng.random()
        if u < 0.05:
            vars_.append([c, ["Fixed", lo, lo]]); x0.append(lo)
        elif u < 0.45:
            vars_.append([c, ["TwoSided", lo, lo + w]]); x0.append(lo + (float(rng.integers(0, int(w) + 1)) if integer else float(rng.random() * w)))
        elif u < 0.75:
            vars_.append([abs(c), ["Lower", lo, 0.0]]); x0.append(lo + (float(rng.integers(0, 3)) if integer else float(abs(rng.normal()))))
        else:  # no Free variables: with dozens of them the reference's phase-1 setup returns None / panics
            vars_.append([-abs(c), ["Upper", 0.0, lo]]); x0.append(lo - (float(rng.integers(0, 3)) if integer else float(abs(rng.normal()))))
    cons = []
    dens = rng.choice([0.05, 0.3, 1.0])
    for i in range(m):
        a = np.where(rng.random(n) < dens, rng.integers(-3, 4, size=n).astype(float) if integer else rng.normal(size=n), 0.0)
        ax = float(np.dot(a, x0))
        op = str(rng.choice(["Lte", "Gte", "Eq"], p=[0.45, 0.4, 0.15]))
        slack = float(rng.integers(0, 4)) if integer else float(abs(rng.normal()))
        rhs = ax + slack if op == "Lte" else (ax - slack if op == "Gte" else ax)
        cons.append([[[j, float(a[j])] for j in range(n) if a[j] != 0.0], op, rhs])
    return {"vars": vars_, "constraints": cons}


def flat(v):
    return _E().FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)

SEAM_MAX_ITER = [2000]
# ellp_opts.pipeline for the seam calls: 0 = the engine as a user gets it (m <= 128: the persistent
# exact kernel of ellp_small.inc), 1 = the explicit-inverse engine with its default maintenance
SEAM_PIPELINE = [0]


def seam(view, which, max_iter=None):
    max_iter = SEAM_MAX_ITER[0] if max_iter is None else max_iter
    ov = view.copy()
    if which == "primal":
        st_o, it_o, err_o = eo.primal_solve_with_initial(ov, max_iter)
    else:
        st_o, it_o, err_o = eo.dual_solve_with_initial(ov, max_iter)
    fp = flat(view)
    opts = _E().default_opts(max_iter=max_iter, pipeline=SEAM_PIPELINE[0])
    st_g, stats, err_g = (_E().primal_solve_with_initial if which == "primal" else _E().dual_solve_with_initial)(fp, opts)
    return ov, st_o, it_o, fp, st_g, stats, err_o, err_g


@pytest.fixture(params=[0, 1, 2], ids=["default-path", "explicit-inverse", "two-launch"])
def pipeline(request):
    SEAM_PIPELINE[0] = request.param
    yield request.param
    SEAM_PIPELINE[0] = 0

def compare(tag, ov, st_o, it_o, fp, st_g, stats, err_o, err_g, out):
    if st_o != st_g:
        out.append((tag, "status", st_o, st_g, err_o, err_g)); return False
    if st_o in (0, 3):
        if stats.iters != it_o or not np.array_equal(fp.B, ov.B) or not np.array_equal(fp.N[:fp.nN], ov.N[:ov.nN]) or not np.array_equal(fp.Nb[:fp.nN], ov.Nb[:ov.nN]):
            dobj = abs(fp.obj() - ov.obj())
            out.append((tag, "path", it_o, int(stats.iters), dobj)); return dobj < 1e-8 * (1 + abs(ov.obj()))
        sc = 1 + (np.abs(ov.x).max() if ov.x.size else 0)
        if np.abs(fp.x - ov.x).max() > 1e-8 * sc:
            out.append((tag, "x", float(np.abs(fp.x - ov.x).max()))); return False
    return True



def _campaign(seed0, count, make=None):
    make = make or random_fixture
    bad = []
    n = {"primal": 0, "dual": 0}
    for s in range(seed0, seed0 + count):
        fx = make(np.random.default_rng(s))
        prob = eo.Problem.from_fixture(fx)
        p1, err = eo.primal_phase1(prob)
        if p1 is not None and not err:
            v1 = p1.view()
            if v1.m > 0 and v1.nN > 0:
                n["primal"] += 1
                r = seam(v1, "primal")
                ok = compare((s, "primal1"), *r, bad)
                ov = r[0]
                if ok and r[1] == 0 and abs(ov.obj()) < 1e-10:
                    p1.store_point(ov)
                    p2 = eo.primal_phase2(p1)
                    compare((s, "primal2"), *seam(p2.view(), "primal"), bad)
        d1, err = eo.dual_phase1(prob)
        if d1 is not None and not err:
            v1 = d1.view()
            if v1.m > 0:  # nN == 0 (every variable boxed): Optimal at once (dual…:175-177), then phase 2
                n["dual"] += 1
                r = seam(v1, "dual")
                ok = compare((s, "dual1"), *r, bad)
                ov = r[0]
                if ok and r[1] == 0:
                    d1.store_point(ov)
                    d2, err2 = eo.dual_phase2(d1)
                    if d2 is not None and not err2:
                        v2 = d2.view()
                        if v2.m > 0:
                            compare((s, "dual2"), *seam(v2, "dual"), bad)
    return n, bad


def test_random_lps_primal_exact_dual_same_objective(pipeline):
    n, bad = _campaign(1000, 400)
    if pipeline == 0:  # the exact kernel: every run is the oracle's, pivot for pivot, dual ties included
        assert not bad, bad[:5]
    assert n["primal"] > 250 and n["dual"] > 250
    hard = [b for b in bad if b[1] != "path"]                       # status or point differs
    assert not hard, hard[:5]
    primal_paths = [b for b in bad if b[0][1].startswith("primal")]  # primal must be pivot for pivot
    assert not primal_paths, primal_paths[:5]
    dual_paths = [b for b in bad if b[0][1].startswith("dual")]
    assert all(b[4] < 1e-8 for b in dual_paths), dual_paths[:5]      # same objective on a tie-broken path
    assert len(dual_paths) <= 0.03 * n["dual"], (len(dual_paths), n["dual"])


def test_random_feasible_bounded_lps_both_phases(pipeline):
    """The same checks on LPs that are feasible and bounded by construction, so that phase 2 runs too
    (most purely random LPs end infeasible or unbounded in phase 1)."""
    n, bad = _campaign(7000, 300, feasible_fixture)
    if pipeline == 0:  # the exact kernel shares even the reference's rounding-fragile panics (quirk Q1)
        assert not bad, bad[:5]
    assert n["primal"] > 250 and n["dual"] > 200
    # Quirk Q1 (primal…:359 + assert :402) makes the reference itself rounding-fragile on box-bounded
    # LPs: a TwoSided basic that sits ON its lower bound up to the last bit gives lambda_i = (lb - x)/d
    # = -1e-16/|d| when x is one ulp below, and the reference then panics "lambda >= 0."; with x one
    # ulp the other way it carries on.  Oracle and engine round the FTRAN differently (LU solve vs
    # B^-1 a_q), so they can fall on different sides (seeds 7140, 7184: x differs by 5e-16).  Such
    # cases are set aside and must stay rare; everything else must agree as above.
    fragile = [b for b in bad if b[1] == "status" and ("lambda >= 0" in str(b[4]) or "lambda >= 0" in str(b[5]))]
    assert len(fragile) <= 0.03 * n["primal"], fragile
    bad = [b for b in bad if b not in fragile]
    hard = [b for b in bad if b[1] != "path"]
    assert not hard, hard[:5]
    primal_paths = [b for b in bad if b[0][1].startswith("primal")]
    assert not primal_paths, primal_paths[:5]
    dual_paths = [b for b in bad if b[0][1].startswith("dual")]
    assert all(b[4] < 1e-8 for b in dual_paths), dual_paths[:5]
    assert len(dual_paths) <= 0.05 * n["dual"], (len(dual_paths), n["dual"])


def test_random_lps_end_to_end_through_the_host_mirror():
    """solver.solve(problem) (C++ host mirror + GPU loops, resident two-phase primal) against the oracle's
    solve() on 300 random and 200 feasible LPs, both solvers: same outcome (a reference panic shows up
    as an exception), same objective."""
    from ellp_amd import DualSimplexSolver, PrimalSimplexSolver, Problem
    cases = [random_fixture(np.random.default_rng(s)) for s in range(5000, 5300)]
    cases += [feasible_fixture(np.random.default_rng(s)) for s in range(9000, 9200)]
    n_opt = n_fragile = 0
    for k, fx in enumerate(cases):
        for name, S in (("primal", PrimalSimplexSolver), ("dual", DualSimplexSolver)):
            o = eo.solve(eo.Problem.from_fixture(fx), name)
            try:
                r = S.default().solve(Problem.from_fixture(fx))
                got = r.kind
            except Exception as e:  # EllPError / panic of the reference
                got = "raised"
            if o.status < 0:
                if "lambda >= 0" in o.err:  # quirk Q1 fragility, see above: either outcome is the reference's
                    n_fragile += 1
                    continue
                assert got == "raised", (k, name, o.status, o.err, got)
                continue
            if got == "raised" and name == "primal":
                n_fragile += 1  # the engine fell on the panicking side of the same quirk
                continue
            assert got == eo.STATUS_NAME[o.status], (k, name, eo.STATUS_NAME[o.status], got)
            if got == "optimal":
                n_opt += 1
                assert abs(r.solution.obj() - o.obj) <= 1e-8 * (1 + abs(o.obj)), (k, name, r.solution.obj(), o.obj)
    assert n_opt > 120  # the reference reports many of the constructed LPs infeasible (quirk Q7) or panics in dual phase 2
    assert n_fragile <= 15, n_fragile


def test_random_wide_lps_first_400_pivots(pipeline):
    """Few rows, thousands of columns of every bound kind: several columns per pricing block, bound
    flips, Fixed and TwoSided entering variables, reduced costs up to 1e7 (near-singular bases).  The
    reference's rules are not sound on such LPs (quirk Q1 leaves nonbasic variables strictly inside
    their bounds, its "optimal" points violate bounds, HiGHS finds better objectives) and the rounding
    differences between two implementations grow from 1e-14 to 1e-5 over 2000 pivots on IDENTICAL
    bases, so only the first 400 pivots of each primal phase are compared — pivot for pivot.  This is
    the campaign that caught the absorbed `key > M - 4 EPS` test (premature Optimal at pivot 162 of
    seed 322)."""
    SEAM_MAX_ITER[0] = 400
    try:
        n, bad = _campaign(300, 16, wide_fixture)
    finally:
        SEAM_MAX_ITER[0] = 2000
    assert n["primal"] >= 14
    primal = [b for b in bad if b[0][1].startswith("primal") and not (b[1] == "status" and "lambda >= 0" in str(b[4]) + str(b[5]))]
    assert not primal, primal[:5]


def _permuted(fx, rng):
    """The same LP with its variables and constraints in another order (the reference builds its
    problems by iterating HashMaps, tests/problems/mod.rs:657-674, so every order occurs)."""
    n = len(fx["vars"])
    perm = rng.permutation(n)            # new index k holds old variable perm[k]
    inv = np.empty(n, dtype=int)
    inv[perm] = np.arange(n)
    rows = [fx["constraints"][i] for i in rng.permutation(len(fx["constraints"]))]
    return {"vars": [fx["vars"][j] for j in perm],
            "constraints": [[[[int(inv[j]), a] for j, a in coeffs], op, rhs] for coeffs, op, rhs in rows]}


def netlib_order_case(base, ka, fx, trial, bad, ref_errors=None):
    """One variable/constraint order of a netlib LP: primal and dual, phase by phase at the seam,
    oracle against engine (compare(): status, pivots, point); both must end at the pinned optimum
    (tests/problems/mod.rs:658-672).  The reference itself does not survive every order: its primal
    guard is ABSOLUTE (any |U_ii| < 1e-10 is Err("A_B is not invertible"), primal…:175-179) and BLEND
    has coefficients of 1e-4, so about one order in a hundred ends in that error in the oracle — and
    must end in the same error, at the same iteration, in the engine.  Such orders are appended to
    `ref_errors` (or fail the case when that is None)."""
    prob = eo.Problem.from_fixture(fx)
    p1, err = eo.primal_phase1(prob)
    r = seam(p1.view(), "primal", 5000)
    compare((trial, "primal1"), *r, bad)
    primal_ok = True
    if r[1] < 0 and ref_errors is not None:
        assert r[4] == r[1] and r[5].iters == r[2], (trial, "primal1", r[1], r[4], r[2], r[5].iters, r[7])
        ref_errors.append((trial, "primal1", r[1], r[6]))
        primal_ok = False
    else:
        assert r[1] == r[4] == eo.OPTIMAL and abs(r[0].obj()) < 1e-10 and abs(r[3].obj()) < 1e-9, (trial, r[1], r[4], r[7])
    if primal_ok:
        p1.store_point(r[0])
        r2 = seam(eo.primal_phase2(p1).view(), "primal", 5000)
        compare((trial, "primal2"), *r2, bad)
        if r2[1] < 0 and ref_errors is not None:
            assert r2[4] == r2[1] and r2[5].iters == r2[2], (trial, "primal2", r2[1], r2[4], r2[7])
            ref_errors.append((trial, "primal2", r2[1], r2[6]))
        else:
            assert r2[1] == r2[4] == eo.OPTIMAL and abs(r2[3].obj() / ka["obj"] - 1.0) < 1e-6, (trial, r2[1], r2[4], r2[7])
    # dual, at the seam
    d1, err = eo.dual_phase1(prob)
    rd = seam(d1.view(), "dual", 20000)
    compare((trial, "dual1"), *rd, bad)
    assert rd[1] == rd[4] == eo.OPTIMAL, (trial, "dual1", rd[1], rd[4], rd[7])
    d1.store_point(rd[0])
    d2, err2 = eo.dual_phase2(d1)
    assert d2 is not None and not err2
    rd2 = seam(d2.view(), "dual", 20000)
    compare((trial, "dual2"), *rd2, bad)
    assert rd2[1] == rd2[4] == eo.OPTIMAL, (trial, "dual2", rd2[1], rd2[4], rd2[7])
    assert abs(rd2[3].obj() / ka["obj"] - 1.0) < 1e-6 and abs(rd2[0].obj() / ka["obj"] - 1.0) < 1e-6, (trial, rd2[3].obj())


NETLIB_ORDERS = 60


@pytest.mark.parametrize("name", ["afiro", "adlittle", "blend"])
def test_netlib_in_random_orders_explicit_inverse(name, monkeypatch):
    """The explicit-inverse engine (pipeline 1) on 6 orders per problem, as in round 1.  It is NOT
    what small LPs run on by default any more: ADLITTLE's dual passes through bases of condition
    1e7-1e9, where an explicit inverse is only good to 1e-7 however it is maintained, and about 2 % of
    its variable orders end wrongly (tests/golden/netlib_orders.json)."""
    import os
    import zlib
    from helpers import GOLDEN, known_answers, read_mps
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = read_mps(os.path.join(GOLDEN, ka["file"]))
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    SEAM_PIPELINE[0] = 1
    try:
        bad = []
        for trial in range(6):
            netlib_order_case(base, ka, _permuted(base, rng), trial, bad)
    finally:
        SEAM_PIPELINE[0] = 0
    # pivot for pivot except where a degenerate tie falls the other way (same objective then)
    assert all(b[1] == "path" and b[4] < 1e-8 * (1 + abs(ka["obj"])) for b in bad), bad
    assert len([b for b in bad if b[0][1].startswith("primal")]) <= 3, bad


@pytest.mark.parametrize("name", ["afiro", "adlittle", "blend"])
def test_netlib_in_random_orders(name):
    """Real, sparse, degenerate LPs in 60 random variable/constraint orders each (the size of the
    campaign that found the round-1 dual failures; the reference iterates HashMaps, so every order
    occurs): primal phase by phase pivot for pivot against the oracle, and the pinned optimum at the
    end, for the primal loop and for the dual loop (objective only: exact-minimum ties)."""
    import os
    from helpers import GOLDEN, known_answers, read_mps
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = read_mps(os.path.join(GOLDEN, ka["file"]))
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    bad, ref_errors = [], []
    for trial in range(NETLIB_ORDERS):
        fx = _permuted(base, rng)
        netlib_order_case(base, ka, fx, trial, bad, ref_errors)
    assert not bad, bad[:5]  # the exact kernel: the oracle's path in every order, dual ties included
    assert len(ref_errors) <= 2, ref_errors  # orders the reference's own absolute guard rejects


def _regression_orders():
    import json
    import os
    from helpers import GOLDEN
    path = os.path.join(GOLDEN, "netlib_orders.json")
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return json.load(f)["orders"]


@pytest.mark.parametrize("case", _regression_orders(), ids=lambda c: f"{c['name']}-{c['tag']}")
def test_netlib_regression_orders(case):
    """Variable/constraint orders of the netlib LPs on which an earlier engine ended wrongly
    (tests/golden/netlib_orders.json: the permutations themselves, found by tests/campaign/netlib_orders.py);
    each must now end at the pinned optimum through both loops."""
    import os
    from helpers import GOLDEN, known_answers, read_mps
    ka = next(p for p in known_answers()["netlib"] if p["name"] == case["name"])
    base = read_mps(os.path.join(GOLDEN, ka["file"]))
    perm = case["var_perm"]
    inv = np.empty(len(perm), dtype=int)
    inv[np.asarray(perm)] = np.arange(len(perm))
    rows = [base["constraints"][i] for i in case["row_perm"]]
    fx = {"vars": [base["vars"][j] for j in perm],
          "constraints": [[[[int(inv[j]), a] for j, a in coeffs], op, rhs] for coeffs, op, rhs in rows]}
    bad = []
    netlib_order_case(base, ka, fx, case["tag"], bad)
    assert not bad, bad
