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

def flat(v):
    return _E().FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)

def seam(view, which, max_iter=2000):
    ov = view.copy()
    if which == "primal":
        st_o, it_o, err_o = eo.primal_solve_with_initial(ov, max_iter)
    else:
        st_o, it_o, err_o = eo.dual_solve_with_initial(ov, max_iter)
    fp = flat(view)
    opts = _E().default_opts(max_iter=max_iter)
    st_g, stats, err_g = (_E().primal_solve_with_initial if which == "primal" else _E().dual_solve_with_initial)(fp, opts)
    return ov, st_o, it_o, fp, st_g, stats, err_o, err_g

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



def _campaign(seed0, count):
    bad = []
    n = {"primal": 0, "dual": 0}
    for s in range(seed0, seed0 + count):
        fx = random_fixture(np.random.default_rng(s))
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
            if v1.m > 0 and v1.nN > 0:
                n["dual"] += 1
                r = seam(v1, "dual")
                ok = compare((s, "dual1"), *r, bad)
                ov = r[0]
                if ok and r[1] == 0:
                    d1.store_point(ov)
                    d2, err2 = eo.dual_phase2(d1)
                    if d2 is not None and not err2:
                        v2 = d2.view()
                        if v2.m > 0 and v2.nN > 0:
                            compare((s, "dual2"), *seam(v2, "dual"), bad)
    return n, bad


def test_random_lps_primal_exact_dual_same_objective():
    n, bad = _campaign(1000, 400)
    assert n["primal"] > 250 and n["dual"] > 250
    hard = [b for b in bad if b[1] != "path"]                       # status or point differs
    assert not hard, hard[:5]
    primal_paths = [b for b in bad if b[0][1].startswith("primal")]  # primal must be pivot for pivot
    assert not primal_paths, primal_paths[:5]
    dual_paths = [b for b in bad if b[0][1].startswith("dual")]
    assert all(b[4] < 1e-8 for b in dual_paths), dual_paths[:5]      # same objective on a tie-broken path
    assert len(dual_paths) <= 0.03 * n["dual"], (len(dual_paths), n["dual"])
