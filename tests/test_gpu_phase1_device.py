"""Primal phase 1 built on the device (ellp_engine_create_primal_phase1; primal_problem.rs:236-246, SURVEY.md
§8 f2): from the standard form and the nonbasic start the engine makes b~ = b - A v, the artificial columns
signum(b~_i) e_i and their values |b~_i| itself.  Checked against the phase-1 arrays of the oracle's own
construction, then both engines (this one, and one created from the oracle's arrays) run to the same end."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def _problem(m, n, seed, mixed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-0.5, 1.1, size=(m, n))
    x0 = rng.uniform(0.2, 1.0, size=n)
    vars_ = []
    for j in range(n):
        u = rng.random() if mixed else 0.0
        if u < 0.5:
            vars_.append([float(rng.normal()), ["Lower", float(rng.integers(-1, 2)), 0.0]])
        elif u < 0.75:
            vars_.append([float(rng.normal()), ["Upper", 0.0, float(rng.integers(1, 4))]])
        else:
            lo = float(rng.integers(-2, 1))
            vars_.append([float(rng.normal()), ["TwoSided", lo, lo + 3.0]])
    cons = []
    for i in range(m):
        ax = float(A[i] @ x0)
        op = ["Lte", "Gte", "Eq"][i % 3] if mixed else "Lte"
        rhs = ax + (1.0 if op == "Lte" else (-1.0 if op == "Gte" else 0.0))
        cons.append([[[j, float(A[i, j])] for j in range(n)], op, rhs])
    return eo.Problem.from_fixture({"vars": vars_, "constraints": cons})


@pytest.mark.parametrize("m,n,mixed", [(40, 90, True), (150, 320, True), (300, 500, False)])
def test_phase1_arrays_made_on_the_device(m, n, mixed):
    E = _E()
    p1, err = eo.primal_phase1(_problem(m, n, 7 + m, mixed))
    assert p1 is not None and not err
    v = p1.view()
    ns = v.n - v.m                      # columns of the standard form (originals + slacks); the rest are artificials
    assert v.n_c == v.n                 # no free variables: primal_problem.rs:236-246 is the branch taken
    assert np.array_equal(v.B, np.arange(ns, v.n)) and np.array_equal(v.N[:v.nN], np.arange(ns))
    A = np.asarray(v.A).reshape(v.n, v.m)          # column j = A[j]
    opts = E.default_opts(max_iter=None, pipeline=1)
    eng = E.Engine.primal_phase1(v.m, ns, A[:ns].reshape(-1), v.b, v.kind[:ns], v.lb[:ns], v.ub[:ns], v.x[:ns],
                                 v.Nb[:ns], opts)
    eng.read_point()
    fp = eng.fp
    np.testing.assert_array_equal(fp.B, v.B)
    np.testing.assert_array_equal(fp.N, v.N[:v.nN])
    np.testing.assert_array_equal(fp.Nb, v.Nb[:v.nN])
    np.testing.assert_allclose(fp.x, v.x, rtol=0, atol=1e-12 * (1 + np.abs(v.x).max()))
    # the artificial columns themselves: B^-1 of the device-made basis = diag(signum(b~))
    W = eng.tap(E.TAP_BINV, v.m * v.m).reshape(v.m, v.m)
    want = np.diag(A[ns:][np.arange(v.m), np.arange(v.m)])     # oracle's artificial block is diagonal +-1
    np.testing.assert_array_equal(W, want)
    # both engines to the end of phase 1
    ref_fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN])
    ref = E.Engine(E.ENGINE_PRIMAL, ref_fp, opts)
    st_r, stats_r, _ = ref.run(1 << 40)
    ref.read_point()
    ref.close()
    st, stats, msg = eng.run(1 << 40)
    eng.read_point()
    eng.close()
    assert st == st_r, msg
    assert stats.iters == stats_r.iters
    np.testing.assert_array_equal(fp.B, ref_fp.B)
    np.testing.assert_allclose(fp.x, ref_fp.x, rtol=0, atol=1e-9 * (1 + np.abs(ref_fp.x).max()))


# ---- the dual's phase-1 point (ellp_engine_create_dual_phase1; dual_problem.rs:162-214)
@pytest.mark.parametrize("m,n,pipeline", [(150, 300, 1), (200, 500, 1), (40, 90, 0)])
def test_dual_phase1_point_made_on_the_device(m, n, pipeline):
    """y = B^-T c_B, d = c - A^T y, labels / values by the sign of d, x_B = B^-1 (b - A x) from the basis the
    LU of A^T picked: against the oracle's construction (LU solves), then both engines run phase 1.  The last
    case is a small LP: the point is made from an explicit inverse, the loop then runs in the persistent kernel."""
    E = _E()
    d1, err = eo.dual_phase1(eo.synth_problem(20260301, m, n))
    assert d1 is not None and not err
    v = d1.view()
    assert v.n_c == v.n and v.nN > 0
    opts = E.default_opts(max_iter=None, pipeline=pipeline)
    eng = E.Engine.dual_phase1(v.m, v.n, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.B, v.N[:v.nN], opts)
    eng.read_point()
    fp = eng.fp
    np.testing.assert_array_equal(fp.B, v.B)
    np.testing.assert_array_equal(fp.N, v.N[:v.nN])
    sc = 1 + max(np.abs(v.x).max(), np.abs(v.y).max(), np.abs(v.d).max())
    np.testing.assert_allclose(fp.y, v.y, rtol=0, atol=1e-10 * sc)
    np.testing.assert_allclose(fp.d, v.d, rtol=0, atol=1e-10 * sc)
    # labels: by the sign of d, which is only decided where |d| is above roundoff
    Nidx = v.N[:v.nN]
    clear = np.abs(v.d[Nidx]) > 1e-9 * sc
    np.testing.assert_array_equal(fp.Nb[clear], v.Nb[:v.nN][clear])
    if clear.all():
        np.testing.assert_allclose(fp.x, v.x, rtol=0, atol=1e-9 * sc)
    ref_fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)
    ref = E.Engine(E.ENGINE_DUAL, ref_fp, opts)
    st_r, stats_r, _ = ref.run(1 << 40)
    ref.read_point()
    ref.close()
    st, stats, msg = eng.run(1 << 40)
    eng.read_point()
    eng.close()
    assert st == st_r == E.OPTIMAL, msg
    obj = lambda f: eo_dual_obj(v, f)
    assert abs(obj(fp) - obj(ref_fp)) < 1e-8 * (1 + abs(obj(ref_fp)))
    if clear.all():
        assert stats.iters == stats_r.iters
        np.testing.assert_array_equal(fp.B, ref_fp.B)


def eo_dual_obj(v, f):
    """dual_simplex_solver.rs:184 on a FlatProblem's y and d"""
    o = float(np.dot(v.b, f.y))
    for i in range(v.n_c):
        k, di = int(v.kind[i]), float(f.d[i])
        if k == 1: o += v.lb[i] * di
        elif k == 2: o += v.ub[i] * di
        elif k == 3: o += (v.lb[i] if di > 0 else v.ub[i]) * di
        elif k == 4: o += v.lb[i] * di
    return o


def test_dual_phase1_refuses_other_bounds():
    E = _E()
    d1, _ = eo.dual_phase1(eo.synth_problem(20260301, 150, 300))
    v = d1.view()
    kind = np.array(v.kind).copy()
    kind[int(v.N[0])] = 1                     # Lower: "bounds should always be fixed or two-sided"
    with pytest.raises(E.EllpHipError) as ei:
        E.Engine.dual_phase1(v.m, v.n, v.A, v.c, v.b, kind, v.lb, v.ub, v.B, v.N[:v.nN], E.default_opts(pipeline=1))
    assert "fixed or two-sided" in str(ei.value)
