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
