"""Parity at BASELINE.json's full sizes (configs 3 and 4: m=2000, n=5000).

The oracle re-factorises every iteration, so only a short window is affordable: from the same
start, after W iterations, the engine's (B, N, x[, y, d]) must equal the oracle's — same
pivots.  Beyond the window, size-independent invariants of the simplex loop are checked:
A x = b is preserved by every pivot, the phase-1 objective never increases, the point stays
within its bounds, and W A_B = I."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu

M, N_STRUCT, SEED = 2000, 5000, 20260301


def _view(f):
    class V:
        pass
    v = V()
    for k, val in f.items():
        setattr(v, k, val.copy() if hasattr(val, "copy") else val)
    v.nB, v.nN = len(f["B"]), len(f["N"])
    return v


@pytest.fixture(scope="module")
def primal_flat():
    from ellp_amd import synth
    return synth.primal_phase1_flat(SEED, M, N_STRUCT)


@pytest.fixture(scope="module")
def dual_flat():
    from ellp_amd import synth
    return synth.dual_start_flat(SEED, M, N_STRUCT)


def test_c3_primal_window_parity(primal_flat):
    from ellp_amd import _engine as E
    W = 40
    f = primal_flat
    ov = _view(f)
    st_o, it_o, _ = eo.primal_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    st_g, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, msg
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_array_equal(fp.Nb, ov.Nb)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))


def test_c3_primal_long_window_parity(primal_flat):
    """3000 pivots (500 on a slow host) at full size: the engine's basis equals that of the explicit-B^-1 CPU loop
    (same pivot rules as the oracle, checked against it in tests/test_oracle_binv.py; the
    LU-per-iteration loop itself would need ~40 minutes for this window)."""
    from ellp_amd import _engine as E
    f = primal_flat
    _, it_p, _, secs_p = eo.primal_binv_solve_with_initial(_view(f), 40)  # probe the host's rate
    W = 3000 if it_p / max(secs_p, 1e-9) > 100.0 else 500               # keep the CPU side under ~30 s
    ov = _view(f)
    st_o, it_o, msg_o, _ = eo.primal_binv_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    st_g, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, (msg, msg_o)
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_array_equal(fp.Nb, ov.Nb)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))


def test_c4_dual_window_parity(dual_flat):
    from ellp_amd import _engine as E
    W = 40
    f = dual_flat
    ov = _view(f)
    st_o, it_o, _ = eo.dual_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"], f["y"], f["d"])
    st_g, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, msg
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))
    np.testing.assert_allclose(fp.y, ov.y, rtol=0, atol=1e-9 * (1 + np.abs(ov.y).max()))
    np.testing.assert_allclose(fp.d, ov.d, rtol=0, atol=1e-9 * (1 + np.abs(ov.d).max()))


def test_c4_dual_long_window_parity(dual_flat):
    """3000 dual pivots (500 on a slow host) at full size against the explicit-B^-1 CPU dual loop (same
    rules as the oracle's, checked against it in tests/test_oracle_binv.py)."""
    from ellp_amd import _engine as E
    f = dual_flat
    _, it_p, _, secs_p = eo.dual_binv_solve_with_initial(_view(f), 40)
    W = 3000 if it_p / max(secs_p, 1e-9) > 100.0 else 500
    ov = _view(f)
    st_o, it_o, msg_o, _ = eo.dual_binv_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"], f["y"], f["d"])
    st_g, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, (msg, msg_o)
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_array_equal(fp.Nb, ov.Nb)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))
    np.testing.assert_allclose(fp.d, ov.d, rtol=0, atol=1e-8 * (1 + np.abs(ov.d).max()))
    np.testing.assert_allclose(fp.y, ov.y, rtol=0, atol=1e-8 * (1 + np.abs(ov.y).max()))


def test_c3_primal_invariants_over_a_long_run(primal_flat):
    """3000 iterations in slices: A x = b, objective non-increasing, bounds, W A_B = I."""
    from ellp_amd import _engine as E
    f = primal_flat
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    A = f["A"].reshape((f["n"], f["m"])).T
    b = f["b"]
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None))
    prev_obj = fp.obj()
    scale = 1.0 + np.abs(b).max()
    for _ in range(6):
        st, stats, msg = eng.run(500)
        assert st == E.MAXITER, msg
        eng.read_point()
        assert np.max(np.abs(A @ fp.x - b)) < 1e-8 * scale          # every pivot keeps A x = b
        assert fp.x.min() > -1e-8 * scale                            # all variables are Lower(0)
        obj = fp.obj()
        assert obj <= prev_obj + 1e-8 * scale                        # phase-1 objective never increases
        prev_obj = obj
        assert sorted(np.concatenate([fp.B, fp.N]).tolist()) == list(range(f["n"]))  # B u N is a partition
    assert eng.inverse_residual() < 1e-10
    eng.close()


def test_c4_dual_invariants_over_a_long_run(dual_flat):
    """Dual loop: dual feasibility of the nonbasics (d_j >= -eps at Lower), y consistent with d,
    dual objective non-decreasing."""
    from ellp_amd import _engine as E
    f = dual_flat
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"], f["y"], f["d"])
    A = f["A"].reshape((f["n"], f["m"])).T
    eng = E.Engine(E.ENGINE_DUAL, fp, E.default_opts(max_iter=None))
    prev = -np.inf
    for _ in range(4):
        st, stats, msg = eng.run(500)
        assert st == E.MAXITER, msg
        eng.read_point()
        assert fp.d[fp.N].min() > -1e-8                               # nonbasics stay dual feasible
        np.testing.assert_allclose(fp.d, f["c"] - A.T @ fp.y, rtol=0, atol=1e-8)   # d = c - A^T y
        assert np.max(np.abs(fp.d[fp.B])) < 1e-8                     # basics have zero reduced cost
        dual_obj = float(np.dot(f["b"], fp.y))                        # bounds are all Lower(0)
        assert dual_obj >= prev - 1e-8
        assert abs(dual_obj - stats.obj) < 1e-6 * (1 + abs(dual_obj))  # incremental obj (dual…:316) tracks it
        prev = dual_obj
    eng.close()


# ---- m >= 3072: k_update2 takes 8 rows per block (second group streamed) and k_ftran2 streams its
# rows (no register prefetch) — the geometry of config 5 (m = 4000), at a size the checkers can afford
M_BIG, N_BIG = 3104, 3500


def test_big_m_primal_window_parity():
    from ellp_amd import _engine as E
    from ellp_amd import synth
    f = synth.primal_phase1_flat(SEED, M_BIG, N_BIG)
    W = 400
    ov = _view(f)
    st_o, it_o, msg_o, _ = eo.primal_binv_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    st_g, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, (msg, msg_o)
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_array_equal(fp.Nb, ov.Nb)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))


def test_big_m_dual_window_parity():
    from ellp_amd import _engine as E
    from ellp_amd import synth
    f = synth.dual_start_flat(SEED, M_BIG, N_BIG)
    W = 6
    ov = _view(f)
    st_o, it_o, _ = eo.dual_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"], f["y"], f["d"])
    eng = E.Engine(E.ENGINE_DUAL, fp, E.default_opts(max_iter=None))
    st_g, stats, msg = eng.run(W)
    eng.read_point()
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, msg
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))
    np.testing.assert_allclose(fp.d, ov.d, rtol=0, atol=1e-9 * (1 + np.abs(ov.d).max()))
    # and 300 more pivots keep B^-1 an inverse (every row block, both row groups, was rewritten)
    st_g, stats, msg = eng.run(300)
    assert st_g == E.MAXITER, msg
    assert eng.inverse_residual() < 1e-10
    eng.close()


def test_c5_primal_window_parity():
    """Config 5 itself (m=4000, n=40000: 1.4 GB of nonbasic columns streamed with non-temporal loads from
    2048 pricing blocks, 8-row update blocks, streamed FTRAN rows): the first 400 pivots (150 on a slow
    host) equal those of the explicit-B^-1 CPU loop."""
    from ellp_amd import _engine as E
    from ellp_amd import synth
    f = synth.primal_phase1_flat(20260305, 4000, 40000)
    _, it_p, _, secs_p = eo.primal_binv_solve_with_initial(_view(f), 10)
    W = 400 if it_p / max(secs_p, 1e-9) > 30.0 else 150
    ov = _view(f)
    st_o, it_o, msg_o, _ = eo.primal_binv_solve_with_initial(ov, W)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    del f
    st_g, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, (msg, msg_o)
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N)
    np.testing.assert_array_equal(fp.Nb, ov.Nb)
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))


# ---- config 4 on the reference's OWN dual phase-1 arrays (dual_problem.rs:89-256) at full size: the box
# problem of the config-3 LP (every variable TwoSided [0,1], rows `= 0`), basis from the LU of A^T,
# nonbasics at lower / upper by the sign of d — built by the oracle's setup (its two rank-check QRs and
# the LU share their independent columns out over the host cores, bit for bit the one-thread result).
# The covering-LP windows above start from a slack basis with Lower labels only; these run the dual loop
# with TwoSided basics leaving at either bound and Upper-labelled nonbasics entering, at m = 2000.
@pytest.fixture(scope="module")
def dual_phase1_c3():
    eo.set_setup_threads(eo.host_threads())
    try:
        d1, err = eo.dual_phase1(eo.synth_problem(SEED, M, N_STRUCT))
    finally:
        eo.set_setup_threads(1)
    assert d1 is not None and not err
    v = d1.view()
    assert v.m == M and v.nN == N_STRUCT
    assert set(np.unique(v.kind)) == {3}                       # all TwoSided
    nb = np.bincount(v.Nb[:v.nN], minlength=3)
    assert nb[0] > 1000 and nb[1] > 1000                       # nonbasics at lower AND at upper
    return v


def _dflat(v):
    from ellp_amd import _engine as E
    return E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)


def test_c4_reference_dual_phase1_window_parity(dual_phase1_c3):
    """40 pivots against the LU-per-iteration oracle (0.7 s per pivot at this size)"""
    from ellp_amd import _engine as E
    W = 40
    v = dual_phase1_c3
    ov = v.copy()
    st_o, it_o, _ = eo.dual_solve_with_initial(ov, W)
    fp = _dflat(v)
    st_g, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, msg
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N[:ov.nN])
    np.testing.assert_array_equal(fp.Nb, ov.Nb[:ov.nN])
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))
    np.testing.assert_allclose(fp.y, ov.y, rtol=0, atol=1e-9 * (1 + np.abs(ov.y).max()))
    np.testing.assert_allclose(fp.d, ov.d, rtol=0, atol=1e-9 * (1 + np.abs(ov.d).max()))


def test_c4_reference_dual_phase1_long_window_parity(dual_phase1_c3):
    """1500 pivots against the explicit-B^-1 CPU dual loop (same rules, checked against the LU oracle in
    tests/test_oracle_binv.py): same basis, same labels — leaving sides Lower and Upper both occur"""
    from ellp_amd import _engine as E
    W = 1500
    v = dual_phase1_c3
    ov = v.copy()
    st_o, it_o, msg_o, _ = eo.dual_binv_solve_with_initial(ov, W)
    fp = _dflat(v)
    st_g, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=W))
    assert st_g == st_o == E.MAXITER and stats.iters == it_o == W, (msg, msg_o)
    np.testing.assert_array_equal(fp.B, ov.B)
    np.testing.assert_array_equal(fp.N, ov.N[:ov.nN])
    np.testing.assert_array_equal(fp.Nb, ov.Nb[:ov.nN])
    changed = fp.Nb != v.Nb[:v.nN]
    assert changed.sum() > 50                                   # labels did move
    np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-8 * (1 + np.abs(ov.x).max()))
    np.testing.assert_allclose(fp.d, ov.d, rtol=0, atol=1e-8 * (1 + np.abs(ov.d).max()))
    np.testing.assert_allclose(fp.y, ov.y, rtol=0, atol=1e-8 * (1 + np.abs(ov.y).max()))


def test_c4_dual_phase1_built_on_the_device_at_full_size(dual_phase1_c3):
    """SURVEY.md §8 f2 at config 3's size: the LU of A^T that picks the basis (ellp_hip_lu_transposed) gives the
    oracle's B and N exactly (bitwise the same pivots), and the point made from the resident B^-1
    (ellp_engine_create_dual_phase1: y, d, labels by the sign of d, x_B) is the oracle's to rounding."""
    from ellp_amd import _engine as E
    v = dual_phase1_c3
    A = np.asarray(v.A).reshape(v.n, v.m).T                    # m x n, column j = variable j
    piv, ud = E.lu_transposed(A)
    perm = np.arange(v.n)
    for i, p in enumerate(piv):                                 # PermutationSequence::permute_rows
        perm[i], perm[p] = perm[p], perm[i]
    np.testing.assert_array_equal(perm[:v.m], v.B)
    np.testing.assert_array_equal(perm[v.m:], v.N[:v.nN])
    assert np.abs(ud).min() >= 1e-10
    eng = E.Engine.dual_phase1(v.m, v.n, v.A, v.c, v.b, v.kind, v.lb, v.ub, perm[:v.m], perm[v.m:], E.default_opts(max_iter=None))
    eng.read_point()
    fp = eng.fp
    eng.close()
    sc = 1 + max(np.abs(v.x).max(), np.abs(v.y).max(), np.abs(v.d).max())
    np.testing.assert_allclose(fp.y, v.y, rtol=0, atol=1e-9 * sc)
    np.testing.assert_allclose(fp.d, v.d, rtol=0, atol=1e-9 * sc)
    clear = np.abs(v.d[v.N[:v.nN]]) > 1e-8 * sc                # the sign of d decides the label only where d is not roundoff
    np.testing.assert_array_equal(fp.Nb[clear], v.Nb[:v.nN][clear])
    assert clear.mean() > 0.99
    if clear.all():
        np.testing.assert_allclose(fp.x, v.x, rtol=0, atol=1e-8 * sc)
