"""Error conventions and edge cases of the C ABI on the device (the reference's Err/panic
paths at the seam: primal_simplex_solver.rs:124-151, :175-179)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def _tiny(E, A, c, x, B, N, Nb, kind=None, lb=None, ub=None):
    m, n = A.shape
    kind = np.ones(n, np.uint8) if kind is None else kind
    lb = np.zeros(n) if lb is None else lb
    ub = np.zeros(n) if ub is None else ub
    return E.FlatProblem(m, n, n, np.asfortranarray(A).reshape(-1, order="F"), c, np.zeros(m), kind, lb, ub, x, B, N, Nb)


def test_singular_basis_is_an_error():
    """`Err("invalid B, A_B is not invertible")` (primal…:175-179)."""
    E = _E()
    A = np.array([[1.0, 2.0, 1.0, 0.0], [2.0, 4.0, 0.0, 1.0]])  # columns 0 and 1 are parallel
    fp = _tiny(E, A, np.array([1.0, 1.0, 0.0, 0.0]), np.zeros(4), [0, 1], [2, 3], [0, 0])
    st, _, msg = E.primal_solve_with_initial(fp)
    assert st == E.ERR_SINGULAR and "not invertible" in msg


def test_all_columns_basic_is_optimal():
    """N empty -> Optimal without iterating (primal…:149-151)."""
    E = _E()
    A = np.eye(3)
    fp = _tiny(E, A, np.ones(3), np.ones(3), [0, 1, 2], [], [])
    st, stats, _ = E.primal_solve_with_initial(fp)
    assert st == E.OPTIMAL and stats.iters == 0


@pytest.mark.parametrize("pipeline", [0, 1])
def test_nan_in_data_is_reported(pipeline):
    """A NaN in a nonbasic column.  A documented DEVIATION, pinned on both sides: the reference's "NaN detected"
    expect() (primal…:282) is unreachable — a NaN operand fails the `>= EPS` test in front of it — so its max_by
    orders the NaN key by variable index, the column enters, the ratio test sees no finite ratio and the loop
    returns Unbounded after one iteration (the oracle does exactly that); the engine refuses to pivot on a NaN and
    returns ELLP_ERR_NAN from the same iteration."""
    from oracle import ellp_oracle as eo
    E = _E()
    A = np.array([[1.0, 1.0, 1.0, 0.0], [1.0, np.nan, 0.0, 1.0]])
    fp = _tiny(E, A, np.array([-1.0, -1.0, 0.0, 0.0]), np.array([0, 0, 1.0, 1.0]), [2, 3], [0, 1], [0, 0])

    class V:
        pass
    v = V()
    v.m, v.n, v.n_c, v.nB, v.nN = 2, 4, 4, 2, 2
    v.A, v.c, v.b = fp.A.copy(), fp.c.copy(), fp.b.copy()
    v.kind, v.lb, v.ub, v.x = fp.kind.copy(), fp.lb.copy(), fp.ub.copy(), fp.x.copy()
    v.B, v.N, v.Nb = fp.B.copy(), fp.N.copy(), fp.Nb.copy()
    st_o, it_o, _ = eo.primal_solve_with_initial(v, 100)
    assert (st_o, it_o) == (eo.UNBOUNDED, 1)
    st, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=100, pipeline=pipeline))
    assert st == E.ERR_NAN, (st, stats.iters, msg)
    assert stats.iters == (1 if pipeline == 0 else 0)  # the persistent kernel counts the loop body it is in; the
    #                                                    explicit-inverse engine counts an iteration when FTRAN enters it


def test_unbounded_direction():
    """min -x0 s.t. x0 - x1 + s = 1, x >= 0: ray along (1,1) -> Unbounded (primal…:404-406)."""
    E = _E()
    A = np.array([[1.0, -1.0, 1.0]])
    fp = _tiny(E, A, np.array([-1.0, -1.0, 0.0]), np.array([0.0, 0.0, 1.0]), [2], [0, 1], [0, 0])
    st, _, _ = E.primal_solve_with_initial(fp)
    assert st == E.UNBOUNDED


def test_bound_flip_without_basis_change():
    """An entering TwoSided variable whose own span is the tightest ratio just moves to its other
    bound (primal…:223-231): min -x0, 0 <= x0 <= 1, x0 + s = 5."""
    E = _E()
    A = np.array([[1.0, 1.0]])
    kind = np.array([3, 1], np.uint8)
    fp = _tiny(E, A, np.array([-1.0, 0.0]), np.array([0.0, 5.0]), [1], [0], [0], kind=kind,
               lb=np.zeros(2), ub=np.array([1.0, 0.0]))
    st, stats, _ = E.primal_solve_with_initial(fp)
    assert st == E.OPTIMAL
    assert stats.bound_flips == 1 and stats.pivots == 0
    np.testing.assert_allclose(fp.x, [1.0, 4.0])
    assert fp.Nb[0] == 1  # now at its upper bound


def test_slices_equal_one_run():
    """ellp_engine_run in slices (tableau resident) == one call: same iterations, basis, point."""
    E = _E()
    from ellp_amd import synth
    f = synth.primal_phase1_flat(11, 60, 140)

    def mk():
        return E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"],
                             f["x"], f["B"], f["N"], f["Nb"])
    a = mk()
    e1 = E.Engine(E.ENGINE_PRIMAL, a, E.default_opts(max_iter=None))
    st1, s1, _ = e1.run(10 ** 9)
    e1.read_point()
    e1.close()
    b = mk()
    e2 = E.Engine(E.ENGINE_PRIMAL, b, E.default_opts(max_iter=None))
    total = 0
    while True:
        st2, s2, _ = e2.run(7)
        if st2 != E.MAXITER:
            break
    e2.read_point()
    e2.close()
    assert st1 == st2 == E.OPTIMAL and s1.iters == s2.iters
    np.testing.assert_array_equal(a.B, b.B)
    np.testing.assert_array_equal(a.x, b.x)


def test_max_iter_counts_loop_bodies():
    """iter > max_iter -> MaxIter after exactly max_iter bodies (primal…:163-168)."""
    E = _E()
    from ellp_amd import synth
    f = synth.primal_phase1_flat(3, 30, 80)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"],
                       f["x"], f["B"], f["N"], f["Nb"])
    st, stats, _ = E.primal_solve_with_initial(fp, E.default_opts(max_iter=5))
    assert st == E.MAXITER and stats.iters == 5


def _unbounded_after_k(m=400, n=40, seed=3):
    """a dense LP (all rows Lte, b > 0, x >= 0) whose variable 0 has a very negative cost but a non-positive column
    EXCEPT that it only becomes the entering variable after a few ordinary pivots (its cost is the least negative):
    phase-2 arrays from the slack basis; the oracle tells at which iteration the ray is found"""
    from oracle import ellp_oracle as eo
    rng = np.random.default_rng(seed)
    A = rng.uniform(0.1, 1.1, size=(m, n))
    A[:, 0] = -rng.uniform(0.0, 1.0, size=m)          # nothing ever blocks x0
    b = A[:, 1:] @ rng.uniform(0.0, 1.0, size=n - 1) + 1.0
    c = -rng.uniform(1.0, 2.0, size=n)
    c[0] = -1e-3                                       # Dantzig takes the others first
    p = eo.Problem()
    for j in range(n):
        p.add_var(c[j], ("Lower", 0.0, 0.0))
    p.add_dense_constraints(A, "Lte", b)
    p1, err = eo.primal_phase1(p)
    v = p1.view()
    st, it, _ = eo.primal_solve_with_initial(v, 100000)
    assert st == eo.OPTIMAL and abs(v.obj()) < 1e-9
    p1.store_point(v)
    return eo, eo.primal_phase2(p1).view()


@pytest.mark.parametrize("pipeline", [2, 1])
def test_unbounded_ray_found_in_the_last_permitted_iteration(pipeline):
    """max_iter = the iteration in which the reference finds the ray: it runs max_iter FULL loop bodies
    (primal_simplex_solver.rs:162-202), so the answer is Unbounded, not MaxIter — also on the two-launch pipeline
    (m >= 384), whose slices leave the last ratio test open."""
    E = _E()
    eo, v2 = _unbounded_after_k()
    ov = v2.copy()
    st_o, it_o, _ = eo.primal_solve_with_initial(ov, 100000)
    assert st_o == eo.UNBOUNDED and it_o >= 3
    for budget, want in ((it_o, E.UNBOUNDED), (it_o - 1, E.MAXITER), (it_o + 5, E.UNBOUNDED)):
        ob = v2.copy()
        st_b, it_b, _ = eo.primal_solve_with_initial(ob, budget)
        assert st_b == want
        fp = E.FlatProblem(v2.m, v2.n, v2.n_c, v2.A, v2.c, v2.b, v2.kind, v2.lb, v2.ub, v2.x, v2.B, v2.N[:v2.nN], v2.Nb[:v2.nN])
        st, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=budget, pipeline=pipeline))
        assert st == want, (budget, st, msg)
        assert stats.iters == it_b
        np.testing.assert_array_equal(fp.B, ob.B)
        np.testing.assert_allclose(fp.x, ob.x, rtol=0, atol=1e-9 * (1 + np.abs(ob.x).max()))
        if want == E.MAXITER:
            assert abs(stats.obj - float(np.dot(ob.c, ob.x))) <= 1e-9 * (1 + abs(stats.obj))  # the objective of THAT point
    # resident engine, unlimited budget: a slice that stops right before the ray stays open; read_point completes it
    fp = E.FlatProblem(v2.m, v2.n, v2.n_c, v2.A, v2.c, v2.b, v2.kind, v2.lb, v2.ub, v2.x, v2.B, v2.N[:v2.nN], v2.Nb[:v2.nN])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=pipeline))
    st, stats, msg = eng.run(it_o)
    eng.read_point()
    # either the slice itself completed its last iteration (three launches; or the two-launch pipeline with tiny-pivot
    # maintenance, which closes every batch) or it was left open and read_point completed it
    assert st == E.UNBOUNDED or (st == E.MAXITER and eng.closing_status == E.UNBOUNDED), (st, eng.closing_status)
    eng.close()
