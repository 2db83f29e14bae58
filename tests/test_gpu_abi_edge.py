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


def test_nan_in_data_is_reported():
    E = _E()
    A = np.array([[1.0, 1.0, 1.0, 0.0], [1.0, np.nan, 0.0, 1.0]])
    fp = _tiny(E, A, np.array([-1.0, -1.0, 0.0, 0.0]), np.array([0, 0, 1.0, 1.0]), [2, 3], [0, 1], [0, 0])
    st, _, msg = E.primal_solve_with_initial(fp)
    assert st in (E.ERR_NAN, E.ERR_SINGULAR, E.ERR_PANIC), (st, msg)


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
