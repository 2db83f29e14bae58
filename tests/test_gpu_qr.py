"""the standard form's rank check (`A.transpose().col_piv_qr()`,
standard_form.rs:142) on the device, in both of its modes (ellp_qr.hip).  EXACT (ELLP_QR_EXACT=1): bitwise the host
loop's pivots and |R_ii|.  FAST (the default): the same steps with parallel reductions — the same pivots up to the
rank, |R_ii| to rounding.  Either way the standard form, and everything downstream, must be identical whichever side
computed the QR."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, known_answers

pytestmark = pytest.mark.gpu
KA = known_answers()


@pytest.fixture(autouse=True, params=["exact", "fast", "default"])
def qr_mode(request):
    """exact: ELLP_QR_EXACT=1; fast: =0 (no fall-back: the fast kernels alone); default: unset — fast, done again in the exact
    mode when two pivot candidates come within 1e-12 of each other"""
    old = os.environ.get("ELLP_QR_EXACT")
    if request.param == "default":
        os.environ.pop("ELLP_QR_EXACT", None)
    else:
        os.environ["ELLP_QR_EXACT"] = "1" if request.param == "exact" else "0"
    yield request.param
    if old is None:
        os.environ.pop("ELLP_QR_EXACT", None)
    else:
        os.environ["ELLP_QR_EXACT"] = old


def host_col_piv_qr_of_transpose(A):
    """Plain restatement of ellp_amd/csrc/host/dense.h ColPivQR (= oracle col_piv_qr) on M = A^T,
    every sum sequential in the same order."""
    M = np.array(A.T, dtype=np.float64, order="F")
    rows, cols = M.shape
    mn = min(rows, cols)
    piv, rd = [], []
    for i in range(mn):
        pj, best = i, abs(M[i, i])
        for j in range(i, cols):
            for r in range(i, rows):
                v = abs(M[r, j])
                if v > best:
                    best, pj = v, j
        if pj != i:
            M[:, [i, pj]] = M[:, [pj, i]]
        piv.append(pj)
        x = M[i:, i]
        sqn = 0.0
        for v in x:
            sqn += v * v
        norm = np.sqrt(sqn)
        signed_norm = -norm if x[0] < 0.0 else norm
        factor = (sqn + abs(x[0]) * norm) * 2.0
        rd.append(norm)
        x[0] += signed_norm
        if factor == 0.0:
            continue
        sf = np.sqrt(factor)
        n2 = 0.0
        for r in range(len(x)):
            x[r] /= sf
            n2 += x[r] * x[r]
        n2 = np.sqrt(n2)
        if n2 != 0.0:
            for r in range(len(x)):
                x[r] /= n2
        for j in range(i + 1, cols):
            cj = M[i:, j]
            dot = 0.0
            for r in range(len(x)):
                dot += x[r] * cj[r]
            f2 = -2.0 * dot
            for r in range(len(x)):
                cj[r] = f2 * x[r] + cj[r]
    return np.array(piv, dtype=np.int64), np.array(rd)


def _cases():
    rng = np.random.default_rng(11)
    yield "wide", rng.uniform(-1, 1, size=(7, 19))
    yield "tall", rng.uniform(-1, 1, size=(15, 6))          # more rows than variables
    yield "square", rng.uniform(-1, 1, size=(9, 9))
    A = rng.integers(-2, 3, size=(8, 14)).astype(float)      # many exact ties
    yield "ties", A
    A = rng.uniform(-1, 1, size=(8, 12))
    A[5] = A[2]                                               # a redundant row
    A[6] = 2.0 * A[1] - A[3]
    yield "rank-deficient", A
    A = rng.uniform(-1, 1, size=(6, 10))
    A[3] = 0.0                                                # a zero row: factor == 0 at the end
    A[4] = 0.0
    yield "zero-rows", A
    A = rng.uniform(-1, 1, size=(12, 40))
    A[2] *= 10.0                                              # the two largest candidates of step 0 differ in the last bits:
    A[7] = A[2, ::-1] * (1.0 + 3e-15)                         # which of them pivots first is decided by a 1e-15 difference
    yield "near-ties", A
    yield "one-row", rng.uniform(-1, 1, size=(1, 5))
    yield "one-col", rng.uniform(-1, 1, size=(5, 1))


@pytest.mark.parametrize("name,A", list(_cases()), ids=[n for n, _ in _cases()])
def test_device_qr_is_the_host_loop(name, A, qr_mode):
    from ellp_amd import _engine as E
    piv_d, rd_d = E.qr_transposed(A)
    piv_h, rd_h = host_col_piv_qr_of_transpose(A)
    if qr_mode == "exact":
        np.testing.assert_array_equal(piv_d, piv_h)
        np.testing.assert_array_equal(rd_d, rd_h)  # bitwise
        return
    if qr_mode == "default" and name == "near-ties":
        # two DIFFERENT candidates within 1e-12 of each other at step 0: the default falls back to the exact mode, so the pivot
        # ORDER (the row order of the standard form, standard_form.rs:142-181) is the host loop's to the last position
        np.testing.assert_array_equal(piv_d, piv_h)
        np.testing.assert_array_equal(rd_d, rd_h)
        return
    # fast: below the rank the trailing block is rounding noise and its largest entry is anybody's; what the
    # standard form consumes (standard_form.rs:143-181) are the pivots up to the rank and which |R_ii| are < EPS
    if name == "near-ties" and qr_mode == "fast":
        pytest.skip("without the fall-back the order of the two near-tied pivots is the fast reductions' — the case the default mode exists for")
    rank = int((rd_h >= 1e-10).sum())
    np.testing.assert_array_equal(piv_d[:rank], piv_h[:rank])
    np.testing.assert_allclose(rd_d, rd_h, rtol=1e-12, atol=1e-13)
    assert int((rd_d >= 1e-10).sum()) == rank


def _phase1_arrays(prob, solver):
    ph = prob._debug_phase1(solver)
    return None if ph is None else {k: np.array(v) if hasattr(v, "__len__") else v for k, v in ph.items()}


@pytest.mark.parametrize("solver", ["primal", "dual"])
def test_standard_form_is_identical_with_device_qr(solver):
    """Every fixture and netlib problem: the phase-1 arrays built on top of the standard form are the
    same whether the QR ran on the host or on the device."""
    from ellp_amd import Problem, parse_mps
    probs = [(fx["name"], lambda fx=fx: Problem.from_fixture(fx)) for fx in KA["problems"]]
    probs += [(fx["name"], lambda fx=fx: parse_mps(open(os.path.join(GOLDEN, fx["file"])).read())) for fx in KA["netlib"]]
    old = os.environ.get("ELLP_QR_DEVICE")
    try:
        for name, make in probs:
            os.environ["ELLP_QR_DEVICE"] = "0"
            a = _phase1_arrays(make(), solver)
            os.environ["ELLP_QR_DEVICE"] = "1"
            b = _phase1_arrays(make(), solver)
            assert (a is None) == (b is None), name
            if a is None:
                continue
            assert a.keys() == b.keys()
            for k in a:
                if isinstance(a[k], np.ndarray):
                    np.testing.assert_array_equal(a[k], b[k], err_msg=f"{name}:{k}")
                else:
                    assert a[k] == b[k], (name, k)
    finally:
        if old is None:
            os.environ.pop("ELLP_QR_DEVICE", None)
        else:
            os.environ["ELLP_QR_DEVICE"] = old


def test_full_api_solve_of_a_mid_size_dense_lp_uses_the_device_qr():
    """m=600, n=1500 through Problem -> solver.solve(): above the size threshold the standard form's
    QR runs on the device; the optimum must match the oracle's objective for the same LP solved from
    the directly built phase arrays (which skip the QR: full row rank keeps the rows in order)."""
    import time
    from ellp_amd import Bound, ConstraintOp, PrimalSimplexSolver, Problem, synth
    m, n = 600, 1500
    A, b, c = synth.dense_lp(20260301, m, n)
    p = Problem()
    ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
    for i in range(m):
        p.add_constraint([(ids[j], float(A[i, j])) for j in range(n)], ConstraintOp.Lte, float(b[i]))
    t0 = time.time()
    res = PrimalSimplexSolver.new(None).solve(p)
    dt = time.time() - t0
    assert res.kind == "optimal"
    from scipy.optimize import linprog
    ref = linprog(c, A_ub=A, b_ub=b, bounds=[(0, None)] * n, method="highs")
    assert abs(res.solution.obj() - ref.fun) < 1e-7 * (1 + abs(ref.fun))
    assert dt < 120


def test_standard_form_identical_on_random_lps():
    """300 random LPs (every operator and bound kind, integer data with exact ties, redundant rows):
    the phase-1 arrays are identical whichever side ran the rank check."""
    from ellp_amd import Problem
    from test_gpu_random import feasible_fixture, random_fixture
    cases = [random_fixture(np.random.default_rng(s)) for s in range(20000, 20150)]
    cases += [feasible_fixture(np.random.default_rng(s)) for s in range(21000, 21150)]
    old = os.environ.get("ELLP_QR_DEVICE")
    n_cmp = 0
    try:
        for k, fx in enumerate(cases):
            for solver in ("primal", "dual"):
                out = []
                for side in ("0", "1"):
                    os.environ["ELLP_QR_DEVICE"] = side
                    try:
                        out.append(_phase1_arrays(Problem.from_fixture(fx), solver))
                    except Exception as e:  # a panic of the reference's setup: must be raised on both sides
                        out.append(("raised", type(e).__name__))
                a, b = out
                if isinstance(a, tuple) or isinstance(b, tuple):
                    assert a == b, (k, solver, a if isinstance(a, tuple) else "ok", b if isinstance(b, tuple) else "ok")
                    continue
                assert (a is None) == (b is None), (k, solver)
                if a is None:
                    continue
                n_cmp += 1
                for key in a:
                    if isinstance(a[key], np.ndarray):
                        np.testing.assert_array_equal(a[key], b[key], err_msg=f"case {k} {solver}: {key}")
                    else:
                        assert a[key] == b[key], (k, solver, key)
    finally:
        if old is None:
            os.environ.pop("ELLP_QR_DEVICE", None)
        else:
            os.environ["ELLP_QR_DEVICE"] = old
    assert n_cmp > 250


@pytest.mark.parametrize("m,n", [(150, 400), (60, 5000), (300, 40)])
def test_standard_form_identical_on_mid_size_dense_lps(m, n):
    """Shapes that exercise the multi-block paths of the QR kernels: more than 256 columns of A^T,
    several row chunks in the update, a pivot column longer than one LDS chunk (n + m > 4096), and a
    tall A.  The phase-1 arrays must be identical whichever side ran the QR."""
    from ellp_amd import Bound, ConstraintOp, Problem, synth
    A, b, c = synth.dense_lp(20260301, m, n)
    A[m // 2] = A[1] + 0.5 * A[2]          # a redundant row: the rank decision matters
    b[m // 2] = b[1] + 0.5 * b[2]

    def build():
        p = Problem()
        ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
        for i in range(m):
            op = ConstraintOp.Eq if i in (1, 2, m // 2) else ConstraintOp.Lte
            p.add_constraint(list(zip(ids, A[i].tolist())), op, float(b[i]))
        return p
    old = os.environ.get("ELLP_QR_DEVICE")
    try:
        os.environ["ELLP_QR_DEVICE"] = "0"
        a = _phase1_arrays(build(), "primal")
        os.environ["ELLP_QR_DEVICE"] = "1"
        d = _phase1_arrays(build(), "primal")
    finally:
        if old is None:
            os.environ.pop("ELLP_QR_DEVICE", None)
        else:
            os.environ["ELLP_QR_DEVICE"] = old
    assert (a is None) == (d is None)
    if a is not None:
        assert a["m"] == d["m"] == m - 1            # the redundant row is gone on both sides
        for key in a:
            if isinstance(a[key], np.ndarray):
                np.testing.assert_array_equal(a[key], d[key], err_msg=key)
            else:
                assert a[key] == d[key], key
