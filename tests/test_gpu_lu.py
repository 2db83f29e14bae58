"""the basis of dual phase 1 (`std_form.A.transpose().lu()`, dual_problem.rs:139-160) on the
device.  Bar: bitwise the host loop's pivots and U_ii — so the phase-1 basis, and everything downstream, is
identical whichever side computed the LU."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu


def host_lu_of_transpose(A):
    """Plain restatement of ellp_amd/csrc/host/dense.h LU (= oracle lu_factor_inplace) on M = A^T: first maximum,
    zero column skipped, multipliers a * (1 / diag), update (-c_k[i]) * c_i[r] + c_k[r] skipped for a zero c_k[i]."""
    M = np.array(A.T, dtype=np.float64, order="F")
    rows, cols = M.shape
    piv, ud = [], []
    for i in range(min(rows, cols)):
        col = np.abs(M[i:, i])
        with np.errstate(invalid="ignore"):
            p = i
            best = col[0]
            for r in range(1, len(col)):
                if col[r] > best:
                    best, p = col[r], i + r
        diag = M[p, i]
        if diag == 0.0:
            piv.append(i)
            ud.append(M[i, i])
            continue
        if p != i:
            M[[i, p], :] = M[[p, i], :]
        piv.append(p)
        inv = 1.0 / diag
        M[i + 1:, i] = M[i + 1:, i] * inv
        for k in range(i + 1, cols):
            f = -M[i, k]
            if f == 0.0:
                continue
            M[i + 1:, k] = f * M[i + 1:, i] + M[i + 1:, k]
        ud.append(M[i, i])
    return np.array(piv, dtype=np.int64), np.array(ud)


def _cases():
    rng = np.random.default_rng(12)
    yield "wide", rng.uniform(-1, 1, size=(7, 19))
    yield "square", rng.uniform(-1, 1, size=(9, 9))
    yield "ties", rng.integers(-2, 3, size=(8, 14)).astype(float)       # many exact ties, zeros in the pivot rows
    A = rng.uniform(-1, 1, size=(8, 12))
    A[5] = A[2]                                                          # rank-deficient: a zero column of A^T late on
    yield "rank-deficient", A
    A = rng.uniform(-1, 1, size=(6, 10))
    A[3] = 0.0                                                           # a zero row of A = a zero column of A^T: skipped
    yield "zero-column", A
    yield "one-row", rng.uniform(-1, 1, size=(1, 5))
    A = np.zeros((5, 9))
    A[np.arange(5), np.arange(5) + 2] = 1.0                              # a permutation: every step swaps, nothing to update
    yield "permutation", A
    yield "sparse", rng.uniform(-1, 1, size=(40, 130)) * (rng.random((40, 130)) < 0.1)
    yield "mid", rng.uniform(-1, 1, size=(130, 300))                    # more than one block of rows, several waves per row


@pytest.mark.parametrize("name,A", list(_cases()), ids=[c[0] for c in _cases()])
def test_device_lu_is_bitwise_the_host_loop(name, A):
    from ellp_amd import _engine as E
    piv_h, ud_h = host_lu_of_transpose(A)
    piv_d, ud_d = E.lu_transposed(A)
    np.testing.assert_array_equal(piv_d, piv_h)
    assert ud_d.tobytes() == ud_h.tobytes(), (name, np.abs(ud_d - ud_h).max())


def test_dual_phase1_basis_from_the_device_equals_the_oracles(monkeypatch):
    """the host mirror's DualPhase1 with the LU on the device (forced, the LP is below the size threshold) against
    the oracle's own construction: same B, same N order"""
    import ellp_amd
    monkeypatch.setenv("ELLP_LU_DEVICE", "1")
    prob_o = eo.synth_problem(20260301, 60, 140)
    d1, err = eo.dual_phase1(prob_o)
    assert d1 is not None and not err
    v = d1.view()
    # the same LP through the user API of the host mirror
    from ellp_amd import synth
    A, b, c = synth.dense_lp(20260301, 60, 140)
    p = ellp_amd.Problem()
    ids = [p.add_var(float(c[j]), ellp_amd.Bound.Lower(0.0)) for j in range(140)]
    for i in range(60):
        p.add_constraint(list(zip(ids, A[i].tolist())), ellp_amd.ConstraintOp.Lte, float(b[i]))
    f = p._debug_phase1("dual")
    np.testing.assert_array_equal(f["B"], v.B)
    np.testing.assert_array_equal(f["N"], v.N[:v.nN])
    np.testing.assert_array_equal(f["x"], v.x)
    assert np.asarray(f["y"]).tobytes() == np.asarray(v.y).tobytes()


def test_needs_a_row_per_column():
    from ellp_amd import _engine as E
    with pytest.raises(E.EllpHipError):
        E.lu_transposed(np.ones((5, 3)))
