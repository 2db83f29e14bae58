"""Deterministic synthetic LP family of BASELINE.json's configs 3-5.

Counter-based splitmix64 -> U[0,1): draw k uses state seed + (k+1)*0x9E3779B97F4A7C15, so the
whole matrix is generated vectorised.  Draw order: A[i][j] = 0.1 + u (i outer, j inner), then
x0[j] = u, b = A x0, then c[j] = -(0.1 + u).  All rows `<=`, all variables Lower(0).
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix_uniform(seed, start, count):
    """u_k for k in [start, start+count): float64 in [0,1)."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def dense_lp(seed, m, n):
    """Returns A (m x n, Fortran order), b (m), c (n)."""
    A = np.empty((m, n), dtype=np.float64, order="F")
    rows_per = max(1, (1 << 22) // max(n, 1))
    for i0 in range(0, m, rows_per):
        i1 = min(m, i0 + rows_per)
        u = splitmix_uniform(seed, i0 * n, (i1 - i0) * n).reshape(i1 - i0, n)
        A[i0:i1, :] = 0.1 + u
    x0 = splitmix_uniform(seed, m * n, n)
    # b_i = sum_j A_ij x0_j accumulated in j order (matches the C generator's loop)
    b = np.zeros(m)
    for j in range(n):
        b += A[:, j] * x0[j]
    c = -(0.1 + splitmix_uniform(seed, m * n + n, n))
    return A, b, c


def primal_phase1_flat(seed, m, n):
    """Primal phase-1 StandardForm + Point for the family, built directly (no QR row
    re-ordering): what primal_problem.rs:80-261 produces for an
    all-`<=`, all-Lower(0) LP whose rows are kept in order.
      columns: [n structurals | m slacks, row i's slack in column n+m-1-i (standard_form.rs:115-136)
                | m artificials, +e_i because b_i > 0 (primal_problem.rs:236-246)]
    Returns a dict of C-ABI shaped arrays."""
    A, b, c = dense_lp(seed, m, n)
    ntot = n + m
    ncols = ntot + m
    Af = np.zeros((m, ncols), dtype=np.float64, order="F")
    Af[:, :n] = A
    idx = np.arange(m)
    Af[idx, n + m - 1 - idx] = 1.0
    Af[idx, ntot + idx] = 1.0
    cc = np.zeros(ncols)
    cc[ntot:] = 1.0
    x = np.zeros(ncols)
    x[ntot:] = np.abs(b)
    return dict(
        m=m, n=ncols, n_c=ncols, A=Af.reshape(-1, order="F"), c=cc, b=b.copy(),
        kind=np.full(ncols, 1, dtype=np.uint8), lb=np.zeros(ncols), ub=np.zeros(ncols), x=x,
        B=np.arange(ntot, ncols, dtype=np.int64), N=np.arange(ntot, dtype=np.int64),
        Nb=np.zeros(ntot, dtype=np.uint8), c_struct=c, n_struct=n)


def primal_phase2_from(flat, x, B, N, Nb):
    """primal_problem.rs:263-291 for the family: restore c on the structurals, artificials
    become Fixed(0) with cost 0."""
    out = dict(flat)
    n, m = flat["n_struct"], flat["m"]
    ntot = n + m
    c = np.zeros(flat["n_c"])
    c[:n] = flat["c_struct"]
    kind = flat["kind"].copy()
    kind[ntot:] = 4
    out.update(c=c, kind=kind, x=np.array(x, copy=True), B=np.array(B, copy=True),
               N=np.array(N, copy=True), Nb=np.array(Nb, copy=True))
    return out


def covering_lp(seed, m, n):
    """Covering variant of the family for the DUAL loop: min c.x, c > 0, A x >= b, x >= 0 with the
    same A and b = A x0 (> 0) and c_j = +(0.1 + u).  Returns A (Fortran order), b, c."""
    A, b, c = dense_lp(seed, m, n)
    return A, b, -c


def dual_start_flat(seed, m, n):
    """StandardForm + DualFeasiblePoint for covering_lp, built directly: rows `>=` get slack
    coefficient -1, row i's slack in column n+m-1-i (standard_form.rs:115-136).  The slack basis is
    dual feasible (y = 0, d = c >= 0) and primal infeasible (s = -b < 0), so
    DualSimplexSolver::solve_with_initial (dual_simplex_solver.rs:110) starts pivoting at once —
    this is what dual phase 2 looks like, without the reference's phase-1 detour (whose
    first-infeasible leaving rule needs ~20x more pivots per doubling of m)."""
    A, b, c = covering_lp(seed, m, n)
    ncols = n + m
    Af = np.zeros((m, ncols), dtype=np.float64, order="F")
    Af[:, :n] = A
    idx = np.arange(m)
    Af[idx, n + m - 1 - idx] = -1.0
    cc = np.zeros(ncols)
    cc[:n] = c
    x = np.zeros(ncols)
    x[n + m - 1 - idx] = -b  # A_B = -I (permuted): s = -b
    B = (n + m - 1 - idx).astype(np.int64)
    return dict(
        m=m, n=ncols, n_c=ncols, A=Af.reshape(-1, order="F"), c=cc, b=b.copy(),
        kind=np.full(ncols, 1, dtype=np.uint8), lb=np.zeros(ncols), ub=np.zeros(ncols), x=x,
        B=B, N=np.arange(n, dtype=np.int64), Nb=np.zeros(n, dtype=np.uint8),
        y=np.zeros(m), d=cc.copy(), c_struct=c, n_struct=n)
