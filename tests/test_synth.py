"""The numpy generator (product side, used by bench.py) draws the same LP as the oracle's C
generator, and the directly-built phase-1 arrays agree with what the oracle's setup produces
(up to the QR's row order, which the direct construction deliberately skips)."""
import numpy as np

from ellp_amd import synth
from oracle import ellp_oracle as eo


def test_generator_matches_c():
    for seed, m, n in [(20260301, 20, 50), (7, 33, 17), (20260305, 64, 200)]:
        A, b, c = synth.dense_lp(seed, m, n)
        A2, b2, c2 = eo.synth_dense_lp(seed, m, n)
        np.testing.assert_array_equal(A, A2)
        np.testing.assert_array_equal(b, b2)
        np.testing.assert_array_equal(c, c2)


def test_direct_phase1_solves_to_highs_objective():
    """Solving the directly-built phase 1 / phase 2 with the ORACLE loops gives the optimum
    SURVEY §8d lists (independent HiGHS value)."""
    seed, m, n = 20260301, 20, 50
    flat = synth.primal_phase1_flat(seed, m, n)

    class V:  # minimal Phase-like view for the oracle binding
        pass
    v = V()
    for k, val in flat.items():
        setattr(v, k, val)
    v.nB, v.nN = len(flat["B"]), len(flat["N"])
    st, it1, _ = eo.primal_solve_with_initial(v)
    assert st == eo.OPTIMAL and abs(np.dot(v.c, v.x)) < 1e-9
    f2 = synth.primal_phase2_from(flat, v.x, v.B, v.N, v.Nb)
    v2 = V()
    for k, val in f2.items():
        setattr(v2, k, val)
    v2.nB, v2.nN = len(f2["B"]), len(f2["N"])
    st, it2, _ = eo.primal_solve_with_initial(v2)
    assert st == eo.OPTIMAL
    assert abs(np.dot(v2.c, v2.x) - (-21.72074513030114)) < 1e-8


def test_dual_start_is_dual_feasible_and_solves_to_highs():
    """The directly-built dual start satisfies the reference's dual-feasibility assertion
    (dual…:139-151) and the ORACLE's dual loop drives it to the HiGHS optimum."""
    from scipy.optimize import linprog
    seed, m, n = 20260301, 20, 50
    f = synth.dual_start_flat(seed, m, n)

    class V:
        pass
    v = V()
    for k, val in f.items():
        setattr(v, k, val)
    v.nB, v.nN = len(f["B"]), len(f["N"])
    # y = A_B^-T c_B = 0, d = c - A^T y = c >= 0, x_B = A_B^-1 b = -b
    assert np.all(f["d"][f["N"]] >= 0) and np.all(f["x"][f["B"]] < 0)
    st, it, msg = eo.dual_solve_with_initial(v)
    assert st == eo.OPTIMAL, msg
    A, b, c = synth.covering_lp(seed, m, n)
    h = linprog(c, A_ub=-A, b_ub=-b, bounds=(0, None), method="highs")
    assert abs(np.dot(v.c, v.x) - h.fun) < 1e-8
