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
