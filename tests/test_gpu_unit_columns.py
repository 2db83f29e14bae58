"""Unit columns in the primal pricing pass ("sparse A", first step; PriceArgs::vs_row): the slack
and artificial columns every standard-form LP carries (standard_form.rs:115-136, primal_problem.rs:236-246) are
priced from their single entry instead of being streamed.  The dot product is the same number either way (every
other term is an exact zero), so an engine that skips them must take EXACTLY the pivots of one that streams them
(ellp_opts.flags = ELLP_FLAG_DENSE_PRICING) and end with the same bits of x — on every pricing kernel: the
wave-per-column-pair and the block-per-column-group shapes of the two-launch pipeline, the three-launch kernels, phase
1 (artificials basic, slacks nonbasic) and phase 2 (artificials nonbasic and fixed)."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def _run(f, iters, **opts):
    E = _E()
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    st, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=iters, **opts))
    return st, stats, fp, msg


@pytest.mark.parametrize("m,n,iters,pipeline", [
    (520, 5300, 1500, 0),    # k_price2_wave (config 3's kernel): ld >= 512 and >= 5 columns per block
    (400, 900, 3000, 0),     # k_price2<T, false>: block per column group
    (400, 900, 1500, 1),     # three launches: k_price<T, 0, false>
    (520, 5300, 600, 1),     # three launches: k_price_wave<0>
    (400, 4000, 1500, 0),    # k_price2<T, false> with several groups of four columns per block
    (400, 4000, 800, 1),     # k_price<T, 0, false>, likewise
    (1000, 20000, 600, 0),   # A_N beyond 160 MB: k_price2<T, true> (non-temporal stream; config 5's kernel)
])
def test_same_pivots_with_and_without_the_shortcut(m, n, iters, pipeline):
    from ellp_amd import synth
    E = _E()
    f = synth.primal_phase1_flat(20260301 + m, m, n)
    st_a, sa, fa, msg_a = _run(f, iters, pipeline=pipeline)
    st_b, sb, fb, msg_b = _run(f, iters, pipeline=pipeline, flags=1)
    assert st_a == st_b, (msg_a, msg_b)
    assert sa.iters == sb.iters and sa.pivots == sb.pivots
    np.testing.assert_array_equal(fa.B, fb.B)
    np.testing.assert_array_equal(fa.N, fb.N)
    np.testing.assert_array_equal(fa.Nb, fb.Nb)
    assert fa.x.tobytes() == fb.x.tobytes()


def test_phase_two_with_fixed_artificials_and_the_oracle():
    """both phases of a 200 x 420 LP on the explicit-inverse engine (pipeline 2), phase 2 with its 200 artificial
    columns nonbasic and Fixed(0) (quirk Q6): pivot for pivot the oracle's, with the shortcut on"""
    E = _E()
    p1, err = eo.primal_phase1(eo.synth_problem(77, 200, 420))
    ph = p1
    for phase in (1, 2):
        v = ph.view()
        ov = v.copy()
        st_o, it_o, _ = eo.primal_solve_with_initial(ov, 100000)
        fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN])
        st_g, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=100000, pipeline=2))
        assert st_g == st_o == E.OPTIMAL and stats.iters == it_o, (phase, st_g, st_o, stats.iters, it_o, msg)
        np.testing.assert_array_equal(fp.B, ov.B)
        np.testing.assert_allclose(fp.x, ov.x, rtol=0, atol=1e-9 * (1 + np.abs(ov.x).max()))
        if phase == 1:
            ph.store_point(ov)
            ph = eo.primal_phase2(ph)
