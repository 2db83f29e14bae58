"""The optional per-iteration objective trace (what the reference's
`debug!("{iter}  |  {obj:.8E}")` line prints, primal_simplex_solver.rs:161 / dual_simplex_solver.rs:189):
ellp_opts.trace_len > 0 keeps (iteration, objective) of the last iterations in a ring on the device.
Checked on every execution path against the oracle's objective after the same number of iterations."""
import numpy as np
import pytest

from oracle import ellp_oracle as eo

pytestmark = pytest.mark.gpu


def _E():
    from ellp_amd import _engine as E
    return E


def _view(f):
    class V:
        pass
    v = V()
    for k, val in f.items():
        setattr(v, k, val.copy() if hasattr(val, "copy") else val)
    v.nB, v.nN = len(f["B"]), len(f["N"])
    return v


@pytest.mark.parametrize("m,n,pipeline", [(50, 120, 0), (50, 120, 1), (150, 400, 1), (150, 400, 2)],
                         ids=["persistent-workgroup", "three-launch-small", "three-launch", "two-launch"])
def test_primal_objective_trace(m, n, pipeline):
    E = _E()
    from ellp_amd import synth
    f = synth.primal_phase1_flat(20260301, m, n)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=pipeline, trace_len=1 << 14))
    st, stats, msg = eng.run(37)          # slices: the trace must not care
    st, stats, msg = eng.run(1 << 40)
    assert st == E.OPTIMAL, msg
    its, objs = eng.read_trace()
    eng.read_point()
    eng.close()
    # one entry per completed iteration (the last loop body only finds "optimal": no entry)
    assert len(its) == stats.iters - 1 and np.all(np.diff(its.astype(np.int64)) == 1) and its[0] == 1
    scale = 1.0 + abs(objs[0])
    assert np.all(np.diff(objs) <= 1e-9 * scale)               # phase-1 objective never increases
    assert abs(objs[-1] - fp.obj()) < 1e-8 * scale             # carried incrementally, still c.x at the end
    for k in (1, 7, 60, int(its[-1])):                         # the oracle's c.x after k iterations
        ov = _view(f)
        eo.primal_solve_with_initial(ov, k)
        assert abs(objs[k - 1] - float(np.dot(f["c"], ov.x))) < 1e-8 * scale, k


def test_dual_objective_trace_and_ring_wraps():
    E = _E()
    from ellp_amd import synth
    f = synth.dual_start_flat(20260301, 60, 150)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"], f["y"], f["d"])
    eng = E.Engine(E.ENGINE_DUAL, fp, E.default_opts(max_iter=None, trace_len=16))
    st, stats, msg = eng.run(1 << 40)
    assert st == E.OPTIMAL and stats.iters > 40, msg
    its, objs = eng.read_trace()
    eng.read_point()
    eng.close()
    assert len(its) == 16 and its[-1] == stats.iters - 1 and np.all(np.diff(its.astype(np.int64)) == 1)
    assert np.all(np.diff(objs) >= -1e-9)                       # the dual objective never decreases
    assert abs(objs[-1] - float(np.dot(f["b"], fp.y))) < 1e-8 * (1 + abs(objs[-1]))  # all bounds Lower(0): b.y


def test_trace_is_off_by_default():
    E = _E()
    from ellp_amd import synth
    f = synth.primal_phase1_flat(3, 30, 70)
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                       f["B"], f["N"], f["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None))
    eng.run(20)
    its, objs = eng.read_trace()
    eng.close()
    assert len(its) == 0
