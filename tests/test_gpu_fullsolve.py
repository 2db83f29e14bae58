"""Whole solves at BASELINE.json's full sizes through the user API, checked the way the reference's own tests check
a solve (assert_optimal!: status, objective AND optimal point, tests/problems/mod.rs:9-35) — against an INDEPENDENT
optimum: SciPy-HiGHS' vertex of the same LP, committed as a small fixture keyed by (generator, seed, m, n)
(tests/golden/synth_optimum_*.json, made by tools/highs_fixture.py: the support of x and its values).

Tolerances (the stated f64 tolerances of this repository, README.md): objective relative 1e-9, x absolute 1e-8
(the reference's own are 1e-8 absolute, relative 1e-6: mod.rs:6-7).

config 3 (2000 x 5000, primal): ~6.6e5 pivots, ~30 s of GPU — runs in the routine suite.
config 5 (4000 x 40000, primal): 2.77e6 pivots, 12 minutes of GPU — marked slow: run it with ELLP_SLOW=1 (and let it print: a
silent run of that length is taken for hung); the result of the run made for this repository (tools/seam_solve.py, the same
solve with the hand-off on the device) is profiles/r03_full_solve_c5_dantzig.json: objective rel. 6e-15, max |dx| 1.7e-9.
The same LP under the steepest-edge extension: 2.1e5 pivots, 81 s in the loops, objective rel. 1.7e-14, max |dx| 1.9e-9
(profiles/r03_full_solve_c5_steepest_edge.json)."""
import json
import os
import time

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _fixture(seed, m, n):
    path = os.path.join(GOLDEN, f"synth_optimum_{seed}_{m}x{n}.json")
    if not os.path.exists(path):
        pytest.skip(f"{path} is not committed (tools/highs_fixture.py {m} {n} {seed} makes it)")
    with open(path) as f:
        fx = json.load(f)
    assert (fx["seed"], fx["m"], fx["n"]) == (seed, m, n)
    x = np.zeros(n)
    x[np.asarray(fx["support"], dtype=np.int64)] = np.asarray(fx["values"])
    return fx, x


def _solve(seed, m, n, tag):
    from ellp_amd import Bound, ConstraintOp, PrimalSimplexSolver, Problem, synth
    fx, x_ref = _fixture(seed, m, n)
    A, b, c = synth.dense_lp(seed, m, n)
    p = Problem()
    ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
    for i in range(m):
        p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
    t0 = time.perf_counter()
    res = PrimalSimplexSolver.new(None).solve(p)
    dt = time.perf_counter() - t0
    assert res.kind == "optimal"
    obj = res.solution.obj()
    x = np.asarray(res.solution.x())[:n]
    rec = {"config": tag, "seed": seed, "m": m, "n": n, "status": res.kind, "objective": obj,
           "fixture_objective": fx["objective"], "rel_diff_objective": abs(obj - fx["objective"]) / abs(fx["objective"]),
           "max_abs_diff_x": float(np.abs(x - x_ref).max()), "iterations_phase1_phase2": list(res.iters),
           "solve_s": round(dt, 2), "pivots_per_s_incl_setup": round(sum(res.iters) / dt, 1)}
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"full_solve_{tag}.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps(rec))
    # assert_optimal!: objective and point
    assert abs(obj - fx["objective"]) <= 1e-9 * abs(fx["objective"]), rec
    assert np.abs(x - x_ref).max() <= 1e-8, rec
    assert np.all(A @ x <= b + 1e-8) and x.min() >= -1e-9
    return rec


def test_c3_full_solve_matches_committed_optimum():
    _solve(20260301, 2000, 5000, "c3")


@pytest.mark.slow
@pytest.mark.parametrize("flags", [0, 4], ids=["dantzig", "steepest-edge"])
def test_c5_full_solve_matches_committed_optimum(flags):
    """flags = 4: the steepest-edge extension (2.1e5 pivots, 90 s; profiles/r03_full_solve_c5_steepest_edge.json).
    config 5 on the arrays the reference's phases hand over at the seam (ellp_amd/synth.py builds them directly: 160 M
    coefficients through the Python Problem API would take longer than the solve), both phases on one resident engine, in
    slices so that progress is visible"""
    from ellp_amd import _engine as E
    from ellp_amd import synth
    seed, m, n = 20260305, 4000, 40000
    fx, x_ref = _fixture(seed, m, n)
    flat = synth.primal_phase1_flat(seed, m, n)
    t0 = time.perf_counter()
    iters, secs = [], []
    # both phases on ONE resident engine, the hand-off on the device (ellp_engine_rephase) — what the host mirror's
    # PrimalSimplexSolver does.  (A second engine created at phase 1's end basis also works under the reference's rule; under
    # steepest edge that start met a run of tiny pivots it did not recover from after 3e4 iterations — DESIGN.md §5.)
    fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"], flat["lb"], flat["ub"],
                       flat["x"], flat["B"], flat["N"], flat["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, flags=flags))
    for phase in (1, 2):
        st, loop_s = E.MAXITER, 0.0
        while st == E.MAXITER:
            st, stats, msg = eng.run(100000 if flags == 0 else 20000)
            loop_s += stats.t_loop_s
            print(f"phase {phase}: {int(stats.iters)} iterations, {loop_s:.1f} s, objective {stats.obj:.12g}", flush=True)
        assert st == E.OPTIMAL, msg
        iters.append(int(stats.iters))
        secs.append(loop_s)
        eng.read_point()
        if phase == 1:
            assert abs(fp.obj()) < 1e-7
            f2 = synth.primal_phase2_from(flat, fp.x, fp.B, fp.N, fp.Nb)
            eng.rephase(f2["c"], f2["kind"], f2["lb"], f2["ub"])
    eng.close()
    c_full = np.zeros(len(fp.x)); c_full[:n] = synth.dense_lp(seed, m, n)[2]
    x = fp.x[:n]
    obj = float(c_full[:n] @ x)
    rec = {"config": "c5", "flags": flags, "seed": seed, "m": m, "n": n, "status": "optimal", "objective": obj, "fixture_objective": fx["objective"],
           "rel_diff_objective": abs(obj - fx["objective"]) / abs(fx["objective"]), "max_abs_diff_x": float(np.abs(x - x_ref).max()),
           "iterations_phase1_phase2": iters, "loop_s_phase1_phase2": [round(v, 2) for v in secs],
           "pivots_per_s_in_the_loops": round(sum(iters) / sum(secs), 1), "wall_s": round(time.perf_counter() - t0, 1)}
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"full_solve_c5_flags{flags}.json"), "w") as fh:
        json.dump(rec, fh)
    print(json.dumps(rec))
    assert abs(obj - fx["objective"]) <= 1e-9 * abs(fx["objective"]), rec
    assert np.abs(x - x_ref).max() <= 1e-8, rec
