#!/usr/bin/env python3
"""tools/seam_solve.py M N SEED FLAGS [fixture.json] — a synthetic LP of SURVEY §8d's family to optimality at the seam (phase 1,
hand-off on the device, phase 2 on one resident engine) with ellp_opts.flags = FLAGS (4: steepest edge), in slices with
progress; feasibility of the end point; against a HiGHS vertex (tools/highs_fixture.py) when one is given."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ellp_amd import _engine as E, synth
m, n, seed, flags = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fixture = sys.argv[5] if len(sys.argv) > 5 else None
t0 = time.perf_counter()
flat = synth.primal_phase1_flat(seed, m, n)
fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"], flat["lb"], flat["ub"], flat["x"], flat["B"], flat["N"], flat["Nb"])
eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, flags=flags))
iters, secs = [], []
step = int(os.environ.get("SEAM_SLICE", "20000"))
for phase in (1, 2):
    st, loop_s = E.MAXITER, 0.0
    while st == E.MAXITER:
        st, stats, msg = eng.run(step)
        loop_s += stats.t_loop_s
        extra = ""
        if os.environ.get("SEAM_WATCH"):
            eng.read_point()
            jm = int(np.argmin(fp.x))
            extra = f", min x {fp.x[jm]:.3e} at variable {jm} ({'basic' if jm in set(fp.B.tolist()) else 'nonbasic'})"
        print(f"phase {phase}: {int(stats.iters)} iterations, {loop_s:.1f} s, objective {stats.obj:.12g}, maintenance {int(stats.refactors)}{extra}", flush=True)
    assert st == E.OPTIMAL, (st, msg)
    iters.append(int(stats.iters)); secs.append(loop_s)
    if phase == 1:
        eng.read_point()
        assert abs(fp.obj()) < 1e-7, fp.obj()
        f2 = synth.primal_phase2_from(flat, fp.x, fp.B, fp.N, fp.Nb)
        if os.environ.get("SEAM_FRESH"):  # phase 2 on a NEW engine from the hand-over arrays (B^-1 rebuilt, steepest-edge weights initialised there)
            eng.close()
            fp = E.FlatProblem(f2["m"], f2["n"], f2["n_c"], f2["A"], f2["c"], f2["b"], f2["kind"], f2["lb"], f2["ub"], f2["x"], f2["B"], f2["N"], f2["Nb"])
            eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, flags=flags))
        else:
            eng.rephase(f2["c"], f2["kind"], f2["lb"], f2["ub"])
eng.read_point()
res = eng.inverse_residual()
eng.close()
A, b, c = synth.dense_lp(seed, m, n)
x = fp.x[:n]
rec = {"m": m, "n": n, "seed": seed, "flags": flags, "objective": float(c @ x), "iterations_phase1_phase2": iters,
       "loop_s_phase1_phase2": [round(v, 2) for v in secs], "max_Ax_minus_b": float((A @ x - b).max()), "min_x": float(x.min()),
       "max_artificial": float(np.abs(fp.x[n + m:]).max()), "inverse_residual_end": res, "wall_s": round(time.perf_counter() - t0, 1)}
if fixture:
    fx = json.load(open(fixture))
    x_ref = np.zeros(n); x_ref[np.asarray(fx["support"])] = np.asarray(fx["values"])
    rec["fixture_objective"] = fx["objective"]
    rec["rel_diff_objective"] = abs(rec["objective"] - fx["objective"]) / abs(fx["objective"])
    rec["max_abs_diff_x"] = float(np.abs(x - x_ref).max())
print(json.dumps(rec))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", f"seam_solve_{m}x{n}_flags{flags}.json"), "w"))
