#!/usr/bin/env python3
"""Full two-phase primal solve of the synthetic dense LP on the GPU engine, objective checked
against an independent solver (SciPy HiGHS).  Usage: python tools/full_solve.py [m n [seed]]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from ellp_amd import _engine as E  # noqa: E402
from ellp_amd import synth  # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20260301
    flat = synth.primal_phase1_flat(seed, m, n)
    out = {"m": m, "n": n, "seed": seed}
    t0 = time.perf_counter()
    fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"],
                       flat["lb"], flat["ub"], flat["x"], flat["B"], flat["N"], flat["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None))
    st, stats, msg = eng.run(1 << 60)
    eng.read_point()
    res1 = eng.inverse_residual()
    eng.close()
    t1 = time.perf_counter()
    out["phase1"] = {"status": E.STATUS_NAME[st], "iters": int(stats.iters), "obj": fp.obj(), "loop_s": stats.t_loop_s,
                     "maintenance": int(stats.refactors), "inverse_residual_end": res1, "msg": msg}
    print("phase 1:", out["phase1"], flush=True)
    assert st == E.OPTIMAL and abs(fp.obj()) < 1e-7
    f2 = synth.primal_phase2_from(flat, fp.x, fp.B, fp.N, fp.Nb)
    fp2 = E.FlatProblem(f2["m"], f2["n"], f2["n_c"], f2["A"], f2["c"], f2["b"], f2["kind"], f2["lb"], f2["ub"],
                        f2["x"], f2["B"], f2["N"], f2["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp2, E.default_opts(max_iter=None))
    st, stats, msg = eng.run(1 << 60)
    eng.read_point()
    res2 = eng.inverse_residual()
    eng.close()
    t2 = time.perf_counter()
    out["phase2"] = {"status": E.STATUS_NAME[st], "iters": int(stats.iters), "obj": fp2.obj(), "loop_s": stats.t_loop_s,
                     "maintenance": int(stats.refactors), "inverse_residual_end": res2, "msg": msg}
    print("phase 2:", out["phase2"], flush=True)
    out["wall_s"] = {"phase1": t1 - t0, "phase2": t2 - t1}
    out["pivots_per_s"] = (out["phase1"]["iters"] + out["phase2"]["iters"]) / (out["phase1"]["loop_s"] + out["phase2"]["loop_s"])
    x = fp2.x[:n]
    A, b, c = synth.dense_lp(seed, m, n)
    out["feasibility"] = {"max_Ax_minus_b": float(np.max(A @ x - b)), "min_x": float(x.min())}
    try:
        from scipy.optimize import linprog
        th = time.perf_counter()
        h = linprog(c, A_ub=A, b_ub=b, bounds=(0, None), method="highs")
        out["highs"] = {"obj": float(h.fun), "status": int(h.status), "seconds": time.perf_counter() - th}
        out["abs_diff_vs_highs"] = abs(fp2.obj() - h.fun)
        out["rel_diff_vs_highs"] = abs(fp2.obj() - h.fun) / max(1.0, abs(h.fun))
    except Exception as ex:  # pragma: no cover
        out["highs"] = {"error": repr(ex)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
