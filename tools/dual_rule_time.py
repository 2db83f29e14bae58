#!/usr/bin/env python3
"""tools/dual_rule_time.py [sizes...] — the dual solver through the user API (DualSimplexSolver::new(None).solve) with the
reference's leaving-row rule and with the opt-in largest-violation rule (ellp_opts.flags = 2): iterations and wall time to
optimality, optimum against the committed HiGHS value where there is one.  FLAGS=2,18 (environment) chooses the flag sets:
16 adds the bound-flipping ratio test (LU-per-iteration kernels, up to 1,024 rows)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ellp_amd import Bound, ConstraintOp, DualSimplexSolver, Problem, synth  # noqa: E402

HIGHS = {(200, 500): -251.6515333670212, (2000, 5000): -2571.5834735467556}
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(200, 500), (600, 1500)]
out = []
for (m, n) in sizes:
    A, b, c = synth.dense_lp(20260301, m, n)
    fl = [int(v) for v in os.environ["FLAGS"].split(",")] if os.environ.get("FLAGS") else ((0, 2) if m <= 200 else (2,))
    for flags in fl:
        p = Problem()
        ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
        for i in range(m):
            p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
        t0 = time.perf_counter()
        res = DualSimplexSolver.new(None, flags=flags).solve(p)
        dt = time.perf_counter() - t0
        rec = {"m": m, "n": n, "flags": flags, "leaving_row_rule": "largest violation (extension)" if flags & 2 else "first violated (reference)",
               "ratio_test": "bound flipping (extension)" if flags & 16 else "reference",
               "status": res.kind, "iterations_phase1_phase2": list(res.iters), "solve_s": round(dt, 2),
               "objective": res.solution.obj() if res.kind == "optimal" else None}
        if (m, n) in HIGHS and rec["objective"] is not None:
            rec["rel_diff_to_highs"] = abs(rec["objective"] - HIGHS[(m, n)]) / abs(HIGHS[(m, n)])
        print(json.dumps(rec), flush=True)
        out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dual_rule_time.json"), "w"), indent=1)
