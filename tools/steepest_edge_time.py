#!/usr/bin/env python3
"""tools/steepest_edge_time.py [MxN ...] — the primal solver through the user API with the reference's Dantzig rule and with the
opt-in steepest-edge extension (ellp_opts.flags = 4): iterations and wall time to optimality."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ellp_amd import Bound, ConstraintOp, PrimalSimplexSolver, Problem, synth  # noqa: E402

HIGHS = {(200, 500): -251.6515333670212, (2000, 5000): -2571.5834735467556}
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(200, 500), (600, 1500), (2000, 5000)]
out = []
for (m, n) in sizes:
    A, b, c = synth.dense_lp(20260301, m, n)
    for flags in (0, 4):
        p = Problem()
        ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
        for i in range(m):
            p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
        t0 = time.perf_counter()
        res = PrimalSimplexSolver.new(None, flags=flags).solve(p)
        dt = time.perf_counter() - t0
        rec = {"m": m, "n": n, "pricing": "steepest edge (extension)" if flags else "Dantzig (reference)", "status": res.kind,
               "iterations_phase1_phase2": list(res.iters), "solve_s": round(dt, 2),
               "objective": res.solution.obj() if res.kind == "optimal" else None}
        if (m, n) in HIGHS and rec["objective"] is not None:
            rec["rel_diff_to_highs"] = abs(rec["objective"] - HIGHS[(m, n)]) / abs(HIGHS[(m, n)])
        print(json.dumps(rec), flush=True)
        out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "steepest_edge_time.json"), "w"), indent=1)
