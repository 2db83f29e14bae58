#!/usr/bin/env python3
"""tools/redo_large_probe.py NAME COPIES TRIAL [TRIAL ...] — the redo above 1,024 rows on single orders of a block-diagonal netlib
replication through the user API (primal): how the solve ends, the counters of both phases and the wall time.
Writes gpurun_out/redo_large_<name>x<copies>.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_hybrid as T  # noqa: E402
import ellp_amd  # noqa: E402

name, copies = sys.argv[1], int(sys.argv[2])
trials = [int(t) for t in sys.argv[3:]]
solver = os.environ.get("SOLVER", "primal")
out = []
for trial, fx, want in T._orders(name, copies, max(trials) + 1):
    if trial not in trials:
        continue
    prob = ellp_amd.Problem.from_fixture(fx)
    cls = ellp_amd.PrimalSimplexSolver if solver == "primal" else ellp_amd.DualSimplexSolver
    t0 = time.time()
    try:
        r = cls.new(400000).solve(prob)
        ok = r.kind == ellp_amd.SolverResult.Optimal and abs(r.solution.obj() / want - 1.0) < 1e-9
        what = [str(r.kind), list(r.iters), r.solution.obj() if r.kind == ellp_amd.SolverResult.Optimal else None]
        if ok:
            v = T.fixture_violation(fx, r.solution.x())
            ok = v[0] < 1e-8 and v[1] < 1e-8
            what.append(list(map(float, v)))
    except (RuntimeError, ellp_amd.EllPError) as ex:
        ok, what = False, [repr(ex)[:160]]
    rec = {"problem": f"{name}x{copies}", "solver": solver, "trial": trial, "at_the_pinned_optimum": bool(ok), "what": what,
           "seconds": round(time.time() - t0, 1)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    path = os.path.join(ROOT, "gpurun_out", f"redo_large_{name}x{copies}_{solver}.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
