"""tools/full_api_solve.py [m n] [--dual] — the user's view: build the dense LP of config 3 through the Problem API,
call PrimalSimplexSolver::new(None).solve(problem) (C++ host mirror: standard form with the rank
check on the device, phase construction, both phases on one resident engine) and compare with the
independent HiGHS objective of SURVEY.md §8d.  Prints one JSON line with the timings."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ellp_amd import Bound, ConstraintOp, DualSimplexSolver, PrimalSimplexSolver, Problem, synth

dual = "--dual" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, n = (int(args[0]), int(args[1])) if len(args) > 1 else (2000, 5000)
A, b, c = synth.dense_lp(20260301, m, n)
t0 = time.perf_counter()
p = Problem()
ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
for i in range(m):
    p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
t_build = time.perf_counter() - t0
t0 = time.perf_counter()
res = (DualSimplexSolver if dual else PrimalSimplexSolver).new(None).solve(p)
t_solve = time.perf_counter() - t0
out = {"solver": "dual" if dual else "primal", "m": m, "n": n, "status": res.kind, "objective": res.solution.obj() if res.kind == "optimal" else None,
       "iterations_phase1_phase2": list(res.iters), "problem_build_s": round(t_build, 2),
       "solve_s": round(t_solve, 2), "pivots_per_s_incl_setup": round(sum(res.iters) / t_solve, 1)}
highs = {(2000, 5000): -2571.583473546866}.get((m, n))
if highs is not None and out["objective"] is not None:
    out["highs_objective"] = highs
    out["rel_diff"] = abs(out["objective"] - highs) / abs(highs)
print(json.dumps(out))
