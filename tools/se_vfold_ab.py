import json, os, sys, time
sys.path.insert(0, '/root/repo')
from ellp_amd import Bound, ConstraintOp, PrimalSimplexSolver, Problem, synth
m, n = 2000, 5000
A, b, c = synth.dense_lp(20260301, m, n)
p = Problem()
ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(n)]
for i in range(m):
    p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
for rep in range(2):
    t0 = time.perf_counter()
    res = PrimalSimplexSolver.new(None, flags=4).solve(p.clone())
    dt = time.perf_counter() - t0
    print(json.dumps({"btran_env": os.environ.get("ELLP_SE_BTRAN"), "status": res.kind, "iters": list(res.iters), "solve_s": round(dt, 3),
                      "rel": abs(res.solution.obj() + 2571.5834735467556) / 2571.58}), flush=True)
