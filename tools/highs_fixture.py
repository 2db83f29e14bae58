#!/usr/bin/env python3
"""tools/highs_fixture.py m n seed [out.json] — the independent optimum (objective AND point) of a synthetic dense LP
of SURVEY.md §8d's family from SciPy's HiGHS, as a small fixture keyed by (generator, seed, m, n): the support of x
and its values (a vertex: at most m nonzeros).  tests/golden/synth_optimum_*.json are made by this script."""
import json
import os
import sys
import time

import numpy as np
from scipy.optimize import linprog

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd import synth  # noqa: E402

m, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A, b, c = synth.dense_lp(seed, m, n)
t0 = time.time()
r = linprog(c, A_ub=A, b_ub=b, bounds=(0, None), method="highs-ds",
            options={"primal_feasibility_tolerance": 1e-10, "dual_feasibility_tolerance": 1e-10})
dt = time.time() - t0
assert r.status == 0, r.message
x = np.asarray(r.x)
# polish the vertex: the basis is (support of x) + (slacks of the rows that are not tight); solve it in double
slack = b - A @ x
sup = np.flatnonzero(x > 1e-7)
tight = np.argsort(slack)[:len(sup)]  # a nondegenerate vertex has as many tight rows as positive variables (HiGHS leaves them
#                                       tight to its feasibility tolerance only: 5.6e-6 at 4000 x 40000)
info = {"generator": "splitmix64 dense LP, SURVEY.md 8d (ellp_amd/synth.py::dense_lp)", "seed": seed, "m": m, "n": n,
        "solver": "scipy.optimize.linprog(method='highs-ds'), feasibility tolerances 1e-10", "seconds": round(dt, 1)}
xs = np.linalg.solve(A[np.ix_(tight, sup)], b[tight])  # x_S = A[tight, S]^-1 b[tight] in double precision
x2 = np.zeros(n)
x2[sup] = xs
y = np.zeros(m)
y[tight] = np.linalg.solve(A[np.ix_(tight, sup)].T, c[sup])
rc = c - A.T @ y
# the polished point must be a feasible vertex with nonnegative reduced costs: then it IS the optimum
assert xs.min() > 0 and (A @ x2 - b).max() < 1e-9 and rc.min() > -1e-9 and y.max() < 1e-9, "not a nondegenerate optimal vertex"
info["polished"] = True
info["polish_max_change"] = float(np.abs(x2 - x).max())
info["dual_check"] = {"min_reduced_cost": float(rc.min()), "max_row_multiplier": float(y.max())}
x = x2
info["objective"] = float(c @ x)
info["highs_objective"] = float(r.fun)
info["support"] = [int(j) for j in np.flatnonzero(x != 0.0)]
info["values"] = [float(x[j]) for j in info["support"]]
info["max_row_violation"] = float(max(0.0, (A @ x - b).max()))
info["min_x"] = float(x.min())
out = sys.argv[4] if len(sys.argv) > 4 else f"tests/golden/synth_optimum_{seed}_{m}x{n}.json"
with open(out, "w") as f:
    json.dump(info, f)
print({k: v for k, v in info.items() if k not in ("support", "values")}, len(info["support"]))
