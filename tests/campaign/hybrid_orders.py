#!/usr/bin/env python3
"""tests/campaign/hybrid_orders.py — tests/test_gpu_hybrid.py at full length, without stopping at the first failure: every
order of every block-diagonal netlib replication through the default engine (the certified hybrid), at the seam and through
the user API; per problem how many orders reach the pinned optimum, how the others end, the hybrid's counters (guarded pivots
handed to the exact kernel, terminal statuses examined, of those not confirmed) and the wall time.  Writes
gpurun_out/hybrid_orders.json (committed as profiles/r04_hybrid_orders.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_hybrid as T  # noqa: E402


def main():
    gold = T._golden()
    out = {"seam": [], "api": []}
    for name, copies, orders in T.SEAM_CASES:
        key = f"{name}x{copies}"
        tally, rec = {}, {"primal_ok": 0, "dual_ok": 0, "other": []}
        t0 = time.time()
        for trial, fx, want in T._orders(name, copies, orders):
            res = T.seam_order(fx, want, tally)
            for solver, (ok, what) in res.items():
                if ok:
                    rec[solver + "_ok"] += 1
                else:
                    rec["other"].append({"trial": trial, "solver": solver, "what": [str(w)[:120] for w in what],
                                         "oracle_reached_optimum": T.oracle_reached_optimum(gold, key, solver, trial, want)})
        rec.update(problem=key, orders=orders, seconds=round(time.time() - t0, 1), counters=tally)
        print(json.dumps(rec), flush=True)
        out["seam"].append(rec)
    import ellp_amd
    for name, copies, orders, orders_dual, _ in T.API_CASES:
        key = f"{name}x{copies}"
        rec = {"primal_ok": 0, "dual_ok": 0, "other": []}
        dual_orders = orders  # all of them (the suite runs fewer dual orders of ADLITTLE x 18: the redos take 20 s each)
        rec["dual_orders"] = dual_orders
        t0 = time.time()
        for trial, fx, want in T._orders(name, copies, orders):
            prob = ellp_amd.Problem.from_fixture(fx)
            for solver, cls in (("primal", ellp_amd.PrimalSimplexSolver), ("dual", ellp_amd.DualSimplexSolver)):
                if solver == "dual" and trial >= dual_orders:
                    continue
                try:
                    r = cls.new(400000).solve(prob.clone())
                    ok = r.kind == ellp_amd.SolverResult.Optimal and abs(r.solution.obj() / want - 1.0) < 1e-9
                    if ok:
                        v = T.fixture_violation(fx, r.solution.x())
                        ok = v[0] < 1e-8 and v[1] < 1e-8
                    what = [r.kind, list(r.iters)]
                except (RuntimeError, ellp_amd.EllPError) as ex:
                    ok, what = False, [repr(ex)[:120]]
                if ok:
                    rec[solver + "_ok"] += 1
                else:
                    rec["other"].append({"trial": trial, "solver": solver, "what": what,
                                         "oracle_reached_optimum": T.oracle_reached_optimum(gold, key, solver, trial, want)})
        rec.update(problem=key, orders=orders, seconds=round(time.time() - t0, 1))
        print(json.dumps(rec), flush=True)
        out["api"].append(rec)
    path = os.path.join(ROOT, "gpurun_out", "hybrid_orders.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
