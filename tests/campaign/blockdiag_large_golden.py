#!/usr/bin/env python3
"""tests/campaign/blockdiag_large_golden.py [workers] — makes tests/golden/blockdiag_large.json: how the ORACLE (the
reference's LU-per-iteration loops, oracle/ellp_oracle.c) and the oracle's restatement of the certified hybrid end on
block-diagonal replications of the reference's netlib fixtures ABOVE 512 rows, in the 30 variable / constraint orders the
GPU suite runs (tests/test_gpu_hybrid.py draws the same orders from the same generator).  The oracle needs 0.5-5 minutes
per solve at these sizes (one LU of a 560-1,036-row basis per iteration), which is why its outcomes are committed as a
fixture instead of being recomputed in the suite; CPU only.

    ADLITTLE x 10  (560 rows), ADLITTLE x 18 (1,008 rows), BLEND x 14 (1,036 rows)

Per order and solver: [stage, status, objective or null, iterations per phase]; stage = the last phase entered
(p1 / p2, d1 / d2, d2-setup = the reference's DualPhase2::from panicked)."""
import json
import os
import sys
import zlib
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps  # noqa: E402

CASES = [("adlittle", 10, 30), ("adlittle", 18, 30), ("blend", 14, 30)]
# SMALL=1: the three problems of 148-336 rows, 60 orders each, into tests/golden/blockdiag_orders.json (seconds per solve)
if os.environ.get("SMALL") == "1":
    CASES = [("adlittle", 3, 60), ("blend", 2, 60), ("adlittle", 6, 60)]
# CASES=blend:25:20 SOLVERS=primal MODES=lu: other problems / a subset of the runs (above 1,024 rows the oracle's dual needs hours)
if os.environ.get("CASES"):
    CASES = [(c.split(":")[0], int(c.split(":")[1]), int(c.split(":")[2])) for c in os.environ["CASES"].split(",")]
SOLVERS = os.environ.get("SOLVERS", "primal,dual").split(",")
MODES = os.environ.get("MODES", "hybrid,lu").split(",")
MAX_ITER = 2000000


def solve(prob, dual, mode):
    from oracle import ellp_oracle as eo

    def run(v):
        if mode == "lu":
            f = eo.dual_solve_with_initial if dual else eo.primal_solve_with_initial
            st, it, msg = f(v, MAX_ITER)
            return st, it, None
        f = eo.dual_hybrid_solve_with_initial if dual else eo.primal_hybrid_solve_with_initial
        st, it, msg, cnt = f(v, MAX_ITER)
        return st, it, cnt

    if not dual:
        p1, err = eo.primal_phase1(prob)
        v = p1.view()
        st, it, c1 = run(v)
        if st != eo.OPTIMAL or not (-1e-10 < v.obj() < 1e-10):  # primal_simplex_solver.rs:42-50 (assert obj > -EPS; obj < EPS)
            return ["p1", st, v.obj(), [it], [c1]]
        p1.store_point(v)
        v2 = eo.primal_phase2(p1).view()
        st2, it2, c2 = run(v2)
        return ["p2", st2, v2.obj(), [it, it2], [c1, c2]]
    d1, err = eo.dual_phase1(prob)
    v = d1.view()
    st, it, c1 = run(v)
    if st != eo.OPTIMAL:
        return ["d1", st, None, [it], [c1]]
    d1.store_point(v)
    d2, err2 = eo.dual_phase2(d1)
    if d2 is None or err2:
        return ["d2-setup", -99, None, [it], [c1]]
    v2 = d2.view()
    st2, it2, c2 = run(v2)
    return ["d2", st2, v2.obj(), [it, it2], [c1, c2]]


def job(args):
    name, copies, trial, fx, dual, mode = args
    from oracle import ellp_oracle as eo
    import time
    t0 = time.time()
    out = solve(eo.Problem.from_fixture(fx), dual, mode)
    return name, copies, trial, dual, mode, out, round(time.time() - t0, 1)


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    only = os.environ.get("ONLY")  # e.g. "adlittle:10"
    jobs = []
    for name, copies, orders in CASES:
        if only and only != f"{name}:{copies}":
            continue
        ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
        base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
        rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
        for t in range(orders):
            fx = permuted_fixture(base, rng)
            for dual in (False, True):
                for mode in ("hybrid", "lu"):
                    if ("dual" if dual else "primal") in SOLVERS and mode in MODES:
                        jobs.append((name, copies, t, fx, dual, mode))
    jobs.sort(key=lambda j: (j[5] == "lu", j[1]))  # the cheap hybrid runs first
    path = os.path.join(GOLDEN, "blockdiag_orders.json" if os.environ.get("SMALL") == "1" else "blockdiag_large.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    todo = [j for j in jobs if f"{j[2]}" not in res.get(f"{j[0]}x{j[1]}", {}).get(("dual" if j[4] else "primal") + ":" + j[5], {})]
    print(len(jobs), "jobs,", len(todo), "to do", flush=True)
    with Pool(workers) as pool:
        for name, copies, t, dual, mode, out, secs in pool.imap_unordered(job, todo):
            key = f"{name}x{copies}"
            sub = res.setdefault(key, {}).setdefault(("dual" if dual else "primal") + ":" + mode, {})
            sub[str(t)] = out[:4] if mode == "lu" else out
            print(key, t, "dual" if dual else "primal", mode, out[:4], secs, "s", flush=True)
            with open(path + ".tmp", "w") as f:
                json.dump(res, f, indent=0, sort_keys=True)
            os.replace(path + ".tmp", path)


if __name__ == "__main__":
    main()
