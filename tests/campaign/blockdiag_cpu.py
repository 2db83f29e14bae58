#!/usr/bin/env python3
"""tests/campaign/blockdiag_cpu.py NAME COPIES ORDERS [seed] — CPU-only study behind tests/test_gpu_blockdiag.py:
block-diagonal replications of a netlib LP in random variable / row orders, through the oracle's
LU-per-iteration loops (the reference's arithmetic) and through its explicit-inverse loops (the large
engine's algorithm on the host): how long the oracle takes and how often the explicit inverse ends
differently."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps  # noqa: E402
from oracle import ellp_oracle as eo  # noqa: E402


def run_primal(prob, binv, max_iter):
    p1, err = eo.primal_phase1(prob)
    v = p1.view()
    its = []
    if binv:
        st, it, msg, _ = eo.primal_binv_solve_with_initial(v, max_iter, threads=1)
    else:
        st, it, msg = eo.primal_solve_with_initial(v, max_iter)
    its.append(it)
    if st != eo.OPTIMAL or abs(v.obj()) > 1e-9:
        return ("p1", st, v.obj(), its, msg)
    p1.store_point(v)
    v2 = eo.primal_phase2(p1).view()
    if binv:
        st, it, msg, _ = eo.primal_binv_solve_with_initial(v2, max_iter, threads=1)
    else:
        st, it, msg = eo.primal_solve_with_initial(v2, max_iter)
    its.append(it)
    return ("p2", st, v2.obj(), its, msg)


def run_dual(prob, binv, max_iter):
    d1, err = eo.dual_phase1(prob)
    v = d1.view()
    its = []
    if binv:
        st, it, msg, _ = eo.dual_binv_solve_with_initial(v, max_iter, threads=1)
    else:
        st, it, msg = eo.dual_solve_with_initial(v, max_iter)
    its.append(it)
    if st != eo.OPTIMAL:
        return ("d1", st, None, its, msg)
    d1.store_point(v)
    d2, err2 = eo.dual_phase2(d1)
    if d2 is None or err2:
        return ("d2-setup", -99, None, its, "")
    v2 = d2.view()
    if binv:
        st, it, msg, _ = eo.dual_binv_solve_with_initial(v2, max_iter, threads=1)
    else:
        st, it, msg = eo.dual_solve_with_initial(v2, max_iter)
    its.append(it)
    return ("d2", st, v2.obj(), its, msg)


def main():
    name, copies, orders = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
    want = copies * ka["obj"]
    rng = np.random.default_rng(1000 * copies + seed)
    out = []
    for t in range(orders):
        fx = permuted_fixture(base, rng)
        prob = eo.Problem.from_fixture(fx)
        row = {"trial": t}
        for solver, fn in (("primal", run_primal), ("dual", run_dual)):
            t0 = time.time()
            a = fn(prob, False, 200000)
            t1 = time.time()
            b = fn(prob, True, 200000)
            t2 = time.time()
            ok_a = a[1] == eo.OPTIMAL and a[2] is not None and abs(a[2] / want - 1) < 1e-6
            ok_b = b[1] == eo.OPTIMAL and b[2] is not None and abs(b[2] / want - 1) < 1e-6
            row[solver] = {"lu": [a[0], a[1], a[2], a[3], round(t1 - t0, 2), ok_a],
                           "binv": [b[0], b[1], b[2], b[3], round(t2 - t1, 2), ok_b]}
        print(json.dumps(row), flush=True)
        out.append(row)
    for solver in ("primal", "dual"):
        n_lu = sum(r[solver]["lu"][5] for r in out)
        n_bi = sum(r[solver]["binv"][5] for r in out)
        same = sum((r[solver]["lu"][1], r[solver]["lu"][5]) == (r[solver]["binv"][1], r[solver]["binv"][5]) for r in out)
        print(solver, "oracle ok", n_lu, "explicit inverse ok", n_bi, "same ending", same, "of", len(out))


if __name__ == "__main__":
    main()
