#!/usr/bin/env python3
"""tests/campaign/hybrid_cpu.py NAME COPIES ORDERS [K] [resync] — CPU-only study of the certified hybrid (round 4, review item 1):
the oracle's explicit-inverse loop (the large engine's algorithm on the host) runs until it reports a terminal
status; the oracle's LU-per-iteration loop (the reference's arithmetic) then runs up to K iterations from that basis.
If its first iteration ends the same way the status is certified; if it pivots on, the explicit-inverse loop takes
over again from where the exact loop stopped.  Counts, per solver, how often the plain explicit-inverse loop, the
hybrid and the oracle reach the pinned optimum (copies x tests/problems/mod.rs:657-674), and how often the
certificate disagreed."""
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

TERMINAL = (eo.OPTIMAL, eo.INFEASIBLE, eo.UNBOUNDED)


def resync(v, dual, mode=1):
    """mode 1: x_B (and for the dual y, d) from a fresh factorisation of the current basis; mode 2: only y, d"""
    A = v.A_matrix()
    B = v.B[:v.nB]
    N = v.N[:v.nN]
    AB = A[:, B]
    if mode in (1, 3):
        try:
            xb = np.linalg.solve(AB, v.b - A[:, N] @ v.x[N])
        except np.linalg.LinAlgError:
            return False
        v.x[B] = xb
    if dual and mode != 3:
        y = np.linalg.solve(AB.T, v.c[B])
        d = v.c[:v.n] - A.T @ y
        d[B] = 0.0
        v.y[:] = y
        v.d[:v.n] = d
    return True


def hybrid(v, dual, K, do_resync, refresh, stats, max_iter=400000):
    fast = eo.dual_binv_solve_with_initial if dual else eo.primal_binv_solve_with_initial
    exact = eo.dual_solve_with_initial if dual else eo.primal_solve_with_initial
    total = 0
    rounds = 0
    while total < max_iter:
        eo.set_continuation(rounds > 0)
        st, it, msg, _ = fast(v, max_iter - total, threads=1, refresh=refresh)
        eo.set_continuation(True)
        total += it
        rounds += 1
        if st == eo.MAXITER:
            return st, total, msg
        if do_resync:
            resync(v, dual, do_resync)
        if st == eo.NEED_EXACT:  # a suspicious pivot: the exact loop takes this iteration and K - 1 more
            st2, it2, msg2 = exact(v, K)
            total += it2
            stats["guards"] = stats.get("guards", 0) + 1
            if st2 != eo.MAXITER:
                return st2, total, msg2
            continue
        st2, it2, msg2 = exact(v, K)
        total += it2
        stats["certs"] += 1
        if st2 == st and it2 <= 1:
            return st2, total, msg2
        stats["disagree"] += 1
        stats.setdefault("how", []).append((st, st2, it2))
        if st2 != eo.MAXITER:
            return st2, total, msg2
        if rounds > 100000:
            return -98, total, "hybrid: too many hand-overs"
    return eo.MAXITER, total, ""


def solve(prob, dual, mode, K, do_resync, refresh, stats):
    """mode: 'lu' | 'binv' | 'hybrid'; returns (stage, status, obj, iters)"""
    def run(v):
        if mode == "lu":
            f = eo.dual_solve_with_initial if dual else eo.primal_solve_with_initial
            st, it, msg = f(v, 400000)
            return st, it
        if mode == "binv":
            f = eo.dual_binv_solve_with_initial if dual else eo.primal_binv_solve_with_initial
            st, it, msg, _ = f(v, 400000, threads=1, refresh=refresh)
            return st, it
        if mode == "chybrid":  # the policy as restated in oracle/ellp_oracle.c (what the engine's tests check against)
            f = eo.dual_hybrid_solve_with_initial if dual else eo.primal_hybrid_solve_with_initial
            st, it, msg, cnt = f(v, 400000, K=K, refresh=refresh, guard_abs=float(os.environ.get("GUARD_ABS", "1e-7")))
            stats["guards"] = stats.get("guards", 0) + cnt[0]
            stats["certs"] += cnt[1]
            stats["disagree"] += cnt[2]
            return st, it
        st, it, msg = hybrid(v, dual, K, do_resync, refresh, stats)
        eo.set_continuation(False)
        return st, it

    if not dual:
        p1, err = eo.primal_phase1(prob)
        v = p1.view()
        st, it = run(v)
        if st != eo.OPTIMAL or abs(v.obj()) > 1e-9:
            return ("p1", st, v.obj(), [it])
        p1.store_point(v)
        v2 = eo.primal_phase2(p1).view()
        st2, it2 = run(v2)
        return ("p2", st2, v2.obj(), [it, it2])
    d1, err = eo.dual_phase1(prob)
    v = d1.view()
    st, it = run(v)
    if st != eo.OPTIMAL:
        return ("d1", st, None, [it])
    d1.store_point(v)
    if not (d1.dual_obj() > -1e-10):  # dual_simplex_solver.rs:45-50: "dual infeasible"
        return ("d1-obj", st, d1.dual_obj(), [it])
    d2, err2 = eo.dual_phase2(d1)
    if d2 is None or err2:
        return ("d2-setup", -99, None, [it])
    v2 = d2.view()
    st2, it2 = run(v2)
    return ("d2", st2, v2.obj(), [it, it2])


def main():
    name, copies, orders = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    do_resync = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    refresh = int(os.environ.get("REFRESH", "64"))
    modes = os.environ.get("MODES", "binv,hybrid").split(",")
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
    want = copies * ka["obj"]
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()) if os.environ.get('SUITE_ORDERS') else 1000 * copies)
    tally = {}
    eo.set_binv_guard(float(os.environ.get("GUARD_REL", "0")), float(os.environ.get("GUARD_ABS", "0")))
    eo.set_binv_zero_tol(float(os.environ.get("ZERO_TOL", "0")))
    for t in range(orders):
        fx = permuted_fixture(base, rng)
        prob = eo.Problem.from_fixture(fx)
        row = {"trial": t}
        for dual in [w == 'dual' for w in os.environ.get('SOLVERS', 'primal,dual').split(',')]:
            for mode in modes:
                stats = {"certs": 0, "disagree": 0}
                t0 = time.time()
                a = solve(prob, dual, mode, K, do_resync, refresh, stats)
                ok = a[1] == eo.OPTIMAL and a[2] is not None and abs(a[2] / want - 1) < 1e-9
                key = ("dual" if dual else "primal") + ":" + mode
                tl = tally.setdefault(key, {"ok": 0, "bad": 0, "certs": 0, "disagree": 0})
                tl["ok" if ok else "bad"] += 1
                tl["certs"] += stats["certs"]
                tl["disagree"] += stats["disagree"]
                tl["guards"] = tl.get("guards", 0) + stats.get("guards", 0)
                tl["iters"] = tl.get("iters", 0) + sum(a[3])
                row[key] = [a[0], a[1], a[2], a[3], ok, stats, round(time.time() - t0, 2)]
        print(json.dumps(row), flush=True)
    print(json.dumps(tally))


if __name__ == "__main__":
    main()
