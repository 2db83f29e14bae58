#!/usr/bin/env python3
"""tools/hybrid_dual_probe.py NAME COPIES ORDERS [refactor_period] — GPU: dual phase 1 of block-diagonal netlib replications
through the default engine (the certified hybrid); how the phase ends, the phase objective from the carried (y, d)
(what dual_simplex_solver.rs:45-50 tests against EPS) and from a d recomputed on the host from the end basis."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ellp_amd  # noqa: E402
from ellp_amd import _engine as E  # noqa: E402
from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps  # noqa: E402


def dual_obj(kind, lb, ub, b, y, d):
    o = float(b @ y)
    for k, l, u, di in zip(kind, lb, ub, d):
        if k == 1:
            o += l * di
        elif k == 2:
            o += u * di
        elif k == 3:
            o += (l if di > 0 else u) * di
        elif k == 4:
            o += l * di
    return o


def main():
    name, copies, orders = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    period = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
    rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
    bad = 0
    for t in range(orders):
        fx = permuted_fixture(base, rng)
        ph = ellp_amd.Problem.from_fixture(fx)._debug_phase1("dual")
        fp = E.FlatProblem(ph["m"], ph["n"], ph["n_c"], ph["A"], ph["c"], ph["b"], ph["kind"], ph["lb"], ph["ub"], ph["x"],
                           ph["B"], ph["N"], ph["Nb"], ph["y"], ph["d"])
        kw = dict(max_iter=None)
        if period:
            kw["refactor_period"] = period
        eng = E.Engine(E.ENGINE_DUAL, fp, E.default_opts(**kw))
        st, stats, msg = eng.run(2000000)
        eng.read_point()
        c = eng.counters()
        eng.close()
        m, n = fp.m, fp.n
        A = fp.A.reshape((n, m)).T
        o_carried = dual_obj(fp.kind, fp.lb, fp.ub, fp.b, fp.y, fp.d)
        y = np.linalg.solve(A[:, fp.B].T, fp.c[fp.B])
        d = fp.c[:n] - A.T @ y
        d[fp.B] = 0.0
        o_fresh = dual_obj(fp.kind, fp.lb, fp.ub, fp.b, y, d)
        ok = st == 0 and o_carried > -1e-10
        bad += 0 if ok else 1
        print(json.dumps(dict(trial=t, status=st, iters=int(stats.iters), obj_carried=o_carried, obj_fresh=o_fresh,
                              max_d_diff=float(np.abs(d - fp.d[:n]).max()), guards=c["hybrid_guards"], certs=c["hybrid_certs"],
                              disagreed=c["hybrid_disagreed"], refreshes=c["refreshes"], ok=bool(ok))), flush=True)
    print("bad", bad, "of", orders)


if __name__ == "__main__":
    main()
