#!/usr/bin/env python3
"""tests/campaign/trace_order.py FAILURES.json INDEX [PHASE] — step a failing netlib order of
tests/campaign/netlib_orders.py through the dual loop one iteration at a time, oracle (CPU) beside engine
(GPU), and print where they part and what B^-1 looks like there."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import GOLDEN, known_answers, read_mps  # noqa: E402
from oracle import ellp_oracle as eo  # noqa: E402
from ellp_amd import _engine as E  # noqa: E402


def fixture(case):
    ka = next(p for p in known_answers()["netlib"] if p["name"] == case["name"])
    base = read_mps(os.path.join(GOLDEN, ka["file"]))
    perm = case["var_perm"]
    inv = np.empty(len(perm), dtype=int)
    inv[np.asarray(perm)] = np.arange(len(perm))
    rows = [base["constraints"][i] for i in case["row_perm"]]
    return {"vars": [base["vars"][j] for j in perm],
            "constraints": [[[[int(inv[j]), a] for j, a in coeffs], op, rhs] for coeffs, op, rhs in rows]}, ka


def flat(v):
    return E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)


def main():
    d = json.load(open(sys.argv[1]))
    cases = d["failures"] if "failures" in d else d["orders"]
    case = cases[int(sys.argv[2])]
    fx, ka = fixture(case)
    prob = eo.Problem.from_fixture(fx)
    d1, err = eo.dual_phase1(prob)
    view = d1.view()
    phase = sys.argv[3] if len(sys.argv) > 3 else "auto"
    ov = view.copy()
    st1, it1, _ = eo.dual_solve_with_initial(ov, 20000)
    print("oracle dual1:", st1, it1)
    fp = flat(view)
    stg, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=20000))
    print("engine dual1:", stg, stats.iters, msg)
    if phase == "dual2" or (phase == "auto" and stg == st1 == 0):
        d1.store_point(ov)
        d2, err2 = eo.dual_phase2(d1)
        view = d2.view()
        ov = view.copy()
        st2, it2, _ = eo.dual_solve_with_initial(ov, 20000)
        print("oracle dual2:", st2, it2)
    # step both
    A = np.asarray(view.A).reshape(view.n, view.m).T  # m x n (column-major storage)
    eng = E.Engine(E.ENGINE_DUAL, flat(view), E.default_opts(max_iter=None))
    prevB = np.array(view.B)
    k = 0
    while k < 5000:
        k += 1
        st, stats, msg = eng.run(1)
        eng.read_point()
        o = view.copy()
        so, io, _ = eo.dual_solve_with_initial(o, k)
        B_e, B_o = eng.fp.B, np.asarray(o.B)
        same = np.array_equal(B_e, B_o)
        AB = A[:, B_e]
        cond = np.linalg.cond(AB)
        res = eng.inverse_residual()
        cnt = eng.counters()
        if not same or st != E.MAXITER or res > 1e-9 or k % 25 == 0:
            ch_e = np.nonzero(B_e != prevB)[0]
            print(f"it {k}: engine st={st} iters={stats.iters} same_basis={same} cond={cond:.2e} resid={res:.2e} "
                  f"maint={cnt['maint_requests']} refresh={cnt['refreshes']} rebuild={cnt['rebuilds']} changed_pos={ch_e.tolist()} {msg}")
        if not same:
            ch_o = np.nonzero(B_o != prevB)[0]
            print("   oracle changed", ch_o.tolist(), "->", B_o[ch_o].tolist(), "| engine ->", B_e[np.nonzero(B_e != prevB)[0]].tolist())
            Wt = eng.tap(E.TAP_BINV, view.m * view.m).reshape(view.m, view.m)
            print("   max|W| engine", np.abs(Wt).max(), " cond(A_B oracle)", np.linalg.cond(A[:, B_o]))
            dx = np.abs(eng.fp.x - np.asarray(o.x)).max()
            print("   max|x_e - x_o|", dx)
            break
        if st != E.MAXITER:
            break
        prevB = B_e.copy()
    eng.close()


if __name__ == "__main__":
    main()
