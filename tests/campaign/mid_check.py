#!/usr/bin/env python3
"""tests/campaign/mid_check.py — quick look at the exact mid-size kernel: bits against the oracle and time per iteration"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps  # noqa: E402
from oracle import ellp_oracle as eo  # noqa: E402
from ellp_amd import _engine as E  # noqa: E402


def flat(v):
    return E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)


def run(tag, v, which, iters):
    ov = v.copy()
    fo = eo.primal_solve_with_initial if which == "primal" else eo.dual_solve_with_initial
    t0 = time.time()
    st_o, it_o, _ = fo(ov, iters)
    t1 = time.time()
    fp = flat(v)
    fg = E.primal_solve_with_initial if which == "primal" else E.dual_solve_with_initial
    t2 = time.time()
    st_g, stats, msg = fg(fp, E.default_opts(max_iter=iters, pipeline=3))
    t3 = time.time()
    same = (st_g == st_o and stats.iters == it_o and np.array_equal(fp.B, ov.B) and fp.x.tobytes() == ov.x.tobytes())
    print(f"{tag} {which} m={v.m} nN={v.nN}: oracle st {st_o} it {it_o} {1e3*(t1-t0)/max(it_o,1):.3f} ms/it | gpu st {st_g} it {stats.iters} "
          f"{1e3*stats.t_loop_s/max(stats.iters,1):.3f} ms/it (wall {t3-t2:.2f}s) | {'EQUAL' if same else 'DIFFERENT'} {msg}", flush=True)
    if not same and st_g == st_o:
        print("   max|dx|", np.abs(fp.x - ov.x).max(), "B equal", np.array_equal(fp.B, ov.B))
    return same


ok = True
for m, n, it in ((129, 300, 50), (150, 380, 200), (300, 700, 100), (513, 1100, 30), (700, 1500, 20), (1024, 2000, 10)):
    p1, _ = eo.primal_phase1(eo.synth_problem(20260301 + m, m, n))
    ok &= run("dense", p1.view(), "primal", it)
for m, n, it in ((160, 300, 100), (600, 1000, 20)):
    p1, _ = eo.dual_phase1(eo.synth_problem(20260301 + m, m, n))
    ok &= run("dense", p1.view(), "dual", it)
for name, copies in (("adlittle", 3), ("blend", 2), ("adlittle", 6)):
    ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
    fx = permuted_fixture(base, np.random.default_rng(7 + copies))
    prob = eo.Problem.from_fixture(fx)
    p1, _ = eo.primal_phase1(prob)
    ok &= run(f"{name}x{copies}", p1.view(), "primal", 100000)
    d1, _ = eo.dual_phase1(prob)
    ok &= run(f"{name}x{copies}", d1.view(), "dual", 100000)
print("ALL EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
