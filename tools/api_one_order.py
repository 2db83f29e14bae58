#!/usr/bin/env python3
"""tools/api_one_order.py NAME COPIES TRIAL SOLVER MAX_ITER — one order of a block-diagonal replication through the user API"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ellp_amd
from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
name, copies, trial, solver, max_iter = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
for t in range(trial + 1):
    fx = permuted_fixture(base, rng)
cls = ellp_amd.PrimalSimplexSolver if solver == "primal" else ellp_amd.DualSimplexSolver
t0 = time.time()
try:
    r = cls.new(max_iter if max_iter > 0 else None).solve(ellp_amd.Problem.from_fixture(fx))
    print(r.kind, r.iters, r.solution.obj() if r.solution else r.obj, "want", copies * ka["obj"], round(time.time() - t0, 2), "s")
except Exception as ex:
    print("exception", repr(ex)[:200], round(time.time() - t0, 2), "s")
