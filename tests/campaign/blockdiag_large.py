#!/usr/bin/env python3
"""tests/campaign/blockdiag_large.py [copies orders] — replicated netlib LPs ABOVE 512 rows (ADLITTLE x 10: m = 560 by default), random
variable / constraint orders, both solvers at the seam: how the oracle ends each phase, how the default engine at that size
(explicit inverse) ends it, and how the exact LU-per-iteration kernel does when ELLP_MID_AUTO_MAX=1024 makes it the choice —
status, iterations, objective, seconds.  (tests/test_gpu_blockdiag.py pins the sizes up to 512 rows.)"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
from oracle import ellp_oracle as eo
from test_gpu_small import flat
from ellp_amd import _engine as E

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 10
orders = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ka = next(p for p in known_answers()["netlib"] if p["name"] == "adlittle")
base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
want = copies * ka["obj"]
rng = np.random.default_rng(zlib.crc32(f"adlittlex{copies}".encode()))
eo.set_setup_threads(8) if hasattr(eo, "set_setup_threads") else None

def engine_phase(view, which, exact):
    if exact: os.environ["ELLP_MID_AUTO_MAX"] = "1024"
    else: os.environ.pop("ELLP_MID_AUTO_MAX", None)
    fp = flat(view)
    f = E.primal_solve_with_initial if which == "primal" else E.dual_solve_with_initial
    t0 = time.time()
    st, stats, err = f(fp, E.default_opts(max_iter=400000))
    return fp, int(st), int(stats.iters), time.time() - t0

out = []
for trial in range(orders):
    prob = eo.Problem.from_fixture(permuted_fixture(base, rng))
    rec = {"trial": trial}
    for which in ("primal", "dual"):
        p1, err = (eo.primal_phase1 if which == "primal" else eo.dual_phase1)(prob)
        m = p1.view().m
        # oracle, both phases
        fo = eo.primal_solve_with_initial if which == "primal" else eo.dual_solve_with_initial
        start_view = p1.view().copy()  # the phase-1 START (store_point below moves p1 to its end)
        t0 = time.time()
        ov = p1.view().copy()
        st1, it1, _ = fo(ov, 400000)
        res_o = {"phase1": [int(st1), int(it1)]}
        ok1 = st1 == eo.OPTIMAL and (which == "dual" or abs(ov.obj()) < 1e-9)
        if ok1:
            p1.store_point(ov)
            p2 = eo.primal_phase2(p1) if which == "primal" else eo.dual_phase2(p1)[0]
            if p2 is not None:
                ov2 = p2.view().copy()
                st2, it2, _ = fo(ov2, 400000)
                res_o["phase2"] = [int(st2), int(it2), ov2.obj() / want - 1.0 if st2 == eo.OPTIMAL else None]
        res_o["s"] = round(time.time() - t0, 1)
        rec[which + "_oracle"] = res_o
        # engine: phase 1 from the same arrays, phase 2 from the ORACLE's phase-1 end point (so that the two phase-2 runs are comparable)
        for tag, exact in (("explicit", False), ("exact", True)):
            fp, st, it, dt = engine_phase(start_view, which, exact)
            r = {"phase1": [st, it, round(dt, 1)], "same_basis_as_oracle": bool(np.array_equal(np.sort(fp.B), np.sort(ov.B[:m])))}
            if ok1 and "phase2" in res_o:
                fp2, st2g, it2g, dt2 = engine_phase(p2.view(), which, exact)
                r["phase2"] = [st2g, it2g, round(dt2, 1), (fp2.obj() / want - 1.0) if st2g == E.OPTIMAL else None]
            rec[which + "_" + tag] = r
    rec["rows"] = int(m)
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"blockdiag_large_x{copies}.json"), "w"), indent=1)
