#!/usr/bin/env python3
"""tools/eta_store_fraction.py — review item 7 (round 4): what fraction of the rows of B^-1 does an eta update really change?
Row i of E W equals row i of W whenever d_i = 0 exactly, so a copy-on-write eta pass could skip the store (and the load) of
those rows.  Measured: the fraction of basic rows with d_i != 0 per pivot, sampled over windows at several stages of a solve,
for config 3, config 5 (first stages) and ADLITTLE x 10.  Writes gpurun_out/eta_store_fraction.json."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ellp_amd import _engine as E  # noqa: E402
from ellp_amd import synth  # noqa: E402


def sample(eng, m, stages, window):
    out = []
    done = 0
    for upto in stages:
        if upto > done:
            st, stats, msg = eng.run(upto - done)
            done = upto
            if st != E.MAXITER:
                out.append({"after": int(stats.iters), "ended": int(st)})
                break
        fr = []
        for _ in range(window):
            st, stats, msg = eng.run(1)
            done += 1
            if st != E.MAXITER:
                break
            d = eng.tap(E.TAP_D, m)
            fr.append(float(np.count_nonzero(d)) / m)
        if fr:
            out.append({"after": done, "pivots": len(fr), "rows_changed_mean": round(float(np.mean(fr)), 4),
                        "rows_changed_min": round(float(np.min(fr)), 4), "rows_changed_max": round(float(np.max(fr)), 4)})
    return out


def main():
    res = {}
    for tag, seed, m, n, stages in (("config3", 20260301, 2000, 5000, [0, 2000, 20000, 100000, 300000, 600000]),
                                    ("config5", 20260305, 4000, 40000, [0, 4000, 40000, 200000])):
        f = synth.primal_phase1_flat(seed, m, n)
        fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"])
        eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=1))  # three launches: d of every pivot is tapped
        res[tag] = sample(eng, m, stages, 60)
        eng.close()
        print(tag, json.dumps(res[tag]), flush=True)
    # ADLITTLE x 10 (560 rows, sparse), primal phase 1 from the host mirror's arrays
    import ellp_amd
    from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
    ka = next(p for p in known_answers()["netlib"] if p["name"] == "adlittle")
    base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), 10)
    fx = permuted_fixture(base, np.random.default_rng(zlib.crc32(b"adlittlex10")))
    ph = ellp_amd.Problem.from_fixture(fx)._debug_phase1("primal")
    fp = E.FlatProblem(ph["m"], ph["n"], ph["n_c"], ph["A"], ph["c"], ph["b"], ph["kind"], ph["lb"], ph["ub"], ph["x"], ph["B"], ph["N"], ph["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=1))
    res["adlittlex10_phase1"] = sample(eng, ph["m"], [0, 200, 500, 800], 60)
    eng.close()
    print("adlittlex10", json.dumps(res["adlittlex10_phase1"]), flush=True)
    path = os.path.join(ROOT, "gpurun_out", "eta_store_fraction.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(res, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
