#!/usr/bin/env python3
"""tools/unit_columns_ab.py — config 3 and config 5 with and without the unit-column shortcut of the primal pricing pass
(ellp_opts.flags bit 0 = stream everything): pivots/s over a window from the same start, and the pricing kernel's
time from the engine's event brackets."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ellp_amd import _engine as E, synth  # noqa: E402

out = []
for (m, n, seed, warm, steps) in ((2000, 5000, 20260301, 300, 3000), (4000, 40000, 20260305, 100, 1000)):
    f = synth.primal_phase1_flat(seed, m, n)
    for flags in (1, 0):
        fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"])
        eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, flags=flags))
        eng.run(warm)
        t0 = time.perf_counter()
        st, stats, msg = eng.run(steps)
        dt = time.perf_counter() - t0
        eng.read_point()
        eng.close()
        fp2 = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"])
        engp = E.Engine(E.ENGINE_PRIMAL, fp2, E.default_opts(max_iter=None, flags=flags, profile=1))
        engp.run(warm)
        st, ps, _ = engp.run(200)
        pd = ps.as_dict()
        engp.close()
        price_us = 1e3 * pd["kernel_ms"]["price"] / pd["kernel_calls"]["price"]
        out.append({"m": m, "n": n, "unit_column_shortcut": flags == 0, "pivots_per_s": round(steps / dt, 1),
                    "us_per_pivot": round(1e6 * dt / steps, 2), "pricing_kernel_us": round(price_us, 2),
                    "basis_crc": int(np.bitwise_xor.reduce(fp.B))})
        print(json.dumps(out[-1]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "unit_columns_ab.json"), "w"), indent=1)
