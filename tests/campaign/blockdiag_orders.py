#!/usr/bin/env python3
"""tests/campaign/blockdiag_orders.py [orders] — tests/test_gpu_blockdiag.py at full length (60 orders of each of the
three replicated netlib LPs, both solvers, default options), without stopping at the first difference; writes
gpurun_out/blockdiag_orders.json (committed as profiles/r03_blockdiag_orders.json)."""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps  # noqa: E402
import test_gpu_blockdiag as T  # noqa: E402


def main():
    orders = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    out = {"orders_per_problem": orders, "problems": []}
    for name, copies, _ in T.CASES:
        ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
        base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
        rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
        tally = {"primal_ok": 0, "primal_reference_fails": 0, "dual_ok": 0, "dual_reference_fails": 0}
        differ = []
        t0 = time.time()
        for trial in range(orders):
            fx = permuted_fixture(base, rng)
            try:
                T.order_case(fx, copies * ka["obj"], tally)
            except AssertionError as e:
                differ.append({"trial": trial, "what": str(e)[:300]})
                print("DIFFERENT", name, copies, trial, str(e)[:200], flush=True)
        rec = {"name": name, "copies": copies, "rows": len(base["constraints"]), "orders": orders,
               "engine_differs_from_oracle": len(differ), "oracle_endings": tally, "seconds": round(time.time() - t0, 1),
               "differences": differ}
        print(json.dumps(rec), flush=True)
        out["problems"].append(rec)
    path = os.path.join(ROOT, "gpurun_out", "blockdiag_orders.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    sys.exit(1 if any(p["engine_differs_from_oracle"] for p in out["problems"]) else 0)


if __name__ == "__main__":
    main()
