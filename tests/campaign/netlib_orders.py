#!/usr/bin/env python3
"""tests/campaign/netlib_orders.py [orders_per_problem] [seed] [pipeline] — the campaign behind
tests/test_gpu_random.py::test_netlib_in_random_orders, without stopping at the first failure: netlib
AFIRO / ADLITTLE / BLEND in random variable and constraint orders through the dual and primal loops,
oracle (CPU) against engine (GPU).  Every order on which the two end differently (status or objective)
is written to gpurun_out/netlib_orders_failures.json WITH its permutations, so that it can be
committed to tests/golden/netlib_orders.json as a regression case."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDEN, known_answers, read_mps  # noqa: E402
from oracle import ellp_oracle as eo  # noqa: E402
import test_gpu_random as T  # noqa: E402


def permuted(fx, rng):
    n = len(fx["vars"])
    perm = rng.permutation(n)
    inv = np.empty(n, dtype=int)
    inv[perm] = np.arange(n)
    rperm = rng.permutation(len(fx["constraints"]))
    rows = [fx["constraints"][i] for i in rperm]
    out = {"vars": [fx["vars"][j] for j in perm],
           "constraints": [[[[int(inv[j]), a] for j, a in coeffs], op, rhs] for coeffs, op, rhs in rows]}
    return out, [int(v) for v in perm], [int(v) for v in rperm]


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failures, total = [], 0
    n_ref_errors = n_paths = 0
    if len(sys.argv) > 3:
        T.SEAM_PIPELINE[0] = int(sys.argv[3])
    for name in ("afiro", "adlittle", "blend"):
        ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
        base = read_mps(os.path.join(GOLDEN, ka["file"]))
        rng = np.random.default_rng(zlib.crc32(name.encode()) + seed)
        for trial in range(count):
            fx, perm, rperm = permuted(base, rng)
            total += 1
            bad, ref_errors = [], []
            try:
                T.netlib_order_case(base, ka, fx, trial, bad, ref_errors)
                n_ref_errors += len(ref_errors)
                hard = [b for b in bad if not (b[1] == "path" and b[4] < 1e-8 * (1 + abs(ka["obj"])))]
                n_paths += len(bad) - len(hard)
                if hard:
                    raise AssertionError(str(hard[:2]))
            except AssertionError as e:
                failures.append({"name": name, "tag": f"s{seed}t{trial}", "var_perm": perm, "row_perm": rperm,
                                 "what": str(e)[:300]})
                print("FAIL", name, seed, trial, str(e)[:200], flush=True)
        print(name, "done", flush=True)
    out = os.path.join(ROOT, "gpurun_out", "netlib_orders_failures.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"orders_per_problem": count, "seed": seed, "total": total, "failures": failures}, f)
    print(json.dumps({"total": total, "failed": len(failures), "orders_the_reference_itself_rejects": n_ref_errors,
                      "other_tie_paths": n_paths, "pipeline": T.SEAM_PIPELINE[0]}))


if __name__ == "__main__":
    main()
