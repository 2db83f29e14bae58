import os, sys, zlib, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import ellp_amd
from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
name, copies, orders = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ka = next(p for p in known_answers()["netlib"] if p["name"] == name)
base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
want = copies * ka["obj"]
bad = []
for t in range(orders):
    fx = permuted_fixture(base, rng)
    prob = ellp_amd.Problem.from_fixture(fx)
    try:
        cls = ellp_amd.PrimalSimplexSolver if os.environ.get("SOLVER") == "primal" else ellp_amd.DualSimplexSolver
        r = cls.new(None).solve(prob)
        ok = r.kind == "optimal" and abs(r.solution.obj() / want - 1) < 1e-9
        what = (r.kind, r.iters)
    except Exception as ex:
        ok, what = False, repr(ex)[:80]
    print("order", t, ok, what, flush=True)
    if not ok:
        bad.append((t, what))
print(len(bad), "bad of", orders, bad)
import json
out = os.path.join("/root/repo" if os.path.isdir("/root/repo") else ".", "gpurun_out")
os.makedirs(out, exist_ok=True)
solver = os.environ.get("SOLVER", "dual")
json.dump({"problem": f"{name}x{copies}", "solver": solver, "orders": orders, "at_the_pinned_optimum": orders - len(bad),
           "others": [{"trial": t, "what": str(w)[:120]} for t, w in bad]},
          open(os.path.join(out, f"api_orders_{name}x{copies}_{solver}.json"), "w"))
