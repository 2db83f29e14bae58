"""tools/small_lp_times.py — the launch-bound regime: the netlib fixtures solved through
the user API (parse_mps -> Primal/DualSimplexSolver.new(None).solve) with the explicit-inverse engine (three
launches per pivot, pipeline=1) and with the persistent single-workgroup kernel (the default at m <= 128).
Prints one JSON object; the first solve of each kind is a warm-up (module load) and is not timed."""
import glob, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd import DualSimplexSolver, PrimalSimplexSolver, parse_mps

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "netlib")
out = {"what": "wall time of solve() through the user API, best of 3, ms", "problems": {}}
files = sorted(glob.glob(os.path.join(root, "*.mps")) + glob.glob(os.path.join(root, "*.MPS")))
for path in files:
    text = open(path).read()
    name = os.path.splitext(os.path.basename(path))[0]
    row = {}
    for sname, S in (("primal", PrimalSimplexSolver), ("dual", DualSimplexSolver)):
        for label, kw in (("three_launch", {"pipeline": 1}), ("persistent", {})):
            best, res = None, None
            for rep in range(4):
                p = parse_mps(text)
                t0 = time.perf_counter()
                try:
                    res = S.new(None, **kw).solve(p)
                    kind, iters = res.kind, sum(res.iters)
                except Exception as ex:  # an Err / panic outcome is an outcome (the reference has them too)
                    kind, iters = "error: " + str(ex)[:60], 0
                dt = (time.perf_counter() - t0) * 1e3
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            row[f"{sname}_{label}"] = {"ms": round(best, 2), "status": kind, "iterations": iters,
                                       "us_per_iteration": round(best * 1e3 / iters, 2) if iters else None}
    out["problems"][name] = row
print(json.dumps(out, indent=1))
