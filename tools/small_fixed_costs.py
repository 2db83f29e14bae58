"""tools/small_fixed_costs.py [name] — where the wall time of a small solve goes besides the iterations: the host
mirror's set-up (standard form with the rank check, phase-1 arrays), engine creation, the run, the read-back,
destruction; best of 5 after a warm-up.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd import PrimalSimplexSolver, parse_mps, _engine as E

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "afiro"
text = open(os.path.join(root, "tests", "golden", "netlib", name + ".mps")).read()
best = {}
def note(k, dt):
    best[k] = min(best.get(k, 1e9), dt * 1e3)
for rep in range(6):
    t = time.perf_counter(); p = parse_mps(text); t1 = time.perf_counter()
    f = p._debug_phase1("primal"); t2 = time.perf_counter()
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"])
    t3 = time.perf_counter(); eng = E.Engine(E.ENGINE_PRIMAL, fp); t4 = time.perf_counter()
    st, stats, msg = eng.run(1 << 40); t5 = time.perf_counter()
    eng.read_point(); t6 = time.perf_counter()
    eng.close(); t7 = time.perf_counter()
    p2 = parse_mps(text); t8 = time.perf_counter(); r = PrimalSimplexSolver.new(None).solve(p2); t9 = time.perf_counter()
    if rep:
        note("parse_mps_ms", t1 - t); note("standard_form_and_phase1_arrays_ms", t2 - t1); note("engine_create_ms", t4 - t3)
        note("run_phase1_ms", t5 - t4); note("read_point_ms", t6 - t5); note("destroy_ms", t7 - t6); note("whole_solve_ms", t9 - t8)
best = {k: round(v, 3) for k, v in best.items()}
best.update(problem=name, phase1_iterations=int(stats.iters), solve_iterations=list(r.iters))
print(json.dumps(best))
