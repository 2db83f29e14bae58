"""tools/partial_time.py [m n] — what partial pricing (ellp_opts.partial_segments, an opt-in extension) buys: primal
phase 1 of the synthetic dense LP run to optimality with P = 1 (every column every iteration, the reference's rule;
default pipeline and the three-launch pipeline partial pricing runs on) and P = 4, 8, 16, 32.  One JSON line each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd import _engine as E, synth

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 6000)
flat = synth.primal_phase1_flat(20260301, m, n)
for P, pipeline in ((1, 0), (1, 1), (4, 1), (8, 1), (16, 1), (32, 1)):
    fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"], flat["lb"], flat["ub"],
                       flat["x"], flat["B"], flat["N"], flat["Nb"])
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, partial_segments=P, pipeline=pipeline))
    t0 = time.perf_counter()
    st, stats, msg = eng.run(1 << 40)
    dt = time.perf_counter() - t0
    eng.read_point()
    print(json.dumps({"m": m, "n": n, "segments": P, "pipeline": pipeline or "default", "status": int(st), "iterations": int(stats.iters),
                      "seconds": round(dt, 3), "us_per_iteration": round(dt / max(1, stats.iters) * 1e6, 2),
                      "phase1_objective": fp.obj()}), flush=True)
    eng.close()
