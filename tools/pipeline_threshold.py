"""tools/pipeline_threshold.py — at which size the two-launch forms (primal: ellp_lagged.inc, dual: ellp_dualfu.inc) start
to pay: microseconds per iteration of pipeline 1 (three launches) and 2 over a fixed window, sizes 160 … 1536."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd import _engine as E, synth

for m, n in ((160, 400), (256, 640), (384, 960), (512, 1280), (768, 1920), (1024, 2560), (1536, 3840)):
    row = {"m": m, "n": n}
    for kind, name in ((E.ENGINE_PRIMAL, "primal"), (E.ENGINE_DUAL, "dual")):
        f = synth.primal_phase1_flat(20260301, m, n) if kind == E.ENGINE_PRIMAL else synth.dual_start_flat(20260301, m, n)
        for pl in (1, 2):
            fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"],
                               f["Nb"], f.get("y"), f.get("d"))
            eng = E.Engine(kind, fp, E.default_opts(max_iter=None, pipeline=pl))
            eng.run(200)
            t0 = time.perf_counter()
            st, stats, msg = eng.run(1500)
            dt = time.perf_counter() - t0
            it = stats.iters - 200
            row[f"{name}_pipeline{pl}_us"] = round(dt / max(1, it) * 1e6, 2) if it > 0 else None
            eng.close()
    print(json.dumps(row), flush=True)
