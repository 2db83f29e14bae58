"""The debug build of the engine (-DELLP_DEBUG_BOUNDS, SURVEY.md §5: "a debug kernel mode with bounds asserts"): every
index a decision commits is checked by the thread that commits it.  The build is loaded in a child process through
ELLP_HIP_LIB; the same solves as the product build, no assertion fires, same results."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "ellp_amd", "libellp_hip_dbg.so")

CODE = r'''
import json, sys
sys.path.insert(0, %r)
from ellp_amd import _engine as E, synth
out = {}
for name, kind, f, pl in (("primal3", E.ENGINE_PRIMAL, synth.primal_phase1_flat(20260301, 300, 700), 1),
                          ("primal2", E.ENGINE_PRIMAL, synth.primal_phase1_flat(20260301, 420, 1000), 2),
                          ("dual3", E.ENGINE_DUAL, synth.dual_start_flat(20260301, 260, 600), 1),
                          ("dual2", E.ENGINE_DUAL, synth.dual_start_flat(20260301, 600, 1500), 2),
                          ("small", E.ENGINE_PRIMAL, synth.primal_phase1_flat(20260301, 40, 100), 0)):
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"],
                       f.get("y"), f.get("d"))
    eng = E.Engine(kind, fp, E.default_opts(max_iter=None, pipeline=pl))
    st, stats, msg = eng.run(1 << 40)
    eng.read_point()
    out[name] = [int(st), int(stats.iters), [int(v) for v in fp.B[:8]], msg]
    eng.close()
print("RESULT " + json.dumps(out))
''' % ROOT


def _run(lib):
    env = dict(os.environ)
    if lib:
        env["ELLP_HIP_LIB"] = lib
    else:
        env.pop("ELLP_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_debug_build_runs_the_same_solves_without_an_assertion():
    if not os.path.exists(DBG):
        pytest.skip("libellp_hip_dbg.so not built (python -m ellp_amd.build --debug-bounds)")
    dbg = _run(DBG)
    ref = _run(None)
    for name, (st, iters, B, msg) in dbg.items():
        assert st == 0, (name, st, msg)            # Optimal; an assertion would have been ELLP_ERR_PANIC (-3)
        assert [st, iters, B] == ref[name][:3], (name, dbg[name], ref[name])
