import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ellp_amd import _engine as E, synth
m, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
total = int(sys.argv[4])
f = synth.primal_phase1_flat(seed, m, n)
def mk(flags):
    fp = E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"])
    return fp, E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, flags=flags))
fa, ea = mk(0)
fb, eb = mk(1)
done = 0
step = 50
while done < total:
    ea.run(step); eb.run(step)
    ea.read_point(); eb.read_point()
    done += step
    if not np.array_equal(fa.B, fb.B) or not np.array_equal(fa.N, fb.N):
        print("differ within iterations", done - step, done, "B diff at", np.flatnonzero(fa.B != fb.B)[:5], "N diff at", np.flatnonzero(fa.N != fb.N)[:5])
        r = ea.tap(E.TAP_R, fa.nN); rb = eb.tap(E.TAP_R, fb.nN)
        break
else:
    print("same basis after", done, "x equal:", fa.x.tobytes() == fb.x.tobytes())
