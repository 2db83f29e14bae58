import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ellp_amd import _engine as E
from oracle import ellp_oracle as eo
p1, err = eo.primal_phase1(eo.synth_problem(20260301, 100, 250))
v = p1.view()
A = np.asarray(v.A).reshape((v.n, v.m)).T
fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN])
eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, flags=4))
def exact(B, N):
    Binv = np.linalg.inv(A[:, B])
    return np.array([1 + np.sum((Binv @ A[:, j]) ** 2) for j in N])
first = int(sys.argv[1]) if len(sys.argv) > 1 else 56
eng.run(first)
eng.read_point()
prev = (fp.B.copy(), fp.N.copy())
for it in range(first, first + 16):
    st, stats, msg = eng.run(1)
    eng.read_point()
    g = eng.tap(7, fp.nN)
    ex = exact(*prev)
    rel = np.abs(g - ex) / ex
    c = eng.counters()
    print("iteration", it + 1, "max rel err", float(rel.max()), "at", int(rel.argmax()), "maintenance so far", c.get("refreshes"), c.get("refactors"), flush=True)
    prev = (fp.B.copy(), fp.N.copy())
