import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ellp_amd import _engine as E
from oracle import ellp_oracle as eo
eo.set_primal_rule(1)
p1, err = eo.primal_phase1(eo.synth_problem(20260301, 100, 250))
v = p1.view()
A = np.asarray(v.A).reshape((v.n, v.m)).T
for k in list(range(28, 125, 4)):
    ov = v.copy()
    eo.primal_solve_with_initial(ov, k)
    fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN])
    st, stats, msg = E.primal_solve_with_initial(fp, E.default_opts(max_iter=k, flags=4))
    same = np.array_equal(fp.B, ov.B)
    if not same:
        print("first difference at iteration", k, "oracle B diff", np.flatnonzero(fp.B != ov.B), "oracle entered", set(ov.B) - set(fp.B), "engine entered", set(fp.B) - set(ov.B))
        # exact SE keys for the basis before iteration k
        ov2 = v.copy(); eo.primal_solve_with_initial(ov2, k - 1)
        B, N, Nb = ov2.B, ov2.N[:ov2.nN], ov2.Nb[:ov2.nN]
        Binv = np.linalg.inv(A[:, B]); u = Binv.T @ v.c[B]; r = v.c[N] - A[:, N].T @ u
        gam = np.array([1 + np.sum((Binv @ A[:, j]) ** 2) for j in N])
        elig = ((r > 0) & (Nb == 1)) | ((r < 0) & (Nb == 0)) | (Nb == 2)
        key = np.where(elig & (np.abs(r) >= 1e-10), r * r / gam, -np.inf)
        order = np.argsort(-key)[:4]
        print("  exact keys top4:", [(int(N[j]), float(key[j])) for j in order])
        break
else:
    print("same up to 124 iterations")
