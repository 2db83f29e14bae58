import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ellp_amd import _engine as E
from oracle import ellp_oracle as eo
from helpers import GOLDEN, read_mps
for name in ("adlittle", "afiro", "blend"):
    fx = read_mps(os.path.join(GOLDEN, "netlib", name + ".mps"))
    p1, err = eo.dual_phase1(eo.Problem.from_fixture(fx))
    v = p1.view()
    for period in (0, 16, 32, 48, 64, 100, 128, 1 << 30):
        fp = E.FlatProblem(v.m, v.n, v.n_c, v.A, v.c, v.b, v.kind, v.lb, v.ub, v.x, v.B, v.N[:v.nN], v.Nb[:v.nN], v.y, v.d)
        st, stats, msg = E.dual_solve_with_initial(fp, E.default_opts(max_iter=1000, refactor_period=period))
        d = fp.d
        dobj = float(sum((v.lb[i] if d[i] > 0 else v.ub[i]) * d[i] for i in range(v.n_c) if v.kind[i] == 3))
        print(f"{name} period={period:>10} status={st} iters={stats.iters} maint={stats.refactors} dual_obj={dobj:.3e} {msg}")
