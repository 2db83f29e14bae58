import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from ellp_amd import _engine as E
from test_gpu_rebuild import dense_basis_problem
seq = [int(v) for v in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for rep in range(reps):
    for m in seq:
        fp, B = dense_basis_problem(m, 100 + m)
        try:
            eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=1))
            print(rep, m, "residual", eng.inverse_residual(), flush=True)
            eng.close()
        except Exception as e:
            print(rep, m, "ERR", e, flush=True)
