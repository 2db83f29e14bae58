"""tools/rebuild_prof.py [m] — one blocked rebuild of a dense basis (engine creation) and nothing else, for rocprofv3
--kernel-trace --stats: which kernels of ellp_rebuild.inc the time goes to."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from ellp_amd import _engine as E
from test_gpu_rebuild import dense_basis_problem
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fp, B = dense_basis_problem(m, 100 + m)
for rep in range(3):
    eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=1))
    print(rep, m, "setup_s", eng.counters()["t_setup_s"], flush=True)
    eng.close()
