"""tools/bl_stamps.py [m] — where k_bl_factor (the sub-panel kernel of the blocked rebuild) spends its time: needs a
-DELLP_BL_STAMPS build of the engine (ELLP_HIP_LIB=ellp_amd/libellp_hip_dbg.so); prints microseconds per launch."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from ellp_amd import _engine as E
from test_gpu_rebuild import dense_basis_problem
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fp, B = dense_basis_problem(m, 100 + m)
eng = E.Engine(E.ENGINE_PRIMAL, fp, E.default_opts(max_iter=None, pipeline=1))
out = (C.c_ulonglong * 8)()
E.lib().ellp_debug_bl_stamps(out)
n = max(1, out[5])
names = ["load tile", "pivot search (scan, DPP, LDS, barrier)", "fold + publish pivot row (LDS, barrier)", "update", "store V_s"]
print(f"m={m}: {out[5]} launches of k_bl_factor; us per launch:", {names[k]: round(out[k] / 100.0 / n, 2) for k in range(5)})
eng.close()
