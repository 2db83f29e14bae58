"""tools/lu_time.py — time the device LU of A^T (ellp_hip_lu_transposed, the basis of dual phase 1) at the standard-form
shapes of configs 3 and 5 (m x (n + m)); prints one JSON line per shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ellp_amd import _engine as E, synth

shapes = [(500, 1250), (2000, 5000)] + ([(4000, 40000)] if "--c5" in sys.argv else [])
for m, n in shapes:
    A, b, c = synth.dense_lp(20260301, m, n)
    S = np.asfortranarray(np.hstack([A, np.eye(m)[:, ::-1]]))  # [A | slacks placed right to left]
    E.lu_transposed(S[:8, :16])  # warm
    t0 = time.perf_counter()
    piv, ud = E.lu_transposed(S)
    dt = time.perf_counter() - t0
    nv = S.shape[1]
    elems = sum((nv - i) * (m - i) for i in range(m))
    print(json.dumps({"m": m, "nv": nv, "device_lu_s": round(dt, 3), "algorithmic_GB": round(16 * elems / 1e9, 1),
                      "GBps": round(16 * elems / dt / 1e9, 1), "min_abs_udiag": float(np.abs(ud).min()),
                      "swaps": int((piv != np.arange(m)).sum())}), flush=True)
