"""tools/qr_time.py — time the device QR of A^T (ellp_hip_qr_transposed) at the standard-form shapes of
configs 3 and 5 (m x (n + m)), and the host loop at a size it finishes quickly, for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ellp_amd import _engine as E, synth

for m, n in ((500, 1250), (2000, 5000), (4000, 40000)):
    A, b, c = synth.dense_lp(20260301, m, n)
    S = np.asfortranarray(np.hstack([A, np.eye(m)[:, ::-1]]))  # [A | slacks placed right to left]
    E.qr_transposed(S[:8, :16])  # warm
    flop = 2.0 * S.shape[1] * m * m
    out = {}
    for mode in (["0", "1"] if "--exact" in sys.argv else ["0"]):
        os.environ["ELLP_QR_EXACT"] = mode
        t0 = time.perf_counter()
        piv, rd = E.qr_transposed(S)
        dt = time.perf_counter() - t0
        out[mode] = (piv, rd)
        print(f"m={m} nv={S.shape[1]}: device QR ({'exact' if mode == '1' else 'fast'}) {dt:.3f} s  (rank {int((rd > 1e-10).sum())}/{m}, ~{flop / 1e9:.0f} GFLOP of work)", flush=True)
    if len(out) == 2:
        print(f"   same pivots: {bool((out['0'][0] == out['1'][0]).all())}, max rel diff of |R_ii|: {float(np.max(np.abs(out['0'][1] - out['1'][1]) / out['1'][1])):.2e}", flush=True)
