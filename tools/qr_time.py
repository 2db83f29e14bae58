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
    t0 = time.perf_counter()
    piv, rd = E.qr_transposed(S)
    dt = time.perf_counter() - t0
    flop = 2.0 * S.shape[1] * m * m
    print(f"m={m} nv={S.shape[1]}: device QR {dt:.3f} s  (rank {int((rd > 1e-10).sum())}/{m}, ~{flop / 1e9:.0f} GFLOP of work)")
