"""tools/stamps.py — dev aid: run the C3 phase-1 iteration on a -DELLP_DBG_STAMPS build of the engine
(ELLP_HIP_LIB=ellp_amd/libellp_hip_dbg.so) and let it print the in-kernel timestamps of the last
iteration at destroy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd import _engine as E, synth

m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
flat = synth.primal_phase1_flat(20260301, m, n)
fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"],
                   flat["lb"], flat["ub"], flat["x"], flat["B"], flat["N"], flat["Nb"], None, None)
eng = E.Engine(E.ENGINE_PRIMAL, fp)
eng.run(1037)
eng.close()
