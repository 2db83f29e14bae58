"""tools/sharded_world1.py — cost of the sharded loop's plumbing on ONE GPU: the library loop with an
in-place ncclAllGather per iteration on a 1-rank communicator (exchange=rccl) and the Python loop
(exchange=torch, world 1: no collective), against the plain engine.  C3 shape by default."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ellp_amd import _engine as E, synth
from ellp_amd.dist import ShardedEngine

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 5000)
f = synth.primal_phase1_flat(20260301, m, n)
def fp():
    return E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"],
                         f["B"], f["N"], f["Nb"])
for name, make in (("engine.run", lambda: E.Engine(E.ENGINE_PRIMAL, fp(), E.default_opts(max_iter=None))),
                   ("sharded, library loop + ncclAllGather", lambda: ShardedEngine(E.ENGINE_PRIMAL, fp(), E.default_opts(max_iter=None), exchange="rccl")),
                   ("sharded, python loop (no collective at world 1)", lambda: ShardedEngine(E.ENGINE_PRIMAL, fp(), E.default_opts(max_iter=None), exchange="torch")),
                   ("column-sharded loop, two-launch kernels + ncclAllGather", lambda: ShardedEngine(E.ENGINE_PRIMAL, fp(), E.default_opts(max_iter=None), exchange="rccl", colshard=True))):
    eng = make()
    eng.run(300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, stats, msg = eng.run(3000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:58s} {3000 / dt:9.1f} pivots/s  {1e6 * dt / 3000:7.2f} us/iteration  (exchange: {getattr(eng, 'exchange_name', '-')})")
    eng.close()
