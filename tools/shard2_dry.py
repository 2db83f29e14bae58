"""tools/shard2_dry.py RANK WORLD PORT [m n steps] — one rank of a column-sharded dry run with all ranks on ONE GPU (storage
and pricing of A_N sharded, packs exchanged through the peer-to-peer mailbox): every rank its own process, started by
tools/shard2_trace.sh so that ONE of them can sit under `rocprofv3 --kernel-trace --stats` — the launches per sharded iteration
are counted from that trace (rank 0 runs nothing but the sharded loop).  Prints the rank's iteration rate and, on the LAST rank,
whether the pivots are the unsharded engine's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m, n, steps = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (2000, 5000, 1000)
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from ellp_amd import _engine as E, synth
from ellp_amd.dist import ShardedEngine
f = synth.primal_phase1_flat(20260301, m, n)
def fp():
    return E.FlatProblem(f["m"], f["n"], f["n_c"], f["A"], f["c"], f["b"], f["kind"], f["lb"], f["ub"], f["x"], f["B"], f["N"], f["Nb"])
opts = E.default_opts(max_iter=None, device=0, pipeline=int(os.environ.get("SHARD_PIPELINE", "0")))  # 0: the engine's default (two launches from m = 384), 1: three
p = fp()
sh = ShardedEngine(E.ENGINE_PRIMAL, p, opts, colshard=True, exchange="mailbox")
sh.run(100)
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
st, stats, msg = sh.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sh.read_point()
info = sh.eng.shard_info()
sh.close()
print(f"rank {rank}/{world}: {steps} sharded iterations in {dt:.3f} s = {1e6 * dt / steps:.1f} us each ({world} processes share one GPU); "
      f"transport {info.get('transport')}, full exchanges {info.get('full_exchanges')}, column requests {info.get('column_requests')}", flush=True)
if rank == world - 1:
    q = fp()
    ref = E.Engine(E.ENGINE_PRIMAL, q, opts)
    ref.run(100 + steps)
    ref.read_point()
    ref.close()
    print("same basis as the unsharded engine:", bool(np.array_equal(p.B, q.B)), flush=True)
dist.barrier()
dist.destroy_process_group()
