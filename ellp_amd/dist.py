"""Column-block sharding over several GPUs, one process per GPU.

Primal engines (the default with more than one rank): the nonbasic columns are sharded in STORAGE as
well as in pricing — each rank keeps only its block of A_N — and the per-iteration exchange is one
small "pack" per rank (its best candidates with their columns), carried by a peer-to-peer mailbox
(hipIpc-mapped memory, direct stores over xGMI), by RCCL, or — for gloo tests — by a host callback; the
loop runs inside the library (`ellp_engine_run_sharded`, ellp_amd/csrc/engine/ellp_shard.inc).

Dual engines and `colshard=False` (the round-1 scheme, kept as a cross-check):

The nonbasic positions are cut into `world` contiguous blocks.  Per simplex iteration every
rank prices its own block (k_price), then ONE all-gather of the engine's exchange buffer
(per-block maxima/argmins, Dantzig keys, reduced costs; `seg` doubles per rank) gives every
rank the complete pricing result, and the rest of the iteration (entering fold, FTRAN, ratio
test, eta update of B^-1, bookkeeping) runs replicated and deterministically on every rank, so
all ranks take the same pivots and hold the same point.  The collective is
`torch.distributed.all_gather_into_tensor` — RCCL over xGMI with backend "nccl"; with backend
"gloo" (tests) the segments are staged through host memory.
"""
import time

import numpy as np

from . import _engine as E


def shard_ranges(n_positions, cpb, world):
    """[(first, last+1)] nonbasic positions priced by each rank (mirrors ellp_engine_set_shard)."""
    nblocks = (max(n_positions, 1) + cpb - 1) // cpb
    nbs = (nblocks + world - 1) // world
    out = []
    for r in range(world):
        a = min(n_positions, r * nbs * cpb)
        b = min(n_positions, (r + 1) * nbs * cpb)
        out.append((a, b))
    return out


def all_gather_segments(full, mine, rank, world, group=None):
    """All-gather `mine` (this rank's segment) into `full` (world * seg).  Works for device
    tensors with the nccl backend and for CPU tensors (or device tensors staged through the
    host) with gloo."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return
    backend = dist.get_backend(group)
    if backend == "nccl":
        # `mine` may be the rank's own slice of `full` (in-place all-gather, supported by NCCL/RCCL
        # when sendbuff == recvbuff + rank * count)
        dist.all_gather_into_tensor(full, mine, group=group)
        return
    src = mine.detach().cpu().contiguous()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    full.copy_(torch.cat(parts).to(full.device))


def _torch_rccl_path():
    """The RCCL library this process already uses through torch (so the engine binds the same copy)."""
    import os
    import torch
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class ShardedEngine:
    """Drives one `Engine` per rank with the pricing pass sharded over the process group.

    exchange="rccl" (default with backend nccl): the per-iteration loop runs inside the library
    (`ellp_engine_run_sharded`) and the all-gather is an in-place ncclAllGather on the engine's own
    stream; torch.distributed only carries the 128-byte unique id once.  exchange="torch": the loop
    runs here, one `all_gather_into_tensor` per iteration (the only choice with backend gloo; also
    the fallback if RCCL cannot be bound).  ELLP_DIST_EXCHANGE overrides."""

    def __init__(self, kind, fp, opts=None, group=None, exchange=None, colshard=None):
        import os
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.eng = E.Engine(kind, fp, opts)
        backend = dist.get_backend(group) if self.world > 1 else None
        exchange = exchange or os.environ.get("ELLP_DIST_EXCHANGE")
        if colshard is None:
            env = os.environ.get("ELLP_DIST_COLSHARD")
            colshard = (kind == E.ENGINE_PRIMAL and self.world > 1) if env is None else env == "1"
        self.colshard = bool(colshard) and kind == E.ENGINE_PRIMAL
        self.exchange_name = None
        self.stream = None
        if self.colshard:
            self.direct = True
            self._init_colshard(backend, exchange)
            return
        exchange = exchange or ("rccl" if backend == "nccl" else "torch")
        if exchange in ("mailbox", "callback"):
            exchange = "rccl" if backend == "nccl" else "torch"
        self.exchange_name = "all-gather of the whole pricing output (" + exchange + ")"
        self.direct = False
        if exchange == "rccl":
            self.direct = self._init_direct(backend)
        if self.direct:
            self.seg = self.eng.segment_doubles(self.world)
            self.stream = None
            return
        # the exchange buffer is a torch allocation handed to the engine, so the collective
        # library works on memory it knows
        self.seg = seg = self.eng.segment_doubles(self.world)
        self.full = torch.zeros(seg * self.world, dtype=torch.float64, device="cuda")
        self.send = torch.empty(seg, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        self.eng.set_shard(self.rank, self.world, self.full.data_ptr())
        # The engine's launches, the staging copy and the collective must be ordered on ONE
        # stream.  It has to be a real (non-null) stream: handle 0 would mean "the engine's own
        # stream", which is non-blocking and unordered with torch's default stream.
        self.stream = torch.cuda.Stream()
        assert self.stream.cuda_stream != 0
        self.eng.set_stream(self.stream.cuda_stream)
        self.mine = self.full[self.rank * seg:(self.rank + 1) * seg]  # this rank's slice (a view)
        self.nccl = self.world > 1 and backend == "nccl"

    def _vote(self, ok, dev):
        """True iff every rank succeeded: the ranks must agree on the transport they use"""
        if self.world == 1:
            return bool(ok)
        flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def _init_colshard(self, backend, exchange):
        """Sharded storage of A_N + the pack exchange.  Transport: the peer-to-peer mailbox if every rank
        can map every peer's memory and a self-test of the hand-off passes (every word of 8 exchanges
        checked), else RCCL (backend nccl), else a host callback over the process group (gloo)."""
        torch, dist = self.torch, self.dist
        dev = "cuda" if backend == "nccl" else "cpu"
        self.eng.shard_columns(self.rank, self.world)
        want = exchange or ("auto" if backend == "nccl" else "callback")
        if self.world == 1:
            self.eng.set_exchange_callback(lambda buf, seg, world: 0)
            self.exchange_name = "none (one rank)"
            return
        if want in ("auto", "mailbox"):
            ok = True
            handles = None
            try:
                handles = self.eng.mailbox_export()
            except Exception:
                ok = False
            if self._vote(ok, dev):
                mine = torch.frombuffer(bytearray(handles), dtype=torch.uint8).to(dev)
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                blob = b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts)
                try:
                    self.eng.mailbox_connect(blob)
                except Exception:
                    ok = False
                if self._vote(ok, dev):
                    if self.world > 1:
                        dist.barrier(group=self.group)
                    try:
                        self.eng.mailbox_selftest(8)
                    except Exception:
                        ok = False
                    if self._vote(ok, dev):
                        self.exchange_name = "pack exchange over the peer-to-peer mailbox"
                        return
            if want == "mailbox":
                raise RuntimeError("the peer-to-peer mailbox could not be set up on every rank")
        if backend == "nccl" and want in ("auto", "rccl", "mailbox"):
            if self._init_direct(backend):
                self.exchange_name = "pack exchange by ncclAllGather"
                return
        # host callback over the process group (gloo; also the last resort with nccl)
        def gather(buf, seg, world):
            import ctypes
            import numpy as np
            arr = np.ctypeslib.as_array((ctypes.c_uint8 * (seg * world)).from_address(buf))
            t = torch.from_numpy(arr)
            mine_ = t[self.rank * seg:(self.rank + 1) * seg].clone()
            if backend == "nccl":
                g = mine_.cuda()
                parts_ = [torch.empty_like(g) for _ in range(world)]
                dist.all_gather(parts_, g, group=self.group)
                t.copy_(torch.cat(parts_).cpu())
            else:
                parts_ = [torch.empty_like(mine_) for _ in range(world)]
                dist.all_gather(parts_, mine_, group=self.group)
                t.copy_(torch.cat(parts_))
            return 0
        self.eng.set_exchange_callback(gather)
        self.exchange_name = "pack exchange staged through the host (process group " + str(backend) + ")"

    def _init_direct(self, backend):
        """Unique id from rank 0 to everyone over the process group, then ncclCommInitRank in the
        library.  Every rank reports whether it succeeded; the direct path is used only if all did
        (otherwise all fall back together — the ranks must agree on the sequence of collectives)."""
        torch, dist = self.torch, self.dist
        path = _torch_rccl_path()
        dev = "cuda" if backend == "nccl" else "cpu"
        ok = 1
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        try:  # every rank binds RCCL here (and votes below) BEFORE anyone enters the blocking init
            raw = E.comm_unique_id(path)
            if self.rank == 0:
                uid.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        except Exception:
            ok = 0
        if self.world > 1:
            dist.broadcast(uid, src=0, group=self.group)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        if self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            return False
        try:
            self.eng.comm_init(bytes(uid.cpu().numpy().tobytes()), self.rank, self.world, path)
            ok = 1
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        if self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def exchange(self):
        """Must be called with self.stream current (run() does)."""
        if self.world == 1:
            return
        if self.nccl:  # in place: sendbuff == recvbuff + rank * count
            self.dist.all_gather_into_tensor(self.full, self.mine, group=self.group)
        else:
            self.send.copy_(self.mine)
            all_gather_segments(self.full, self.send, self.rank, self.world, self.group)

    def run(self, max_iters, poll_interval=16):
        """Up to `max_iters` further iterations; returns (status, Stats, message) like Engine.run."""
        if self.direct:
            return self.eng.run_sharded(max_iters)
        done = 0
        with self.torch.cuda.stream(self.stream):
            status, stats, msg = self.eng.poll()
            it0 = stats.iters
            while status == E.MAXITER and done < max_iters:
                batch = min(poll_interval, max_iters - done)
                # one host call + one collective per iteration: step(2) = rest of iteration k and
                # the pricing of iteration k+1 (the last one prices ahead; a later step(0) simply
                # prices again from the same state)
                self.eng.step(0)
                for k in range(batch):
                    self.exchange()
                    self.eng.step(2 if k + 1 < batch else 1)
                status, stats, msg = self.eng.poll()
                # iterations that really ran: a maintenance request voids the rest of its batch (the
                # device state is replicated, so every rank computes the same count)
                done = stats.iters - it0
        return status, stats, msg

    def read_point(self):
        """the point of the replicated state; raises EllpHipError on any error code (Engine.read_point)"""
        return self.eng.read_point()

    @property
    def closing_status(self):
        """what completing an open iteration produced at the last read_point (OPTIMAL: nothing; UNBOUNDED: the solve ended)"""
        return getattr(self.eng, "closing_status", None)

    def close(self):
        if self.stream is not None:
            self.stream.synchronize()
            self.eng.set_stream(None)
        self.eng.close()
