"""world_size-2 gloo tests of the N>1 path's host logic (runs on CPU): the shard ranges tile
the nonbasic positions exactly as the engine's block split does, and the exchange helper
all-gathers per-rank segments into the same full buffer on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ellp_amd.dist import all_gather_segments, shard_ranges


def test_shard_ranges_tile_positions():
    for n_pos, cpb, world in [(7000, 7, 1), (7000, 7, 2), (7000, 7, 8), (44000, 43, 8), (51, 1, 4), (3, 1, 8)]:
        rs = shard_ranges(n_pos, cpb, world)
        assert len(rs) == world
        assert rs[0][0] == 0 and rs[-1][1] == n_pos
        for (a0, b0), (a1, b1) in zip(rs, rs[1:]):
            assert b0 == a1 and a0 <= b0
        sizes = [b - a for a, b in rs]
        assert max(sizes) - min(s for s in sizes if s > 0 or True) <= max(sizes)  # contiguous, ordered
        # every non-empty shard is a whole number of pricing blocks except possibly the last
        nbs_cpb = ((n_pos + cpb - 1) // cpb + world - 1) // world * cpb
        for a, b in rs[:-1]:
            assert (b - a) in (0, nbs_cpb) or b == n_pos


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, seg, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1234)
        truth = rng.standard_normal(seg * world)  # same on every rank
        full = torch.zeros(seg * world, dtype=torch.float64)
        mine = torch.from_numpy(truth[rank * seg:(rank + 1) * seg].copy())
        full[rank * seg:(rank + 1) * seg] = mine
        all_gather_segments(full, mine, rank, world)
        ok = bool(np.array_equal(full.numpy(), truth))
        # a (key, position) argmax merged from the gathered keys is identical on every rank
        best = int(torch.argmax(full).item())
        gathered = [None] * world
        dist.all_gather_object(gathered, (ok, best))
        out.put((rank, gathered))
    finally:
        dist.destroy_process_group()


def test_exchange_gloo_world2():
    world, seg = 2, 1031
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seg, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered in results:
        assert all(ok for ok, _ in gathered), gathered
        assert len({b for _, b in gathered}) == 1
