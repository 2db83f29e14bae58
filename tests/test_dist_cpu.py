"""world_size-2 gloo tests of the N>1 path's host logic (runs on CPU): the shard ranges tile
the nonbasic positions exactly as the engine's block split does, and the exchange helper
all-gathers per-rank segments into the same full buffer on every rank."""
import os
import socket

import numpy as np
import pytest

from helpers import collect_results
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
    results = collect_results(q, procs, world, 120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered in results:
        assert all(ok for ok, _ in gathered), gathered
        assert len({b for _, b in gathered}) == 1


# ---------------------------------------------------------------------------------------------------
# The selection every rank runs on the gathered packs (ellp_amd/csrc/engine/ellp_shard.inc,
# shard_select_compact: the SAME source the kernel k_sh_select compiles, built for the host and exported
# as ellp_shard_select_compact), fed with packs that two gloo ranks build from their own halves of a
# pricing result and all-gather.  Whenever it says "conclusive" its entering position must be the one
# the reference's sequential fold (primal_simplex_solver.rs:271-287, restated below) finds over ALL
# elements; "not conclusive" is only allowed when a key really lies in the gap or a pack overflowed.
EPS = 1e-10
KC = 2


def reference_fold(keys, nidx):
    """Iterator::max_by with the reference's comparator: |a - b| >= EPS ? by key : by N.index (larger wins)"""
    have, racc, iacc, q = False, 0.0, 0, -1
    for j, (k, i) in enumerate(zip(keys, nidx)):
        if k == -np.inf:
            continue
        if have:
            acc_greater = (racc > k) if abs(racc - k) >= EPS else (iacc > i)
            if acc_greater:
                continue
        have, racc, iacc, q = True, k, i, j
    return q


def build_pack(keys, nidx, pos0, ld, pd):
    """what k_pack writes: local maximum, the candidates within 6 EPS of it (at most KC, in position
    order) with a column each, the overflow flag"""
    pack = np.zeros(pd)
    M = keys.max() if keys.size else -np.inf
    pack[0] = M
    cand = [j for j in range(len(keys)) if M > -np.inf and M - keys[j] < 6 * EPS]
    pack[1] = min(len(cand), KC)
    pack[2] = 1.0 if len(cand) > KC else 0.0
    for c, j in enumerate(cand[:KC]):
        base = 4 + c * (4 + ld)
        pack[base:base + 4] = [keys[j], nidx[j], pos0 + j, -keys[j]]
        pack[base + 4:base + 4 + ld] = float(pos0 + j)  # the "column": recognisable
    return pack


def _select_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ellp_amd import _engine as E
        ld = 16
        pd = E.shard_pack_doubles(ld)
        rng = np.random.default_rng(99)  # the same stream on every rank: each takes its own half
        n_conclusive = n_full = 0
        bad = []
        for case in range(600):
            n = int(rng.integers(2, 40))
            style = case % 4
            if style == 0:      # distinct keys
                keys = rng.random(n) * 10
            elif style == 1:    # exact ties (integer data)
                keys = rng.integers(0, 4, size=n).astype(float)
            elif style == 2:    # near-ties around the maximum, some inside the gap
                keys = 5.0 - rng.integers(0, 9, size=n) * 0.9e-10
            else:               # mostly ineligible
                keys = np.where(rng.random(n) < 0.7, -np.inf, rng.integers(0, 3, size=n).astype(float))
            nidx = rng.permutation(n * 3)[:n].astype(float)
            half = n // 2
            lo, hi = (0, half) if rank == 0 else (half, n)
            mine = torch.from_numpy(build_pack(keys[lo:hi], nidx[lo:hi], lo, ld, pd))
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            packs = torch.cat(parts).numpy()
            verdict, q, sr, sc = E.shard_select_compact(packs, world, ld, EPS)
            want = reference_fold(keys, nidx)
            M = keys.max()
            if verdict == 0:
                n_conclusive += 1
                if q != want:
                    bad.append((case, "q", q, want))
                if q >= 0:
                    col = packs[sr * pd + 4 + sc * (4 + ld) + 4: sr * pd + 4 + sc * (4 + ld) + 4 + ld]
                    if not np.all(col == float(q)):
                        bad.append((case, "column", q, sr, sc))
            else:
                n_full += 1
                in_gap = np.any((M - keys >= 4 * EPS) & (M - keys < 6 * EPS))
                overflow = any(np.sum(M - keys[a:b] < 6 * EPS) > KC for a, b in ((0, half), (half, n)))
                if not (in_gap or overflow):
                    bad.append((case, "needless full exchange"))
        gathered = [None] * world
        dist.all_gather_object(gathered, (n_conclusive, n_full, bad[:5]))
        out.put((rank, gathered))
    finally:
        dist.destroy_process_group()


def test_pack_selection_on_gathered_packs_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_select_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = collect_results(q, procs, world, 180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered in results:
        assert len({(a, b) for a, b, _ in gathered}) == 1          # both ranks decided alike, case by case
        n_conclusive, n_full, bad = gathered[0]
        assert not bad, bad
        assert n_conclusive > 250 and n_full > 50, (n_conclusive, n_full)
