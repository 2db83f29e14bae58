#!/usr/bin/env python3
"""bench.py — simplex pivots/sec + achieved HBM GB/s of the HIP pivot engine.

    python bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the revised-simplex loop (one pass of the hot path:
BTRAN/pricing/entering/FTRAN/ratio/eta-update) on the synthetic dense LP of BASELINE.json's
config 3 (m=2000, n=5000, primal; SURVEY.md §8d generator), started from the reference's own
phase-1 starting basis, tableau already resident in HBM.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured-achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=20260301)
    ap.add_argument("--solver", choices=["primal", "dual"], default="primal")
    ap.add_argument("--cpu-pivots", type=int, default=-1, help="oracle pivots to time (-1 auto, 0 off)")
    ap.add_argument("--same-alg-pivots", type=int, default=2000,
                    help="pivots of the explicit-inverse OpenMP CPU loop to time (primal, N=1)")
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--refactor-period", type=int, default=0)
    ap.add_argument("--btran-mode", type=int, default=0)
    ap.add_argument("--poll", type=int, default=0)
    ap.add_argument("--check", action="store_true", help="compare the first pivots with the oracle")
    return ap.parse_args()


def cpu_baseline(flat, pivots, solver="primal"):
    """The oracle's loop (LU refactorisation every iteration, single thread — the reference's
    algorithm) timed on this host over a bounded window from the same starting basis."""
    from oracle import ellp_oracle as eo

    class V:
        pass
    v = V()
    for k in ("m", "n", "n_c", "A", "c", "b", "kind", "lb", "ub"):
        setattr(v, k, flat[k])
    v.x = flat["x"].copy()
    v.B = flat["B"].copy()
    v.N = flat["N"].copy()
    v.Nb = flat["Nb"].copy()
    v.nB, v.nN = len(v.B), len(v.N)
    if solver == "dual":
        v.y, v.d = flat["y"].copy(), flat["d"].copy()
    eo.set_dense_lu(True)  # pay nalgebra's full (2/3) m^3 per iteration, as the reference does
    t0 = time.perf_counter()
    if solver == "dual":
        st, iters, _ = eo.dual_solve_with_initial(v, pivots)
    else:
        st, iters, _ = eo.primal_solve_with_initial(v, pivots)
    dt = time.perf_counter() - t0
    eo.set_dense_lu(False)
    return iters / dt, iters, dt, v


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    # test hooks (dry runs on a 1-GPU box): ELLP_BENCH_BACKEND=gloo, ELLP_BENCH_DEVICE=0
    backend = os.environ.get("ELLP_BENCH_BACKEND", "nccl")
    if "ELLP_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ELLP_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from ellp_amd import _engine as E
    from ellp_amd import synth

    m, n = args.m, args.n
    dual = args.solver == "dual"
    flat = synth.dual_start_flat(args.seed, m, n) if dual else synth.primal_phase1_flat(args.seed, m, n)
    fp = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"],
                       flat["lb"], flat["ub"], flat["x"], flat["B"], flat["N"], flat["Nb"],
                       flat.get("y"), flat.get("d"))
    kind = E.ENGINE_DUAL if dual else E.ENGINE_PRIMAL
    nN = fp.nN
    ld = (m + 15) // 16 * 16

    def make_engine(profile):
        opts = E.default_opts(max_iter=None, device=local_rank, refactor_period=args.refactor_period,
                              btran_mode=args.btran_mode, poll_interval=args.poll, profile=profile)
        if world > 1:
            # column-block pricing sharded over the ranks, one RCCL all-gather per iteration
            from ellp_amd.dist import ShardedEngine
            return ShardedEngine(kind, fp, opts)
        return E.Engine(kind, fp, opts)

    # ---- timed region: tableau resident, W warm-up steps, then exactly K steps
    eng = make_engine(0)
    st, stats, msg = eng.run(args.warmup)
    assert st in (E.MAXITER,), f"warm-up ended the solve: {E.STATUS_NAME.get(st)} {msg}"
    it0 = stats.iters
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    st, stats, msg = eng.run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps_done = stats.iters - it0
    assert st == E.MAXITER and steps_done == args.steps, (E.STATUS_NAME.get(st), steps_done, msg)
    refactors = stats.refactors
    resid = eng.inverse_residual() if world == 1 else None
    sharded_check = None
    if world > 1:
        # outside the timed region: the sharded run must have taken the pivots of a single-GPU run
        # (rank 0 replays them unsharded) and every rank must hold the same basis
        import zlib
        eng.read_point()
        crc = zlib.crc32(fp.B.tobytes() + fp.N.tobytes() + fp.Nb.tobytes())
        tcrc = torch.tensor([crc], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        allcrc = [torch.zeros_like(tcrc) for _ in range(world)]
        dist.all_gather(allcrc, tcrc)
        ranks_agree = all(int(t.item()) == crc for t in allcrc)
        same_as_single = None
        if rank == 0:
            fp1 = E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"],
                                flat["lb"], flat["ub"], flat["x"], flat["B"], flat["N"], flat["Nb"], flat.get("y"),
                                flat.get("d"))
            ref = E.Engine(kind, fp1, E.default_opts(max_iter=None, device=local_rank,
                                                    refactor_period=args.refactor_period, btran_mode=args.btran_mode))
            ref.run(args.warmup)
            ref.run(args.steps)
            ref.read_point()
            ref.close()
            same_as_single = bool(np.array_equal(fp1.B, fp.B) and np.array_equal(fp1.N, fp.N) and
                                  np.array_equal(fp1.Nb, fp.Nb))
        sharded_check = {"all_ranks_hold_the_same_basis": bool(ranks_agree), "same_pivots_as_one_gpu": same_as_single}
    eng.close()

    # ---- per-kernel durations (HIP events on the engine's stream), same start, separate run
    prof = {}
    if args.profile_steps > 0:
        engp = make_engine(1)
        engp.run(args.warmup)
        st, ps, _ = engp.run(args.profile_steps)
        pd = ps.as_dict()
        for k, ms in pd["kernel_ms"].items():
            prof[k] = {"calls": int(pd["kernel_calls"][k]), "avg_us": 1e3 * ms / max(1, pd["kernel_calls"][k])}
        engp.close()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # N>1: the ranks cooperate on ONE pivot stream (pricing sharded, the rest replicated)
    pivots_per_s = args.steps / dt
    price_bytes = 8.0 * ld * nN
    if world > 1:  # each rank prices only its block of the nonbasic positions
        from ellp_amd.dist import shard_ranges
        nt = 8.0 * ld * nN > 160e6
        cpb = max(1, min(64, (nN + 2047) // 2048 if nt else (nN + 1023) // 1024))
        a0, a1 = shard_ranges(nN, cpb, world)[0]
        price_bytes = 8.0 * ld * (a1 - a0)
    # HBM traffic of the pricing kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE, gfx950 x2 read correction) — only valid for the workload it was measured on
    traffic = None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
            pm = json.load(open(f))
            if pm.get("workload") != f"m={m} n={n} {args.solver}" or world != 1:
                continue
            want = "k_price<"
            for kname, kv in pm["kernels"].items():
                # k_price<T, MODE, NT> or k_price_wave<MODE>: pick the pricing kernel of this solver
                mode = "1" if dual else "0"
                if kname.startswith("k_price_wave<"):
                    hit = kname[len("k_price_wave<"):].rstrip(">").strip() == mode
                elif kname.startswith("k_price<"):
                    hit = kname.split(",")[1].strip().rstrip(">") == mode
                else:
                    hit = False
                if hit:
                    traffic = kv["hbm_bytes_per_launch"]
            if traffic is not None:
                break
    except Exception:
        traffic = None
    roofline = None
    pk = "dprice" if dual else "price"
    if pk in prof:
        t_us = prof[pk]["avg_us"]
        ach = price_bytes / (t_us * 1e-6) / 1e9
        roofline = {"kernel": "pricing pass (k_price_wave<%d> / k_price<T,%d,NT>)" % ((1, 1) if dual else (0, 0)), "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": price_bytes, "avg_us": round(t_us, 3)}
    cpu = None
    cpu_pivots = args.cpu_pivots
    if cpu_pivots < 0:
        # ~10-30 s of CPU work: one oracle pivot costs ~(2/3) m^3 flops of unblocked LU
        est = (2.0 / 3.0) * m ** 3 / 7.3e9 + 1e-3  # measured: 0.73 s per pivot at m=2000 on the box's host
        cpu_pivots = int(max(3, min(2000, 15.0 / est)))
    if world > 1:
        cpu_pivots = 0  # the CPU baseline is reported by the N=1 run only
    if cpu_pivots > 0:
        rate, iters, cdt, _ = cpu_baseline(flat, cpu_pivots, args.solver)
        cpu = {"value": round(rate, 4), "unit": "pivots/s", "cores": 1, "kind": "port",
               "sample": f"first {iters} pivots of the same LP from the same starting basis, {cdt:.1f} s, "
                         "oracle/ellp_oracle.c (LU refactor every iteration, single thread)"}
    cpu_same = None
    if cpu_pivots > 0:
        # the ENGINE's algorithm (explicit B^-1, eta updates, same pivot rules)
        # on all host cores, so the GPU/CPU ratio is not only the reference's LU-per-iteration cost
        from oracle import ellp_oracle as eo

        class _V:
            pass

        def fresh():
            v = _V()
            for k, val in flat.items():
                setattr(v, k, val.copy() if hasattr(val, "copy") else val)
            v.nB, v.nN = len(flat["B"]), len(flat["N"])
            return v
        cores = eo.host_threads()
        same_loop = eo.dual_binv_solve_with_initial if dual else eo.primal_binv_solve_with_initial
        _, it_p, _, secs_p = same_loop(fresh(), 40, threads=cores)  # probe the rate
        k_same = int(max(40, min(args.same_alg_pivots, 8.0 * it_p / max(secs_p, 1e-9))))   # <= ~8 s
        st_c, it_c, msg_c, secs_c = same_loop(fresh(), k_same, threads=cores)
        if it_c > 0 and secs_c > 0:
            cpu_same = {"value": round(it_c / secs_c, 2), "unit": "pivots/s", "cores": cores, "kind": "port",
                        "sample": f"first {it_c} pivots of the same LP from the same starting basis, {secs_c:.1f} s, "
                                  "oracle/ellp_oracle.c eo_%s_binv_solve_with_initial (explicit B^-1 + eta " % ("dual" if dual else "primal") +
                                  "updates like the engine, OpenMP over the host cores)"}
    alg_bytes_per_pivot = 8.0 * ld * nN + (24.0 if dual else 32.0) * m * ld
    out = {
        "metric": f"simplex pivots/sec (dense LP, {args.solver}, tableau resident in HBM)",
        "value": round(pivots_per_s, 2), "unit": "pivots/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 6), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"random dense covering LP m={m} n={n} (seed {args.seed}), dual simplex from the "
                                f"slack basis, std-form {m}x{fp.n} with |N|={nN}" if dual else
                                f"random dense LP m={m} n={n} (seed {args.seed}), primal simplex phase 1, "
                                f"std-form {m}x{fp.n} with |N|={nN}"), "refactors_in_window": int(refactors),
                   "inverse_residual_after": resid,
                   "parallelism": ("single GPU" if world == 1 else
                                   f"column-block pricing sharded over {world} GPUs, 1 all-gather/iteration"),
                   "sharded_check": sharded_check},
        "achieved_GBps_algorithmic": round(alg_bytes_per_pivot * args.steps / dt / 1e9, 1),
        "roofline": roofline, "cpu_baseline": cpu, "kernels_us": {k: round(v["avg_us"], 3) for k, v in prof.items()},
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(pivots_per_s / cpu["value"], 1)
    if cpu_same:
        out["cpu_baseline_same_algorithm"] = cpu_same
        out["speedup_vs_cpu_same_algorithm"] = round(pivots_per_s / cpu_same["value"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
