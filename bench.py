#!/usr/bin/env python3
"""bench.py — simplex pivots/sec + achieved HBM GB/s of the HIP pivot engine.

    python bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the revised-simplex loop (one pass of the hot path:
BTRAN/pricing/entering/FTRAN/ratio/eta-update) on the synthetic dense LP of BASELINE.json's
config 3 (m=2000, n=5000, primal; SURVEY.md §8d generator), started from the reference's own
phase-1 starting basis, tableau already resident in HBM.  Prints ONE JSON line (rank 0).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py …`) BEFORE it touches
torch or HIP, relays their output and exits with their return code.  Under torchrun (WORLD_SIZE
set) it is one of the ranks.  N ranks cooperate on ONE pivot stream: the nonbasic columns (storage
and pricing) are sharded, everything else is replicated ("scaling": "strong").

Besides the headline workload (config 3, so that N=1 is comparable from round to round) every line
carries a `config2` object (netlib AFIRO through the user API against the reference's known answer), a `config4`
object (BASELINE.json's config 4: the same LP shape through the dual loop, N = 1 only) and
a `config5` object: BASELINE.json's config 5 (m=4000, n=40000, primal), the one it names
for 1/2/4/8 GPUs, run the same way — pivots/s, the pricing kernel's rate per GPU and its fraction
of the HBM roofline.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured-achievable)
EVENT_NOTE = ("HIP-event brackets on the engine's stream minus the bracket calibration `event_cost` "
              "(tools/event_cal.hip: an event pair reads that much more than rocprofv3's duration of the "
              "kernel inside it); raw brackets = these + event_cost; the rocprofv3 --kernel-trace --stats "
              "averages of the same command are committed under profiles/")


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
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 engine default, 0 three kernels per iteration, 1 two")
    ap.add_argument("--config5", type=int, default=-1,
                    help="1/0: also run config 5 (m=4000 n=40000) and report it as `config5` (-1: yes when the "
                         "headline workload is config 3)")
    ap.add_argument("--config5-steps", type=int, default=1000)
    ap.add_argument("--config4", type=int, default=-1,
                    help="1/0: also run config 4 (m=2000 n=5000, dual simplex) on one GPU and report it as `config4` "
                         "(-1: yes when the headline is config 3 and N = 1)")
    ap.add_argument("--config4-steps", type=int, default=2000)
    ap.add_argument("--full-solve", type=int, default=1,
                    help="1/0: also solve config 3 to optimality with the steepest-edge extension (`config3_full_solve_steepest_edge`, ~4 s)")
    ap.add_argument("--config4-reference", type=int, default=1,
                    help="1/0: also time the dual loop on the reference's own DualPhase1 arrays of the config-3 LP "
                         "(`config4_reference_phase1`; ~15 s of setup)")
    ap.add_argument("--long-window", type=int, default=3000,
                    help="when --steps < 1000: additionally time a window of this many steps (0 off)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 without torchrun: start the N ranks as a child job.  Nothing here touches HIP
    (torch.cuda.device_count() only counts, it does not initialise a device)."""
    n = args.gpus
    backend = os.environ.get("ELLP_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception as e:  # pragma: no cover
            print(f"bench.py: cannot count HIP devices: {e}", file=sys.stderr)
            return 2
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} HIP devices, this node has {have} "
                  "(ELLP_BENCH_BACKEND=gloo ELLP_BENCH_DEVICE=0 runs the ranks on one device as a dry run)",
                  file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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


class Ctx:
    """process-group context of this rank"""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # test hooks (dry runs on a 1-GPU box): ELLP_BENCH_BACKEND=gloo, ELLP_BENCH_DEVICE=0
        self.backend = os.environ.get("ELLP_BENCH_BACKEND", "nccl")
        if "ELLP_BENCH_DEVICE" in os.environ:
            self.local_rank = int(os.environ["ELLP_BENCH_DEVICE"])
        self.dist = None

    def init(self):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, v):
        if self.dist is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_equal(self, value):
        if self.dist is None:
            return True
        t = self.torch.tensor([value], dtype=self.torch.int64, device="cuda" if self.backend == "nccl" else "cpu")
        parts = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return all(int(p.item()) == value for p in parts)


def _engine_hash():
    from ellp_amd.build import engine_source_hash
    return engine_source_hash()


def _kernel_matches(kname, which, dual):
    """which = 'pricing' | 'dominant' (the pass over B^-1: FTRAN fused with the eta update)"""
    mode = "1" if dual else "0"
    if which == "dominant":
        if dual:
            return kname.startswith("k_dual_fu<")
        return kname.startswith("k_ftran_eta<") and not kname.rstrip(">").endswith("true")
    if kname.startswith("k_price2_wave") or kname.startswith("k_price2<"):  # two-launch pipeline: primal only
        if kname.rstrip(">").endswith("true") and kname.startswith("k_price2_wave"):
            return False  # the column-sharded instantiation
        return not dual
    if kname.startswith("k_price_wave<"):
        return kname[len("k_price_wave<"):].rstrip(">").strip() == mode
    if kname.startswith("k_price<"):
        return kname.split(",")[1].strip().rstrip(">") == mode
    return False


def pmc_traffic(m, n, solver, dual, which="pricing"):
    """HBM traffic of a kernel per launch, from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, gfx950 x2 read correction; tools/pmc_traffic.sh).  Read from profiles/, not measured in this run: the
    file it came from is named beside it, and a file that was measured on OTHER kernels than the ones this run executes
    (its engine_source_hash differs from the sources') is refused — returns (None, why)."""
    import glob
    want = _engine_hash()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if pm.get("workload") != f"m={m} n={n} {solver}":
            continue
        if pm.get("engine_source_hash") != want:
            stale = stale or os.path.relpath(f, ROOT)
            continue
        for kname, kv in pm["kernels"].items():
            if _kernel_matches(kname, which, dual):
                return kv["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, (f"no PMC file for the current engine sources ({want}); newest older one: {stale}" if stale else None)


def rocprof_avg_us(which, dual, accept=None):
    """average duration of a kernel in a committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/*default_kernel_stats.csv) — only from a summary whose side file (.meta.json, tools/rocprof_bench.sh) says
    it was taken on the current engine sources"""
    import csv
    import glob
    want = _engine_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*default_kernel_stats.csv")), reverse=True):
        meta = f[:-4] + ".meta.json"
        try:
            if json.load(open(meta)).get("engine_source_hash") != want:
                continue
            rows = list(csv.DictReader(open(f)))
        except Exception:
            continue
        for row in rows:
            name = row.get("Name") or row.get("KernelName") or ""
            mm = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
            short = mm.group(1) if mm else ""
            if short and _kernel_matches(short, which, dual) and (accept is None or accept(short)):
                try:
                    avg_ns = float(row.get("AverageNs") or row.get("Average") or 0.0)
                except ValueError:
                    continue
                if avg_ns > 0:
                    return round(avg_ns / 1e3, 3), short, os.path.relpath(f, ROOT)
    return None, None, None


def measure(ctx, args, m, n, seed, solver, steps, warmup, profile_steps, long_window=0, flat_override=None):
    """One workload on this process group: W warm-up steps, K timed steps bracketed by barrier +
    synchronize on both sides (max over ranks), an optional longer window, per-kernel event times from
    a second run, and — sharded — the self-check against a single-GPU replay."""
    import numpy as np
    from ellp_amd import _engine as E
    from ellp_amd import synth
    torch = ctx.torch
    world, rank = ctx.world, ctx.rank
    dual = solver == "dual"
    flat = flat_override or (synth.dual_start_flat(seed, m, n) if dual else synth.primal_phase1_flat(seed, m, n))

    def make_fp():
        return E.FlatProblem(flat["m"], flat["n"], flat["n_c"], flat["A"], flat["c"], flat["b"], flat["kind"],
                             flat["lb"], flat["ub"], flat["x"], flat["B"], flat["N"], flat["Nb"], flat.get("y"),
                             flat.get("d"))
    fp = make_fp()
    kind = E.ENGINE_DUAL if dual else E.ENGINE_PRIMAL
    nN = fp.nN
    ld = (m + 15) // 16 * 16

    def opts_for(profile, pipeline=None, flags=0):
        kw = dict(max_iter=None, device=ctx.local_rank, refactor_period=args.refactor_period,
                  btran_mode=args.btran_mode, poll_interval=args.poll, profile=profile, flags=flags)
        if hasattr(E.Opts, "pipeline"):  # ellp_opts.pipeline: 0 engine default, 1 three launches, 2 two
            kw["pipeline"] = (args.pipeline if pipeline is None else pipeline) + 1
        return E.default_opts(**kw)

    def make_engine(profile, f=None):
        if world > 1:
            # primal: the column-sharded loop runs the pipeline the engine was created with (two launches from m = 384:
            # pricing | exchange | eta update + FTRAN) and the single-GPU replay of the self-check below is built with
            # the same options, so both sides run the same kernels.  dual: the replicated stepped loop drives the
            # three-launch kernels; say so explicitly for both sides (the two-launch form sums the FTRAN dot products
            # in another order: a tie could fall the other way and read as a divergence)
            from ellp_amd.dist import ShardedEngine
            return ShardedEngine(kind, f or fp, opts_for(profile, pipeline=None if solver == "primal" else 0))
        return E.Engine(kind, f or fp, opts_for(profile))

    # ---- timed region: tableau resident, W warm-up steps, then exactly K steps
    eng = make_engine(0)
    st, stats, msg = eng.run(warmup)
    assert st in (E.MAXITER,), f"warm-up ended the solve: {E.STATUS_NAME.get(st)} {msg}"
    it0 = stats.iters
    ctx.barrier()
    t0 = time.perf_counter()
    st, stats, msg = eng.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.barrier()
    dt = ctx.max_over_ranks(dt)
    steps_done = stats.iters - it0
    assert st == E.MAXITER and steps_done == steps, (E.STATUS_NAME.get(st), steps_done, msg)
    out = {"dt": dt, "steps": steps, "refactors": int(stats.refactors), "ld": ld, "nN": nN, "n_cols": fp.n,
           "flat": flat}
    total_after = warmup + steps
    if long_window > 0:
        it1 = stats.iters
        ctx.barrier()
        t0 = time.perf_counter()
        st, stats, msg = eng.run(long_window)
        torch.cuda.synchronize()
        ldt = time.perf_counter() - t0
        ctx.barrier()
        ldt = ctx.max_over_ranks(ldt)
        assert st == E.MAXITER and stats.iters - it1 == long_window, (E.STATUS_NAME.get(st), msg)
        out["long_window"] = {"steps": long_window, "value": round(long_window / ldt, 2), "unit": "pivots/s",
                              "ms_per_step": round(1e3 * ldt / long_window, 6)}
        total_after += long_window
    out["resid"] = eng.inverse_residual() if world == 1 else None
    if world > 1:
        # outside the timed region: the sharded run must have taken the pivots of a single-GPU run
        # (rank 0 replays them unsharded) and every rank must hold the same basis
        import zlib
        eng.read_point()
        crc = zlib.crc32(fp.B.tobytes() + fp.N.tobytes() + fp.Nb.tobytes())
        ranks_agree = ctx.all_equal(crc)
        same_as_single = None
        if rank == 0:
            fp1 = make_fp()
            # the sharded loop's kernels; flags = 8 (ELLP_FLAG_NO_CERTIFY): a sharded engine runs without the pivot guard and the
            # certificates of the unsharded default, so the replay does too — the same pivots on both sides by construction
            ref = E.Engine(kind, fp1, opts_for(0, pipeline=None if solver == "primal" else 0, flags=8))
            ref.run(total_after)
            ref.read_point()
            ref.close()
            same_as_single = bool(np.array_equal(fp1.B, fp.B) and np.array_equal(fp1.N, fp.N) and
                                  np.array_equal(fp1.Nb, fp.Nb))
        out["sharded_check"] = {"all_ranks_hold_the_same_basis": bool(ranks_agree),
                                "same_pivots_as_one_gpu": same_as_single,
                                "exchange": getattr(eng, "exchange_name", None)}
        ok = ranks_agree and (same_as_single is None or same_as_single)
        if not ctx.all_equal(1 if ok else 0) or not ok:
            if rank == 0:
                print(json.dumps({"error": "sharded run diverged from the single-GPU pivots", "workload": f"m={m} n={n}",
                                  "sharded_check": out["sharded_check"]}), file=sys.stderr)
            eng.close()
            xname = out["sharded_check"]["exchange"] or ""
            if "mailbox" in xname and os.environ.get("ELLP_DIST_EXCHANGE") != "rccl":
                # the peer-to-peer mailbox passed its self-test and still delivered something wrong: measure
                # again over RCCL (every rank takes this branch: the verdict above is all-reduced) and say so
                os.environ["ELLP_DIST_EXCHANGE"] = "rccl"
                again = measure(ctx, args, m, n, seed, solver, steps, warmup, profile_steps, long_window)
                again["mailbox_failed"] = True
                return again
            raise SystemExit(3)
    eng.close()

    # ---- per-kernel durations (HIP events on the engine's stream), same start, separate run
    prof = {}
    if profile_steps > 0:
        engp = make_engine(1, make_fp())
        engp.run(warmup)
        st, ps, _ = engp.run(profile_steps)
        pd = ps.as_dict()
        for k, ms in pd["kernel_ms"].items():
            prof[k] = {"calls": int(pd["kernel_calls"][k]), "avg_us": 1e3 * ms / max(1, pd["kernel_calls"][k])}
        engp.close()
    out["prof"] = prof
    # bytes the pricing launch of ONE rank streams (its column block)
    price_cols = nN
    if world > 1:
        from ellp_amd.dist import shard_ranges
        nt = 8.0 * ld * nN > 160e6
        cpb = max(1, min(64, (nN + 2047) // 2048 if nt else (nN + 1023) // 1024))
        a0, a1 = shard_ranges(nN, cpb, world)[0]
        price_cols = a1 - a0
    out["price_bytes"] = 8.0 * ld * price_cols
    out["alg_bytes_per_pivot"] = 8.0 * ld * nN + (24.0 if dual else 32.0) * m * ld
    # what the engine really moves per pivot: one pass over A_N, and ONE pass over B^-1 that reads and rewrites it
    # (two-launch forms, m >= 384: the eta update is fused with FTRAN; BTRAN is an O(m) update of u) — three-launch
    # form: + 8 m ld for the separate FTRAN read
    two_pass = m >= 384 and (world == 1 or not dual) and args.pipeline != 0
    out["engine_bytes_per_pivot"] = 8.0 * ld * nN + (16.0 if two_pass else 24.0) * m * ld
    # what the engine moves per iteration: BTRAN is O(m) incremental, and with the two-kernel pipeline
    # the eta update and the next FTRAN share one pass over B^-1
    return out


def roofline_of(meas, m, n, solver, world):
    """roofline of the pricing pass (SURVEY.md §8d's headline kernel); roofline_dominant(): the pass over B^-1"""
    dual = solver == "dual"
    pk = "dprice" if dual else "price"
    prof = meas["prof"]
    if pk not in prof:
        return None
    traffic, src = (None, None)
    if world == 1:
        traffic, src = pmc_traffic(m, n, solver, dual)
    t_us = prof[pk]["avg_us"]
    ach = meas["price_bytes"] / (t_us * 1e-6) / 1e9
    # the summary of the default command holds config 3's, 4's and 5's kernels: config 3 prices with the wave-per-column
    # kernel, config 5 (A_N far beyond the Infinity Cache) with the block kernel
    big = (m, n) == (4000, 40000)
    rp_us, rp_name, rp_file = (rocprof_avg_us("pricing", dual, (lambda k: k.startswith("k_price2<")) if big else (lambda k: "wave" in k))
                               if (world == 1 and (m, n) in ((2000, 5000), (4000, 40000))) else (None, None, None))
    return {"kernel": "pricing pass of one GPU (k_price*<%d>)" % (1 if dual else 0), "bound": "hbm",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": (src + " (rocprofv3 --pmc passes of an earlier run of this command on the same engine sources, "
                                     "not measured in this run)") if traffic else src,
            "frac_traffic": round(traffic / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "avg_us_rocprof": rp_us, "avg_us_rocprof_kernel": rp_name,
            "avg_us_rocprof_source": (rp_file + " (rocprofv3 --kernel-trace --stats of an earlier run of this command on the same engine sources)") if rp_file else None,
            "frac_of_measured_achievable": round(ach / 6290.0, 4),
            "note": "achieved / frac = ALGORITHMIC bytes of the pass (SURVEY.md §8d: 8 m |N|, every nonbasic column once) / its "
                    "duration — the contract's definition, NOT an HBM-utilisation figure: the kernel streams fewer bytes (unit "
                    "columns — slacks, artificials: a third of config 3's nonbasic columns — are priced from their single entry; "
                    "ellp_opts.flags = 1 streams everything, tools/unit_columns_ab.py has both).  frac_traffic = the bytes HBM "
                    "really delivered (`traffic`, PMC) / duration / peak is the utilisation.  peak = the 8 TB/s spec; "
                    "/opt/skills/guides/MI355X_MICROARCH.md measures 6.29 TB/s as achievable by a pure streaming read.  The primal "
                    "kernel's duration includes the ratio-test fold of the previous iteration in its prologue (two-launch "
                    "pipeline), about 2.5 us before the first column is read.",
            "bytes_per_launch": meas["price_bytes"], "avg_us": round(t_us, 3),
            "avg_us_raw": round(t_us + prof.get("event_cost", {}).get("avg_us", 0.0), 3), "timing": EVENT_NOTE}


def roofline_dominant(meas, m, n, solver, world):
    """the pass over B^-1 — FTRAN fused with the eta update (k_ftran_eta / k_dual_fu): the longer of the two launches of
    an iteration at config 3 / 4.  SURVEY §8d bytes: FTRAN 8 m^2 + update 16 m^2; moved: one read + one write of B^-1."""
    dual = solver == "dual"
    prof = meas["prof"]
    if "ftran" not in prof or m < 384:
        return None
    ld = (m + 15) // 16 * 16
    t_us = prof["ftran"]["avg_us"]
    alg, moved = 24.0 * m * ld, 16.0 * m * ld
    traffic, src = pmc_traffic(m, n, solver, dual, "dominant") if world == 1 else (None, None)
    rp_us, rp_name, rp_file = (rocprof_avg_us("dominant", dual, lambda k: ("<8" in k) == (m > 2048))  # this configuration's instantiation
                               if world == 1 else (None, None, None))
    ach = alg / (t_us * 1e-6) / 1e9
    return {"kernel": "k_dual_fu (FTRAN + eta update + x_B / d / y updates)" if dual else "k_ftran_eta (eta update of the previous pivot + FTRAN)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "bytes_per_launch": alg, "bytes_moved_per_launch": moved,
            "frac_moved": round(moved / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": src,
            "frac_traffic": round(traffic / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "avg_us": round(t_us, 3), "avg_us_rocprof": rp_us, "avg_us_rocprof_kernel": rp_name, "avg_us_rocprof_source": rp_file,
            "note": "frac = SURVEY.md §8d's bytes for the steps this kernel replaces (FTRAN 8 m^2 + rank-1 update 16 m^2) / its "
                    "duration; frac_moved = the bytes the fused pass has to move (one read and one write of B^-1, 16 m ld); "
                    "tools/copy_floor.hip puts a copy of this shape at 11.4 us (config 3)."}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    ctx = Ctx()
    if args.gpus > 1 and ctx.world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ctx.world}")
    ctx.init()
    world, rank = ctx.world, ctx.rank
    m, n = args.m, args.n
    dual = args.solver == "dual"
    is_c3 = (m, n, args.seed, args.solver) == (2000, 5000, 20260301, "primal")
    want_c5 = args.config5 == 1 or (args.config5 < 0 and is_c3)
    long_window = args.long_window if (args.steps < 1000 and args.long_window > 0) else 0

    head = measure(ctx, args, m, n, args.seed, args.solver, args.steps, args.warmup, args.profile_steps, long_window)
    c5 = None
    if want_c5:
        c5m = measure(ctx, args, 4000, 40000, 20260305, "primal", args.config5_steps, 100,
                      min(args.profile_steps, 100))
        if rank == 0:
            r5 = roofline_of(c5m, 4000, 40000, "primal", world)
            c5_bytes = c5m["alg_bytes_per_pivot"]
            c5 = {"workload": f"random dense LP m=4000 n=40000 (seed 20260305), primal simplex phase 1, std-form "
                              f"4000x{c5m['n_cols']} with |N|={c5m['nN']}, nonbasic columns sharded over {world} GPU(s)",
                  "value": round(c5m["steps"] / c5m["dt"], 2), "unit": "pivots/s", "steps": c5m["steps"],
                  "ms_per_step": round(1e3 * c5m["dt"] / c5m["steps"], 6),
                  "pricing_GBps_per_gpu": r5["achieved"] if r5 else None,
                  "pricing_frac_of_hbm_peak": r5["frac"] if r5 else None,
                  "pricing_GBps_all_gpus": round(r5["achieved"] * world, 1) if r5 else None,
                  "roofline": r5, "roofline_eta_pass": roofline_dominant(c5m, 4000, 40000, "primal", world),
                  "kernels_us": {k: round(v["avg_us"], 3) for k, v in c5m["prof"].items()},
                  "achieved_GBps_algorithmic": round(c5_bytes * c5m["steps"] / c5m["dt"] / 1e9, 1),
                  "engine_GBps": round(c5m["engine_bytes_per_pivot"] * c5m["steps"] / c5m["dt"] / 1e9, 1),
                  "sharded_check": c5m.get("sharded_check")}
        del c5m
    c4 = None
    if (args.config4 == 1 or (args.config4 < 0 and is_c3)) and world == 1:
        c4m = measure(ctx, args, 2000, 5000, args.seed, "dual", args.config4_steps, 200, min(args.profile_steps, 100))
        r4 = roofline_of(c4m, 2000, 5000, "dual", 1)
        c4 = {"workload": f"random dense covering LP m=2000 n=5000 (seed {args.seed}), dual simplex from the slack basis, "
                          f"std-form 2000x{c4m['n_cols']} with |N|={c4m['nN']}",
              "value": round(c4m["steps"] / c4m["dt"], 2), "unit": "pivots/s", "steps": c4m["steps"],
              "ms_per_step": round(1e3 * c4m["dt"] / c4m["steps"], 6), "roofline": r4,
              "roofline_dominant": roofline_dominant(c4m, 2000, 5000, "dual", 1),
              "kernels_us": {k: round(v["avg_us"], 3) for k, v in c4m["prof"].items()},
              "achieved_GBps_algorithmic": round(c4m["alg_bytes_per_pivot"] * c4m["steps"] / c4m["dt"] / 1e9, 1),
              "engine_GBps": round(c4m["engine_bytes_per_pivot"] * c4m["steps"] / c4m["dt"] / 1e9, 1)}
        del c4m
    c4ref = None
    if (args.config4 == 1 or (args.config4 < 0 and is_c3)) and world == 1 and args.config4_reference:
        # config 4 on the reference's OWN dual phase-1 arrays (dual_problem.rs:89-256: the box problem of the config-3 LP,
        # basis from the LU of A^T, nonbasics at lower / upper by the sign of d), built by this repository's host mirror
        # of ellp's setup (DualPhase1::from_problem: rank check and LU of A^T on the device, the rest on the host)
        try:
            import numpy as np
            from ellp_amd import Bound, ConstraintOp, Problem, synth
            t0 = time.perf_counter()
            A, b, c = synth.dense_lp(args.seed, 2000, 5000)
            p = Problem()
            ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(5000)]
            for i in range(2000):
                p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
            f = p._debug_phase1("dual")
            t_setup = time.perf_counter() - t0
            flat4 = {"m": f["m"], "n": f["n"], "n_c": f["n_c"], "A": f["A"], "c": f["c"], "b": f["b"], "kind": f["kind"],
                     "lb": f["lb"], "ub": f["ub"], "x": f["x"], "B": f["B"], "N": f["N"], "Nb": f["Nb"], "y": f["y"], "d": f["d"]}
            c4r = measure(ctx, args, 2000, 5000, args.seed, "dual", args.config4_steps, 200, min(args.profile_steps, 100),
                          flat_override=flat4)
            r4r = roofline_of(c4r, 2000, 5000, "dual", 1)
            nb = np.bincount(np.asarray(f["Nb"], dtype=np.int64), minlength=3)
            c4ref = {"workload": f"dual phase 1 of the config-3 LP exactly as the reference builds it (dual_problem.rs:89-256): box "
                                 f"problem 2000x{c4r['n_cols']}, every variable TwoSided, |N|={c4r['nN']} with {int(nb[0])} nonbasics at "
                                 f"their lower and {int(nb[1])} at their upper bound, basis from the LU of A^T",
                     "value": round(c4r["steps"] / c4r["dt"], 2), "unit": "pivots/s", "steps": c4r["steps"],
                     "ms_per_step": round(1e3 * c4r["dt"] / c4r["steps"], 6), "roofline": r4r,
                     "kernels_us": {k: round(v["avg_us"], 3) for k, v in c4r["prof"].items()},
                     "engine_GBps": round(c4r["engine_bytes_per_pivot"] * c4r["steps"] / c4r["dt"] / 1e9, 1),
                     "setup_s": round(t_setup, 1)}
            del c4r, flat4, f
        except Exception as ex:  # the headline must not depend on it
            c4ref = {"error": str(ex)[:300]}
    c3se = None
    if is_c3 and world == 1 and args.full_solve:
        # config 3 to OPTIMALITY through the user API with the opt-in steepest-edge extension (ellp_opts.flags = 4; not the
        # reference's pricing rule — its Dantzig rule needs 655 k pivots / 25 s for the same solve, tests/test_gpu_fullsolve.py);
        # objective against the committed independent optimum (tests/golden/synth_optimum_*.json, SciPy-HiGHS)
        try:
            from ellp_amd import Bound, ConstraintOp, PrimalSimplexSolver, Problem, synth
            A, b, c = synth.dense_lp(args.seed, 2000, 5000)
            p = Problem()
            ids = [p.add_var(float(c[j]), Bound.Lower(0.0)) for j in range(5000)]
            for i in range(2000):
                p.add_constraint(list(zip(ids, A[i].tolist())), ConstraintOp.Lte, float(b[i]))
            t0 = time.perf_counter()
            res = PrimalSimplexSolver.new(None, flags=4).solve(p)
            dts = time.perf_counter() - t0
            fxp = os.path.join(ROOT, "tests", "golden", f"synth_optimum_{args.seed}_2000x5000.json")
            ref = json.load(open(fxp))["objective"] if os.path.exists(fxp) else None
            c3se = {"workload": "config 3's LP solved to optimality, PrimalSimplexSolver::new(None).solve with steepest-edge pricing "
                                "(an opt-in extension; set-up included: standard form, rank check on the device, both phases)",
                    "status": res.kind, "iterations_phase1_phase2": list(res.iters), "solve_s": round(dts, 3),
                    "objective": res.solution.obj() if res.kind == "optimal" else None, "reference_objective_highs": ref,
                    "reference_rule_for_comparison": "Dantzig: 654,976 iterations, 24.8 s (profiles/r03_steepest_edge_time.json)"}
            if c3se["objective"] is not None and ref:
                c3se["rel_diff"] = abs(c3se["objective"] - ref) / abs(ref)
            del A, p
        except Exception as ex:  # the headline must not depend on it
            c3se = {"error": str(ex)[:300]}
    c2 = None
    if is_c3 and world == 1:
        # BASELINE.json's config 2: netlib AFIRO through the user API (parse_mps -> PrimalSimplexSolver::new(None).solve),
        # objective against the reference's known answer (tests/problems/mod.rs:661)
        try:
            from ellp_amd import PrimalSimplexSolver, parse_mps
            text = open(os.path.join(ROOT, "tests", "golden", "netlib", "afiro.mps")).read()
            best, res = None, None
            for rep in range(4):
                prob = parse_mps(text)
                t0 = time.perf_counter()
                res = PrimalSimplexSolver.new(None).solve(prob)
                dt2 = time.perf_counter() - t0
                if rep:
                    best = dt2 if best is None else min(best, dt2)
            ref = -464.75314286
            c2 = {"workload": "netlib AFIRO (27 x 32), primal simplex through the user API, both phases on one resident engine "
                              "(the persistent one-workgroup kernel: an LU per iteration, bit for bit the CPU restatement)",
                  "status": res.kind, "objective": res.solution.obj() if res.kind == "optimal" else None,
                  "reference_objective": ref, "iterations_phase1_phase2": list(res.iters), "solve_ms": round(best * 1e3, 3)}
            if c2["objective"] is not None:
                c2["rel_diff"] = abs(c2["objective"] / ref - 1.0)
        except Exception as ex:  # the headline must not depend on it
            c2 = {"error": str(ex)[:200]}
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    if rank != 0:
        return
    dt, steps = head["dt"], head["steps"]
    pivots_per_s = steps / dt
    flat = head["flat"]
    roofline = roofline_of(head, m, n, args.solver, world)
    roof_dom = roofline_dominant(head, m, n, args.solver, world)
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
    ld, nN = head["ld"], head["nN"]
    alg_bytes_per_pivot = head["alg_bytes_per_pivot"]
    ms_per_step = 1e3 * dt / steps
    alg_gbps = alg_bytes_per_pivot * steps / dt / 1e9
    prof = head["prof"]
    out = {
        "metric": f"simplex pivots/sec (dense LP, {args.solver}, tableau resident in HBM)",
        "value": round(pivots_per_s, 2), "unit": "pivots/s", "n_gpus": world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 6), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"random dense covering LP m={m} n={n} (seed {args.seed}), dual simplex from the "
                                f"slack basis, std-form {m}x{head['n_cols']} with |N|={nN}" if dual else
                                f"random dense LP m={m} n={n} (seed {args.seed}), primal simplex phase 1, "
                                f"std-form {m}x{head['n_cols']} with |N|={nN}"),
                   "refactors_in_window": head["refactors"], "inverse_residual_after": head["resid"],
                   "parallelism": ("single GPU" if world == 1 else
                                   f"nonbasic columns (storage + pricing) sharded over {world} GPUs, one small "
                                   "exchange per iteration; B^-1 and the point replicated"),
                   "sharded_check": head.get("sharded_check"),
                   "mailbox_failed_fell_back_to_rccl": bool(head.get("mailbox_failed", False))},
        "achieved_GBps_algorithmic": round(alg_gbps, 1),
        "iteration_roofline": {"bytes_per_step_engine": head["engine_bytes_per_pivot"],
                               "achieved": round(head["engine_bytes_per_pivot"] * steps / dt / 1e9, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(head["engine_bytes_per_pivot"] * steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "bytes of the two passes as if every nonbasic column were streamed (8 ld |N| for the pricing "
                                       "pass + 16 m ld for the one pass that reads and rewrites B^-1; 24 m ld on the three-launch "
                                       "form) / ms_per_step / 8 TB/s.  The pricing pass streams fewer (unit columns, see "
                                       "roofline.note): traffic_bytes_per_step / frac_traffic are the PMC bytes of the two kernels "
                                       "(None without a counter file for the current engine sources).",
                               "traffic_bytes_per_step": (roofline["traffic"] + roof_dom["traffic"]) if (roofline and roof_dom and roofline.get("traffic") and roof_dom.get("traffic")) else None,
                               "frac_traffic": round((roofline["traffic"] + roof_dom["traffic"]) * steps / dt / 1e9 / HBM_PEAK_GBS, 4) if (roofline and roof_dom and roofline.get("traffic") and roof_dom.get("traffic")) else None,
                               "survey_bytes_per_step": alg_bytes_per_pivot,
                               "survey_frac": round(alg_gbps / HBM_PEAK_GBS, 4),
                               "survey_note": "SURVEY.md §8d's figure (8 m |N| + 32 m^2 primal, 24 m^2 dual: a B^-1 GEMV for BTRAN, "
                                              "one for FTRAN, a read + write for the update) counts bytes this engine no longer "
                                              "moves; kept for comparison with earlier rounds, not a claim"},
        "roofline": roofline, "roofline_dominant": roof_dom, "cpu_baseline": cpu,
        "kernels_us": {k: round(v["avg_us"], 3) for k, v in prof.items()}, "kernels_us_note": EVENT_NOTE,
    }
    if "long_window" in head:
        lw = head["long_window"]
        lw["vs_value"] = round(lw["value"] / pivots_per_s, 4)
        out["long_window"] = lw
    if c2 is not None:
        out["config2"] = c2
    if c4 is not None:
        out["config4"] = c4
    if c4ref is not None:
        out["config4_reference_phase1"] = c4ref
    if c3se is not None:
        out["config3_full_solve_steepest_edge"] = c3se
    if c5 is not None:
        out["config5"] = c5
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(pivots_per_s / cpu["value"], 1)
    if cpu_same:
        out["cpu_baseline_same_algorithm"] = cpu_same
        out["speedup_vs_cpu_same_algorithm"] = round(pivots_per_s / cpu_same["value"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
