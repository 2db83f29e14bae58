"""The oracle's restatement of the CERTIFIED HYBRID (oracle/ellp_oracle.c, hybrid_run; DESIGN.md §3.1c — the engine's default
policy for 128 < m <= 1024, NOT the reference's loop): the explicit-inverse loop with a pivot guard, every terminal status and
every guarded iteration handed to the LU-per-iteration loop, certify or redo.  It changes the path, never the answer: at the
seam, phase by phase from the same arrays, it must end as the reference's own loop ends on every known answer of the reference
(tests/problems/mod.rs:130-674) and on netlib — and on the block-diagonal replications it reaches the pinned optimum where
the plain explicit-inverse loop does not.  CPU only."""
import os
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, blockdiag, known_answers, permuted_fixture, read_mps
from oracle import ellp_oracle as eo

KA = known_answers()


def _phases(prob, which, loop):
    """both phases at the seam with `loop` (a function view -> (status, iters)); returns [(status, objective), ...] per phase run"""
    out = []
    if which == "primal":
        p1, err = eo.primal_phase1(prob)
        if p1 is None or err:
            return None
        v = p1.view()
        if v.m == 0:
            return None
        st, it = loop(v, False)
        out.append((st, v.obj()))
        if st != eo.OPTIMAL or not (-1e-10 < v.obj() < 1e-10):
            return out
        p1.store_point(v)
        v2 = eo.primal_phase2(p1).view()
        st2, it2 = loop(v2, False)
        out.append((st2, v2.obj()))
        return out
    d1, err = eo.dual_phase1(prob)
    if d1 is None or err:
        return None
    v = d1.view()
    if v.m == 0:
        return None
    st, it = loop(v, True)
    d1.store_point(v)
    out.append((st, d1.dual_obj() if st == eo.OPTIMAL else None))
    if st != eo.OPTIMAL or not (d1.dual_obj() > -1e-10):
        return out
    d2, err2 = eo.dual_phase2(d1)
    if d2 is None or err2:
        return out + [("d2-setup", None)]
    v2 = d2.view()
    if v2.m == 0:
        return out
    st2, it2 = loop(v2, True)
    out.append((st2, v2.obj() if st2 == eo.OPTIMAL else None))
    return out


def _lu(v, dual):
    st, it, msg = (eo.dual_solve_with_initial if dual else eo.primal_solve_with_initial)(v, 200000)
    return st, it


def _hybrid(**kw):
    def run(v, dual):
        st, it, msg, cnt = (eo.dual_hybrid_solve_with_initial if dual else eo.primal_hybrid_solve_with_initial)(v, 200000, **kw)
        run.counters = [a + b for a, b in zip(run.counters, cnt)]
        return st, it
    run.counters = [0, 0, 0, 0, 0]
    return run


def _same(a, b):
    assert (a is None) == (b is None)
    if a is None:
        return
    assert len(a) == len(b), (a, b)
    for (sa, oa), (sb, ob) in zip(a, b):
        assert sa == sb, (a, b)
        if oa is not None and ob is not None:
            assert abs(oa - ob) <= 1e-9 * (1.0 + abs(oa)), (a, b)


@pytest.mark.parametrize("which", ["primal", "dual"])
@pytest.mark.parametrize("fx", KA["problems"], ids=[p["name"] for p in KA["problems"]])
def test_known_answers_end_as_under_the_reference_loop(fx, which):
    prob = eo.Problem.from_fixture(fx)
    _same(_phases(prob, which, _lu), _phases(prob, which, _hybrid()))


@pytest.mark.parametrize("which", ["primal", "dual"])
@pytest.mark.parametrize("fx", KA["netlib"], ids=[p["name"] for p in KA["netlib"]])
def test_netlib_ends_as_under_the_reference_loop(fx, which):
    prob = eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"])))
    h = _hybrid()
    res = _phases(prob, which, h)
    assert res is not None and res[-1][0] == eo.OPTIMAL and abs(res[-1][1] / fx["obj"] - 1.0) < 1e-6, res
    assert h.counters[1] >= 2  # both phase ends were examined by the exact loop


def test_a_guard_that_refuses_every_small_pivot_changes_nothing_but_the_path():
    """guard_abs = 0.3: the exact loop takes over again and again on ADLITTLE; the same optimum"""
    fx = next(p for p in KA["netlib"] if p["name"] == "adlittle")
    prob = eo.Problem.from_fixture(read_mps(os.path.join(GOLDEN, fx["file"])))
    for which in ("primal", "dual"):
        h = _hybrid(guard_abs=0.3, K=3)
        res = _phases(prob, which, h)
        assert res[-1][0] == eo.OPTIMAL and abs(res[-1][1] / fx["obj"] - 1.0) < 1e-6, (which, res)
        assert h.counters[0] >= 2, h.counters


def test_block_diagonal_orders_reach_the_pinned_optimum():
    """ADLITTLE x 3 and BLEND x 2, the first 10 orders of the GPU suite's (tests/test_gpu_hybrid.py): the hybrid reaches the pinned
    optimum on every primal order — on BLEND x 2 the plain explicit-inverse loop loses some of them (tests/campaign/hybrid_cpu.py: 5 of
    60: a pivot on a structural zero) — and on the dual wherever the reference's own loop does (tests/golden/blockdiag_orders.json)"""
    import json
    gold = json.load(open(os.path.join(GOLDEN, "blockdiag_orders.json")))
    for name, copies in (("adlittle", 3), ("blend", 2)):
        ka = next(p for p in KA["netlib"] if p["name"] == name)
        base = blockdiag(read_mps(os.path.join(GOLDEN, ka["file"])), copies)
        rng = np.random.default_rng(zlib.crc32(f"{name}x{copies}".encode()))
        want = copies * ka["obj"]
        for trial in range(10):
            prob = eo.Problem.from_fixture(permuted_fixture(base, rng))
            res = _phases(prob, "primal", _hybrid())
            assert len(res) == 2 and res[1][0] == eo.OPTIMAL and abs(res[1][1] / want - 1.0) < 1e-9, (name, trial, res)
            res = _phases(prob, "dual", _hybrid())
            ok = len(res) == 2 and res[1][0] == eo.OPTIMAL and abs(res[1][1] / want - 1.0) < 1e-9
            ref = gold[f"{name}x{copies}"]["dual:lu"][str(trial)]
            ref_ok = ref[0] == "d2" and ref[1] == eo.OPTIMAL
            assert ok or not ref_ok or res[-1][0] == "d2-setup", (name, trial, res, ref)
