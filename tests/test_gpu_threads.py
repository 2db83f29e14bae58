"""`cargo test` runs the reference's ~57 tests concurrently on threads of one process
(SURVEY.md §4), so whatever sits behind solve_with_initial must be re-entrant: no global mutable
state, one stream per call.  Here the 25 problems x 2 solvers (+ netlib) are solved from a pool of
8 threads through the host mirror (ctypes releases the GIL during the calls) and every result must
equal the one obtained sequentially."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from ellp_amd import DualSimplexSolver, PrimalSimplexSolver, Problem, parse_mps
from helpers import GOLDEN, check_result, known_answers

pytestmark = pytest.mark.gpu
KA = known_answers()
SOLVERS = {"primal": PrimalSimplexSolver, "dual": DualSimplexSolver}


def _solve(job):
    kind, fx, solver = job
    if kind == "fixture":
        prob = Problem.from_fixture(fx)
    else:
        prob = parse_mps(open(os.path.join(GOLDEN, fx["file"])).read())
    result = SOLVERS[solver].default().solve(prob)
    if result.kind == "optimal":
        return result.kind, result.solution.obj(), np.array(result.solution.x()), result.iters
    return result.kind, None, None, result.iters


def test_concurrent_solves_match_sequential():
    jobs = [("fixture", fx, s) for fx in KA["problems"] for s in ("primal", "dual")]
    jobs += [("netlib", fx, s) for fx in KA["netlib"] for s in ("primal", "dual")]
    jobs = jobs * 2  # 112 solves
    sequential = [_solve(j) for j in jobs[:len(jobs) // 2]] * 2
    with ThreadPoolExecutor(max_workers=8) as pool:
        concurrent = list(pool.map(_solve, jobs))
    for job, a, b in zip(jobs, sequential, concurrent):
        assert a[0] == b[0], (job[1]["name"], job[2], a[0], b[0])
        assert a[3] == b[3], (job[1]["name"], job[2], "iteration counts differ")
        if a[0] == "optimal":
            assert a[1] == b[1]
            np.testing.assert_array_equal(a[2], b[2])
        if job[0] == "fixture":
            check_result(job[1], b[0], b[1], b[2], KA["abs_eps"], KA["rel_eps"])
