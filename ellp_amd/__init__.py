"""ellp_amd — MI355X-native revised-simplex pivot engine behind kehlert/ellp's solver API.

The names exported here mirror the reference crate's public API (src/lib.rs:109-129):

    Problem, Bound, ConstraintOp, PrimalSimplexSolver, DualSimplexSolver, SolverResult,
    EllPError, parse_mps

`Problem` / `StandardForm` / phase construction live in the C++ host mirror
(ellp_amd/csrc/host, libellp_host.so); the per-iteration simplex loops
(`solve_with_initial`) run on the GPU through the C ABI of include/ellp_hip.h
(libellp_hip.so, hand-written HIP for gfx950).  There is no CPU implementation of the loops in
this package: without the built libraries / a HIP device, solving raises.
"""
import ctypes as C
import os

import numpy as np

from . import _engine
from ._engine import MAX_ITER_NONE, Opts

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libellp_host.so")

__all__ = ["Problem", "Bound", "ConstraintOp", "PrimalSimplexSolver", "DualSimplexSolver", "SolverResult",
           "EllPError", "MpsParsingError", "parse_mps"]


class EllPError(Exception):
    """src/error.rs:3-11"""


class MpsParsingError(Exception):
    """src/parse_mps.rs:11-15"""


class Bound:
    """src/problem.rs:190-197. Bound.Free / Lower(lb) / Upper(ub) / TwoSided(lb, ub) / Fixed(v)."""
    KINDS = ("Free", "Lower", "Upper", "TwoSided", "Fixed")

    def __init__(self, kind, lb=0.0, ub=0.0):
        self.kind, self.lb, self.ub = kind, float(lb), float(ub)

    Free = None  # set below

    @staticmethod
    def Lower(lb):
        return Bound(1, lb, 0.0)

    @staticmethod
    def Upper(ub):
        return Bound(2, 0.0, ub)

    @staticmethod
    def TwoSided(lb, ub):
        return Bound(3, lb, ub)

    @staticmethod
    def Fixed(v):
        return Bound(4, v, v)

    @staticmethod
    def from_fixture(b):
        kind, lb, ub = b
        k = Bound.KINDS.index(kind)
        return Bound(k, lb, lb if k == 4 else ub)

    def __repr__(self):
        return f"Bound.{self.KINDS[self.kind]}({self.lb}, {self.ub})"


Bound.Free = Bound(0)


class ConstraintOp:
    """src/problem.rs:298-303"""
    Lte, Eq, Gte = 0, 1, 2
    _BY_NAME = {"Lte": 0, "Eq": 1, "Gte": 2}


class _Result(C.Structure):
    _fields_ = [("status", C.c_int), ("obj", C.c_double), ("nx", C.c_int64), ("x", C.POINTER(C.c_double)),
                ("iters_phase1", C.c_uint64), ("iters_phase2", C.c_uint64), ("err", C.c_char * 512)]


class _FlatPhase(C.Structure):
    _fields_ = ([(k, C.c_int64) for k in ("m", "n", "n_c", "n_B", "n_N")] +
                [(k, C.POINTER(C.c_double)) for k in ("A", "c", "b", "lb", "ub", "x", "y", "d")] +
                [("bound_kind", C.POINTER(C.c_uint8)), ("N_bound", C.POINTER(C.c_uint8)),
                 ("B_index", C.POINTER(C.c_int64)), ("N_index", C.POINTER(C.c_int64))])


_host = None


def host_lib():
    """Loads libellp_host.so (which links libellp_hip.so). Raises if either is missing."""
    global _host
    if _host is not None:
        return _host
    _engine.lib()  # the engine must load first; raises with build instructions if missing
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError(f"{HOST_LIB_PATH} is missing: build it with `python -m ellp_amd.build`")
    L = C.CDLL(HOST_LIB_PATH)
    L.ellp_problem_new.restype = C.c_void_p
    L.ellp_problem_clone.restype = C.c_void_p
    L.ellp_problem_clone.argtypes = [C.c_void_p]
    L.ellp_problem_free.argtypes = [C.c_void_p]
    L.ellp_problem_add_var.restype = C.c_int64
    L.ellp_problem_add_var.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_char_p,
                                       C.c_char_p, C.c_size_t]
    L.ellp_problem_add_var_with_id.restype = C.c_int64
    L.ellp_problem_add_var_with_id.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int64,
                                               C.c_char_p, C.c_char_p, C.c_size_t]
    L.ellp_problem_add_constraint.restype = C.c_int
    L.ellp_problem_add_constraint.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                              C.c_char_p, C.c_size_t]
    L.ellp_problem_num_vars.restype = C.c_int64
    L.ellp_problem_num_vars.argtypes = [C.c_void_p]
    L.ellp_problem_num_constraints.restype = C.c_int64
    L.ellp_problem_num_constraints.argtypes = [C.c_void_p]
    L.ellp_problem_is_feasible.restype = C.c_int
    L.ellp_problem_is_feasible.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.ellp_parse_mps.restype = C.c_void_p
    L.ellp_parse_mps.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.ellp_solve.restype = C.c_int
    L.ellp_solve.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(Opts), C.POINTER(_Result)]
    L.ellp_result_free.argtypes = [C.POINTER(_Result)]
    L.ellp_debug_phase1.restype = C.c_int
    L.ellp_debug_phase1.argtypes = [C.c_void_p, C.c_int, C.POINTER(_FlatPhase), C.c_char_p, C.c_size_t]
    L.ellp_flat_phase_free.argtypes = [C.POINTER(_FlatPhase)]
    _host = L
    return L


class Problem:
    """src/problem.rs:11-154"""

    def __init__(self, _handle=None):
        self._h = _handle if _handle is not None else host_lib().ellp_problem_new()

    def __del__(self):
        try:
            if self._h:
                host_lib().ellp_problem_free(self._h)
                self._h = None
        except Exception:
            pass

    def clone(self):
        return Problem(host_lib().ellp_problem_clone(self._h))

    def add_var(self, obj_coeff, bound, name=None):
        err = C.create_string_buffer(512)
        vid = host_lib().ellp_problem_add_var(self._h, float(obj_coeff), bound.kind, bound.lb, bound.ub,
                                              None if name is None else name.encode(), err, 512)
        if vid < 0:
            raise EllPError(err.value.decode())
        return vid

    def add_var_with_id(self, obj_coeff, bound, var_id, name=None):
        err = C.create_string_buffer(512)
        vid = host_lib().ellp_problem_add_var_with_id(self._h, float(obj_coeff), bound.kind, bound.lb, bound.ub,
                                                      int(var_id), None if name is None else name.encode(), err, 512)
        if vid < 0:
            raise EllPError(err.value.decode())
        return vid

    def add_constraint(self, coeffs, op, rhs):
        ids = np.asarray([c[0] for c in coeffs], dtype=np.int64)
        cf = np.asarray([c[1] for c in coeffs], dtype=np.float64)
        if isinstance(op, str):
            op = ConstraintOp._BY_NAME[op]
        err = C.create_string_buffer(512)
        rc = host_lib().ellp_problem_add_constraint(self._h, len(ids), ids.ctypes.data_as(C.c_void_p),
                                                    cf.ctypes.data_as(C.c_void_p), int(op), float(rhs), err, 512)
        if rc != 0:
            raise EllPError(err.value.decode())

    def is_feasible(self, x):
        xv = np.ascontiguousarray(x, dtype=np.float64)
        return bool(host_lib().ellp_problem_is_feasible(self._h, xv.ctypes.data_as(C.c_void_p), xv.size))

    @property
    def num_vars(self):
        return host_lib().ellp_problem_num_vars(self._h)

    @property
    def num_constraints(self):
        return host_lib().ellp_problem_num_constraints(self._h)

    @staticmethod
    def from_fixture(fx):
        p = Problem()
        for k, (obj, bound) in enumerate(fx["vars"]):
            p.add_var(obj, Bound.from_fixture(bound), f"x{k + 1}")
        for coeffs, op, rhs in fx["constraints"]:
            p.add_constraint(coeffs, op, rhs)
        return p

    def _debug_phase1(self, solver):
        """Flattened phase-1 arrays (what solve() passes to the engine); None if infeasible by setup."""
        f = _FlatPhase()
        err = C.create_string_buffer(512)
        rc = host_lib().ellp_debug_phase1(self._h, 0 if solver == "primal" else 1, C.byref(f), err, 512)
        if rc == 1:
            return None
        if rc < 0:
            raise EllPError(err.value.decode())

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if (p and n) else np.zeros(0, dtype=dt)
        out = dict(m=f.m, n=f.n, n_c=f.n_c, A=arr(f.A, f.m * f.n, np.float64), c=arr(f.c, f.n_c, np.float64),
                   b=arr(f.b, f.m, np.float64), kind=arr(f.bound_kind, f.n_c, np.uint8),
                   lb=arr(f.lb, f.n_c, np.float64), ub=arr(f.ub, f.n_c, np.float64), x=arr(f.x, f.n_c, np.float64),
                   B=arr(f.B_index, f.n_B, np.int64), N=arr(f.N_index, f.n_N, np.int64),
                   Nb=arr(f.N_bound, f.n_N, np.uint8),
                   y=arr(f.y, f.m, np.float64) if f.y else None, d=arr(f.d, f.n_c, np.float64) if f.d else None)
        host_lib().ellp_flat_phase_free(C.byref(f))
        return out


class Solution:
    """src/solver.rs:35-54"""

    def __init__(self, obj, x):
        self._obj, self._x = obj, x

    def obj(self):
        return self._obj

    def x(self):
        return self._x


class SolverResult:
    """src/solver.rs:6-12: Optimal(Solution) | Infeasible | Unbounded | MaxIter{obj}"""
    Optimal, Infeasible, Unbounded, MaxIter = "optimal", "infeasible", "unbounded", "maxiter"

    def __init__(self, kind, solution=None, obj=None, iters=(0, 0)):
        self.kind, self.solution, self.obj, self.iters = kind, solution, obj, iters

    def __repr__(self):
        if self.kind == self.Optimal:
            return f"found optimal point with objective {self.solution.obj()}"
        if self.kind == self.MaxIter:
            return f"reached max iterations, current objective = {self.obj}"
        return f"problem is {self.kind}"


class _Solver:
    _KIND = 0

    def __init__(self, max_iter=1000, **engine_opts):
        """Default-constructed solvers have max_iter 1000 (primal…:19-23); `new(None)` is max_iter=None."""
        self.max_iter = MAX_ITER_NONE if max_iter is None else int(max_iter)
        self._opts = _engine.default_opts(**engine_opts) if engine_opts else None

    @classmethod
    def new(cls, max_iter=None, **engine_opts):
        return cls(max_iter, **engine_opts)

    @classmethod
    def default(cls):
        return cls()

    def solve(self, prob):
        r = _Result()
        host_lib().ellp_solve(prob._h, self._KIND, self.max_iter,
                              C.byref(self._opts) if self._opts is not None else None, C.byref(r))
        try:
            st, msg = r.status, r.err.decode()
            iters = (r.iters_phase1, r.iters_phase2)
            if st == _engine.OPTIMAL:
                x = np.ctypeslib.as_array(r.x, shape=(r.nx,)).copy() if r.nx else np.zeros(0)
                return SolverResult(SolverResult.Optimal, Solution(r.obj, x), iters=iters)
            if st == _engine.INFEASIBLE:
                return SolverResult(SolverResult.Infeasible, iters=iters)
            if st == _engine.UNBOUNDED:
                return SolverResult(SolverResult.Unbounded, iters=iters)
            if st == _engine.MAXITER:
                return SolverResult(SolverResult.MaxIter, obj=r.obj, iters=iters)
            if st in (_engine.ERR_BAD_DIMS, _engine.ERR_SINGULAR):
                raise EllPError(msg)
            if st == _engine.ERR_DEVICE:
                raise _engine.EllpHipError(st, msg)
            raise RuntimeError(f"panic: {msg}")
        finally:
            host_lib().ellp_result_free(C.byref(r))


class PrimalSimplexSolver(_Solver):
    """src/solvers/primal/primal_simplex_solver.rs:15-93"""
    _KIND = 0


class DualSimplexSolver(_Solver):
    """src/solvers/dual/dual_simplex_solver.rs:16-108"""
    _KIND = 1


def parse_mps(text):
    """src/parse_mps.rs:23"""
    err = C.create_string_buffer(512)
    h = host_lib().ellp_parse_mps(text.encode(), err, 512)
    if not h:
        raise MpsParsingError(err.value.decode())
    return Problem(h)
