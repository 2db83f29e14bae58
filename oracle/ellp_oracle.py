"""ctypes binding of the CPU oracle (oracle/ellp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under ellp_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libellp_oracle.so")

KIND = {"Free": 0, "Lower": 1, "Upper": 2, "TwoSided": 3, "Fixed": 4}
OP = {"Lte": 0, "Eq": 1, "Gte": 2}
NB_LOWER, NB_UPPER, NB_FREE = 0, 1, 2
OPTIMAL, INFEASIBLE, UNBOUNDED, MAXITER = 0, 1, 2, 3
NEED_EXACT = 4  # the explicit-inverse loops under set_binv_guard: suspicious pivot, nothing committed
ERR_BAD_DIMS, ERR_SINGULAR, ERR_NAN, ERR_ARG, ERR_PANIC = -1, -2, -3, -5, -6
MAX_ITER_NONE = 2**64 - 1
STATUS_NAME = {0: "optimal", 1: "infeasible", 2: "unbounded", 3: "maxiter"}


def build(force=False):
    """Compile libellp_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "ellp_oracle.c")
    hdr = os.path.join(_HERE, "ellp_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "libellp_oracle.so"])
    return _LIB_PATH


class _Phase(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("n_c", C.c_int64),
        ("A", C.POINTER(C.c_double)), ("c", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double)),
        ("kind", C.POINTER(C.c_uint8)),
        ("lb", C.POINTER(C.c_double)), ("ub", C.POINTER(C.c_double)),
        ("x", C.POINTER(C.c_double)),
        ("nB", C.c_int64), ("nN", C.c_int64),
        ("B", C.POINTER(C.c_int64)), ("N", C.POINTER(C.c_int64)),
        ("Nb", C.POINTER(C.c_uint8)),
        ("y", C.POINTER(C.c_double)), ("d", C.POINTER(C.c_double)),
        ("which", C.c_int),
        ("n_orig_vars", C.c_int64),
    ]


class _Result(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("obj", C.c_double), ("nx", C.c_int64),
        ("x", C.POINTER(C.c_double)), ("iters1", C.c_uint64), ("iters2", C.c_uint64),
        ("err", C.c_char * 256),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP threads sleep instead of spinning
    L = C.CDLL(_LIB_PATH)
    L.eo_problem_new.restype = C.c_void_p
    L.eo_problem_free.argtypes = [C.c_void_p]
    L.eo_add_var.restype = C.c_int64
    L.eo_add_var.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double]
    L.eo_add_var_with_id.restype = C.c_int64
    L.eo_add_var_with_id.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double,
                                     C.c_int64]
    L.eo_add_constraint.restype = C.c_int
    L.eo_add_constraint.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_double]
    L.eo_num_vars.restype = C.c_int64
    L.eo_num_vars.argtypes = [C.c_void_p]
    for name in ("eo_primal_phase1", "eo_dual_phase1"):
        f = getattr(L, name)
        f.restype = C.POINTER(_Phase)
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.eo_primal_phase2.restype = C.POINTER(_Phase)
    L.eo_primal_phase2.argtypes = [C.POINTER(_Phase)]
    L.eo_dual_phase2.restype = C.POINTER(_Phase)
    L.eo_dual_phase2.argtypes = [C.POINTER(_Phase), C.POINTER(C.c_int)]
    L.eo_phase_free.argtypes = [C.POINTER(_Phase)]
    L.eo_phase_obj.restype = C.c_double
    L.eo_phase_obj.argtypes = [C.POINTER(_Phase)]
    L.eo_phase_dual_obj.restype = C.c_double
    L.eo_phase_dual_obj.argtypes = [C.POINTER(_Phase)]
    common = [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
              C.c_int64]
    tail = [C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    L.eo_primal_solve_with_initial.restype = C.c_int
    L.eo_primal_solve_with_initial.argtypes = common + tail
    L.eo_dual_solve_with_initial.restype = C.c_int
    L.eo_dual_solve_with_initial.argtypes = common + [C.c_void_p, C.c_void_p] + tail
    L.eo_dual_binv_solve_with_initial.restype = C.c_int
    L.eo_dual_binv_solve_with_initial.argtypes = common + [C.c_void_p, C.c_void_p] + [
        C.c_uint64, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
    L.eo_primal_binv_solve_with_initial.restype = C.c_int
    L.eo_primal_binv_solve_with_initial.argtypes = common + [
        C.c_uint64, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
    hy_tail = [C.c_uint64, C.POINTER(C.c_uint64), C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
    L.eo_primal_hybrid_solve_with_initial.restype = C.c_int
    L.eo_primal_hybrid_solve_with_initial.argtypes = common + hy_tail
    L.eo_dual_hybrid_solve_with_initial.restype = C.c_int
    L.eo_dual_hybrid_solve_with_initial.argtypes = common + [C.c_void_p, C.c_void_p] + hy_tail
    L.eo_solve.restype = C.c_int
    L.eo_solve.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(_Result)]
    L.eo_result_free.argtypes = [C.POINTER(_Result)]
    L.eo_set_dense_lu.argtypes = [C.c_int]
    L.eo_set_setup_threads.argtypes = [C.c_int]
    L.eo_set_partial_segments.argtypes = [C.c_int]
    L.eo_set_dual_rule.argtypes = [C.c_int]
    L.eo_set_primal_rule.argtypes = [C.c_int]
    L.eo_set_binv_guard.argtypes = [C.c_double, C.c_double]
    L.eo_set_continuation.argtypes = [C.c_int]
    L.eo_set_binv_zero_tol.argtypes = [C.c_double]
    L.eo_synth_dense_lp.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Problem:
    """Mirror of ellp::Problem for the oracle (problem.rs:12-154)."""

    def __init__(self):
        self._h = lib().eo_problem_new()

    def __del__(self):
        try:
            if self._h:
                lib().eo_problem_free(self._h)
                self._h = None
        except Exception:
            pass

    def add_var(self, obj, bound):
        kind, lb, ub = bound
        k = KIND[kind] if isinstance(kind, str) else int(kind)
        if k == KIND["Fixed"]:
            ub = lb
        vid = lib().eo_add_var(self._h, float(obj), k, float(lb), float(ub))
        if vid < 0:
            raise ValueError("invalid variable")
        return vid

    def add_constraint(self, coeffs, op, rhs):
        ids = np.asarray([c[0] for c in coeffs], dtype=np.int64)
        cf = np.asarray([c[1] for c in coeffs], dtype=np.float64)
        o = OP[op] if isinstance(op, str) else int(op)
        rc = lib().eo_add_constraint(self._h, len(coeffs), _ptr(ids), _ptr(cf), o, float(rhs))
        if rc != 0:
            raise ValueError("invalid variable id in constraint")

    def add_dense_constraints(self, A, op, b):
        """rows of a dense matrix (used by the synthetic family)."""
        A = np.asarray(A, dtype=np.float64)
        ids = np.arange(A.shape[1], dtype=np.int64)
        o = OP[op] if isinstance(op, str) else int(op)
        for i in range(A.shape[0]):
            row = np.ascontiguousarray(A[i])
            lib().eo_add_constraint(self._h, A.shape[1], _ptr(ids), _ptr(row), o, float(b[i]))

    @property
    def num_vars(self):
        return lib().eo_num_vars(self._h)

    @staticmethod
    def from_fixture(fx):
        p = Problem()
        for obj, bound in fx["vars"]:
            p.add_var(obj, bound)
        for coeffs, op, rhs in fx["constraints"]:
            p.add_constraint(coeffs, op, rhs)
        return p


class Phase:
    """Flat (C-ABI shaped) view of a standardized problem + feasible point (copies)."""

    FIELDS = ("A", "c", "b", "kind", "lb", "ub", "x", "B", "N", "Nb", "y", "d")

    def __init__(self, ph):
        s = ph.contents
        self.m, self.n, self.n_c = s.m, s.n, s.n_c
        self.nB, self.nN = s.nB, s.nN
        self.which = s.which
        self.n_orig_vars = s.n_orig_vars

        def arr(p, n, dt):
            if not p or n == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True)

        self.A = arr(s.A, s.m * s.n, np.float64)  # column-major, ld = m
        self.c = arr(s.c, s.n_c, np.float64)
        self.b = arr(s.b, s.m, np.float64)
        self.kind = arr(s.kind, s.n_c, np.uint8)
        self.lb = arr(s.lb, s.n_c, np.float64)
        self.ub = arr(s.ub, s.n_c, np.float64)
        self.x = arr(s.x, s.n_c, np.float64)
        self.B = arr(s.B, s.nB, np.int64)
        # N gets head-room: the m == 0 trivial path rewrites it with up to n_c entries
        self.N = np.zeros(max(s.nN, s.n_c), dtype=np.int64)
        self.N[:s.nN] = arr(s.N, s.nN, np.int64)
        self.Nb = np.zeros(max(s.nN, s.n_c), dtype=np.uint8)
        self.Nb[:s.nN] = arr(s.Nb, s.nN, np.uint8)
        self.y = arr(s.y, s.m, np.float64) if s.y else None
        self.d = arr(s.d, s.n_c, np.float64) if s.d else None

    def copy(self):
        import copy
        return copy.deepcopy(self)

    def A_matrix(self):
        return self.A.reshape((self.n, self.m)).T if self.m * self.n else np.zeros((self.m, self.n))

    def obj(self):
        return float(np.dot(self.c, self.x))


class PhaseHandle:
    """Owns an eo_phase*."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __bool__(self):
        return bool(self.ptr)

    def view(self):
        return Phase(self.ptr)

    def obj(self):
        return lib().eo_phase_obj(self.ptr)

    def store_point(self, view):
        """Write x/B/N/Nb (/y/d) of a Phase view back into the C struct (same sizes)."""
        s = self.ptr.contents
        assert view.nB == s.nB and view.n_c == s.n_c
        C.memmove(s.x, view.x.ctypes.data, 8 * s.n_c)
        C.memmove(s.B, view.B.ctypes.data, 8 * s.nB)
        nN = view.nN
        C.memmove(s.N, view.N.ctypes.data, 8 * nN)
        C.memmove(s.Nb, view.Nb.ctypes.data, nN)
        self.ptr.contents.nN = nN
        if s.y and view.y is not None and s.m:
            C.memmove(s.y, view.y.ctypes.data, 8 * s.m)
        if s.d and view.d is not None:
            C.memmove(s.d, view.d.ctypes.data, 8 * s.n_c)

    def dual_obj(self):
        return lib().eo_phase_dual_obj(self.ptr)

    def __del__(self):
        try:
            if self.ptr:
                lib().eo_phase_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def primal_phase1(prob):
    err = C.c_int(0)
    p = lib().eo_primal_phase1(prob._h, C.byref(err))
    return (PhaseHandle(p) if p else None), err.value


def primal_phase2(ph1):
    return PhaseHandle(lib().eo_primal_phase2(ph1.ptr))


def dual_phase1(prob):
    err = C.c_int(0)
    p = lib().eo_dual_phase1(prob._h, C.byref(err))
    return (PhaseHandle(p) if p else None), err.value


def dual_phase2(ph1):
    err = C.c_int(0)
    p = lib().eo_dual_phase2(ph1.ptr, C.byref(err))
    return (PhaseHandle(p) if p else None), err.value


def primal_solve_with_initial(ph, max_iter=MAX_ITER_NONE):
    """Runs the oracle's primal loop on a Phase view IN PLACE. Returns (status, iters, err)."""
    it = C.c_uint64(0)
    err = C.create_string_buffer(256)
    st = lib().eo_primal_solve_with_initial(
        ph.m, ph.n, ph.n_c, _ptr(ph.A), _ptr(ph.c), _ptr(ph.b), _ptr(ph.kind), _ptr(ph.lb),
        _ptr(ph.ub), _ptr(ph.x), _ptr(ph.B), ph.nB, _ptr(ph.N), _ptr(ph.Nb), ph.nN,
        max_iter, C.byref(it), err, 256)
    return st, it.value, err.value.decode()


def host_threads():
    """Threads for the OpenMP loops: the affinity mask, capped at 16 — a one-GPU box exposes all
    of the host's CPUs (256) but grants a share of 16, and spinning OpenMP threads beyond the
    share are far slower than none."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


def primal_binv_solve_with_initial(ph, max_iter=MAX_ITER_NONE, threads=None, refresh=0):
    """The primal loop with the SAME pivoting rules but an explicitly maintained B^-1 (eta updates)
    on `threads` OpenMP threads — the engine's algorithm on the host cores.  In place on a Phase
    view.  Returns (status, iters, err, loop_seconds)."""
    if threads is None:
        threads = host_threads()
    it = C.c_uint64(0)
    secs = C.c_double(0.0)
    err = C.create_string_buffer(256)
    st = lib().eo_primal_binv_solve_with_initial(
        ph.m, ph.n, ph.n_c, _ptr(ph.A), _ptr(ph.c), _ptr(ph.b), _ptr(ph.kind), _ptr(ph.lb),
        _ptr(ph.ub), _ptr(ph.x), _ptr(ph.B), ph.nB, _ptr(ph.N), _ptr(ph.Nb), ph.nN,
        max_iter, C.byref(it), int(threads), int(refresh), C.byref(secs), err, 256)
    return st, it.value, err.value.decode(), secs.value


def dual_binv_solve_with_initial(ph, max_iter=MAX_ITER_NONE, threads=None, refresh=0):
    """The dual loop with the same pivoting rules and an explicitly maintained B^-1 on OpenMP threads.
    Returns (status, iters, err, loop_seconds)."""
    if threads is None:
        threads = host_threads()
    it = C.c_uint64(0)
    secs = C.c_double(0.0)
    err = C.create_string_buffer(256)
    st = lib().eo_dual_binv_solve_with_initial(
        ph.m, ph.n, ph.n_c, _ptr(ph.A), _ptr(ph.c), _ptr(ph.b), _ptr(ph.kind), _ptr(ph.lb),
        _ptr(ph.ub), _ptr(ph.x), _ptr(ph.B), ph.nB, _ptr(ph.N), _ptr(ph.Nb), ph.nN,
        _ptr(ph.y), _ptr(ph.d), max_iter, C.byref(it), int(threads), int(refresh), C.byref(secs), err, 256)
    return st, it.value, err.value.decode(), secs.value


def dual_solve_with_initial(ph, max_iter=MAX_ITER_NONE):
    it = C.c_uint64(0)
    err = C.create_string_buffer(256)
    st = lib().eo_dual_solve_with_initial(
        ph.m, ph.n, ph.n_c, _ptr(ph.A), _ptr(ph.c), _ptr(ph.b), _ptr(ph.kind), _ptr(ph.lb),
        _ptr(ph.ub), _ptr(ph.x), _ptr(ph.B), ph.nB, _ptr(ph.N), _ptr(ph.Nb), ph.nN,
        _ptr(ph.y), _ptr(ph.d), max_iter, C.byref(it), err, 256)
    return st, it.value, err.value.decode()


HYBRID_K = 8            # exact iterations per hand-over
HYBRID_GUARD_ABS = 1e-7  # a pivot below this is not taken by the explicit-inverse loop


def primal_hybrid_solve_with_initial(ph, max_iter=MAX_ITER_NONE, K=HYBRID_K, guard_abs=HYBRID_GUARD_ABS, refresh=64, threads=1):
    """The certified hybrid (the engine's default policy above 128 rows, restated in ellp_oracle.c), primal, in place on a
    Phase view.  Returns (status, iters, err, counters) with counters = [guard hand-overs, terminal statuses examined,
    not confirmed, exact iterations, solves repeated by the exact loop]."""
    it = C.c_uint64(0)
    err = C.create_string_buffer(256)
    cnt = np.zeros(5, dtype=np.uint64)
    st = lib().eo_primal_hybrid_solve_with_initial(
        ph.m, ph.n, ph.n_c, _ptr(ph.A), _ptr(ph.c), _ptr(ph.b), _ptr(ph.kind), _ptr(ph.lb),
        _ptr(ph.ub), _ptr(ph.x), _ptr(ph.B), ph.nB, _ptr(ph.N), _ptr(ph.Nb), ph.nN,
        max_iter, C.byref(it), int(K), float(guard_abs), int(refresh), int(threads), _ptr(cnt), err, 256)
    return st, it.value, err.value.decode(), [int(v) for v in cnt]


def dual_hybrid_solve_with_initial(ph, max_iter=MAX_ITER_NONE, K=HYBRID_K, guard_abs=HYBRID_GUARD_ABS, refresh=64, threads=1):
    """The certified hybrid, dual.  Returns (status, iters, err, counters)."""
    it = C.c_uint64(0)
    err = C.create_string_buffer(256)
    cnt = np.zeros(5, dtype=np.uint64)
    st = lib().eo_dual_hybrid_solve_with_initial(
        ph.m, ph.n, ph.n_c, _ptr(ph.A), _ptr(ph.c), _ptr(ph.b), _ptr(ph.kind), _ptr(ph.lb),
        _ptr(ph.ub), _ptr(ph.x), _ptr(ph.B), ph.nB, _ptr(ph.N), _ptr(ph.Nb), ph.nN,
        _ptr(ph.y), _ptr(ph.d), max_iter, C.byref(it), int(K), float(guard_abs), int(refresh), int(threads), _ptr(cnt), err, 256)
    return st, it.value, err.value.decode(), [int(v) for v in cnt]


class Result:
    def __init__(self, r):
        self.status = r.status
        self.obj = r.obj
        self.x = (np.ctypeslib.as_array(r.x, shape=(r.nx,)).copy()
                  if (r.status == OPTIMAL and r.x and r.nx > 0) else np.zeros(0))
        self.iters = (r.iters1, r.iters2)
        self.err = r.err.decode()

    def __repr__(self):
        return f"Result(status={self.status}, obj={self.obj}, x={self.x}, iters={self.iters})"


def solve(prob, solver="primal", max_iter=1000):
    """PrimalSimplexSolver / DualSimplexSolver ::solve; default max_iter 1000 as Default."""
    r = _Result()
    lib().eo_solve(prob._h, 0 if solver == "primal" else 1,
                   MAX_ITER_NONE if max_iter is None else int(max_iter), C.byref(r))
    out = Result(r)
    lib().eo_result_free(C.byref(r))
    return out


def set_setup_threads(n):
    """threads for the setup factorizations (QR of the rank check, LU of A^T): bitwise the one-thread
    results (columns are shared out, each processed sequentially); 1 restores the default"""
    lib().eo_set_setup_threads(int(n))


def set_dense_lu(on):
    """Timed-baseline mode: no zero-multiplier skip in LU (nalgebra does the full dense work)."""
    lib().eo_set_dense_lu(1 if on else 0)


def synth_dense_lp(seed, m, n):
    """SURVEY §8d family: returns (A (m x n, Fortran order), b, c)."""
    A = np.zeros((m, n), dtype=np.float64, order="F")
    b = np.zeros(m)
    c = np.zeros(n)
    lib().eo_synth_dense_lp(int(seed), m, n, _ptr(A), _ptr(b), _ptr(c))
    return A, b, c


def set_dual_rule(bits):
    """dual extensions in dual_solve_with_initial (not the reference's rules; see ellp_oracle.c): bit 0 bound-flipping
    ratio test, bit 1 leaving row of largest violation; 0 restores the reference's rules"""
    lib().eo_set_dual_rule(int(bits))


def set_primal_rule(rule):
    """primal extension in primal_solve_with_initial (not the reference's rule; see ellp_oracle.c): 1 steepest-edge pricing"""
    lib().eo_set_primal_rule(int(rule))


def set_binv_guard(rel, abs_=0.0):
    """pivot guard of the certified hybrid in the explicit-inverse loops (an extension, see ellp_oracle.c): they stop with
    NEED_EXACT before committing an iteration whose pivot is below abs_ or below rel * max|B^-1 a_q|; 0, 0 = off"""
    lib().eo_set_binv_guard(float(rel), float(abs_))


def set_binv_zero_tol(t):
    """the explicit-inverse loops take entries of B^-1 a_q and of the dual pricing row below t to be zero (certified hybrid)"""
    lib().eo_set_binv_zero_tol(float(t))


def set_continuation(on):
    """the dual loops skip their entry assertion (dual…:139-151): the call continues a solve another loop began"""
    lib().eo_set_continuation(1 if on else 0)


def set_partial_segments(P):
    """partial pricing in primal_solve_with_initial (an extension, see ellp_oracle.c); P <= 1 turns it off"""
    lib().eo_set_partial_segments(int(P))


def synth_problem(seed, m, n):
    A, b, c = synth_dense_lp(seed, m, n)
    p = Problem()
    for j in range(n):
        p.add_var(c[j], ("Lower", 0.0, 0.0))
    p.add_dense_constraints(A, "Lte", b)
    return p
