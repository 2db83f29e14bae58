"""ctypes binding of the C ABI in include/ellp_hip.h (libellp_hip.so).

The library is the product's compute path; there is no Python or CPU substitute.  If it is
missing or cannot be loaded this module raises — it never falls back.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ELLP_HIP_LIB") or os.path.join(_HERE, "libellp_hip.so")  # override: dev builds only

OPTIMAL, INFEASIBLE, UNBOUNDED, MAXITER = 0, 1, 2, 3
ERR_BAD_DIMS, ERR_SINGULAR, ERR_NAN, ERR_DEVICE, ERR_ARG, ERR_PANIC = -1, -2, -3, -4, -5, -6
STATUS_NAME = {0: "optimal", 1: "infeasible", 2: "unbounded", 3: "maxiter", -1: "err_bad_dims",
               -2: "err_singular", -3: "err_nan", -4: "err_device", -5: "err_arg", -6: "err_panic"}
MAX_ITER_NONE = 2**64 - 1
ENGINE_PRIMAL, ENGINE_DUAL = 0, 1
# ellp_opts.flags (include/ellp_hip.h)
FLAG_DENSE_PRICING, FLAG_DUAL_MAX_VIOLATION, FLAG_PRIMAL_STEEPEST_EDGE, FLAG_NO_CERTIFY, FLAG_DUAL_BOUND_FLIPPING = 1, 2, 4, 8, 16
K_NAMES = ["price", "select", "ftran", "ratio", "update", "btran", "refactor", "dleave", "dprice",
           "dselect", "dupdate", "event_cost"]
K_COUNT = 12
TAP_U, TAP_R, TAP_D, TAP_BINV, TAP_KEY, TAP_ALPHA, TAP_STATE = range(7)


class Opts(C.Structure):
    _fields_ = [("max_iter", C.c_uint64), ("eps", C.c_double), ("device", C.c_int32),
                ("refactor_period", C.c_int32), ("btran_mode", C.c_int32),
                ("poll_interval", C.c_int32), ("profile", C.c_int32), ("use_graph", C.c_int32),
                ("pipeline", C.c_int32), ("trace_len", C.c_int32), ("partial_segments", C.c_int32),
                ("flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("iters", C.c_uint64), ("pivots", C.c_uint64), ("bound_flips", C.c_uint64),
                ("refactors", C.c_uint64), ("obj", C.c_double), ("t_loop_s", C.c_double),
                ("t_setup_s", C.c_double), ("kernel_ms", C.c_double * K_COUNT),
                ("kernel_calls", C.c_uint64 * K_COUNT)]

    def as_dict(self):
        d = {k: getattr(self, k) for k in ("iters", "pivots", "bound_flips", "refactors", "obj",
                                           "t_loop_s", "t_setup_s")}
        d["kernel_ms"] = {K_NAMES[i]: self.kernel_ms[i] for i in range(K_COUNT) if self.kernel_calls[i]}
        d["kernel_calls"] = {K_NAMES[i]: self.kernel_calls[i] for i in range(K_COUNT) if self.kernel_calls[i]}
        return d


# int fn(void *user, void *host_buffer, int64_t segment_bytes, int world)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)


class EllpHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAME.get(status, status)}: {msg}")
        self.status = status
        self.msg = msg


_lib = None
_PROBLEM_ARGS = [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                 C.c_int64]


def lib():
    """Loads libellp_hip.so; raises if it is missing (build it with `python -m ellp_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP engine is the only compute path of ellp_amd. "
            "Build it with `python -m ellp_amd.build` (hipcc --offload-arch=gfx950).")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.ellp_default_opts.argtypes = [C.POINTER(Opts)]
    L.ellp_hip_device_count.restype = C.c_int
    L.ellp_hip_abi_version.restype = C.c_int
    tail = [C.POINTER(Opts), C.POINTER(Stats), C.c_char_p, C.c_size_t]
    L.ellp_primal_solve_with_initial.restype = C.c_int
    L.ellp_primal_solve_with_initial.argtypes = _PROBLEM_ARGS + tail
    L.ellp_dual_solve_with_initial.restype = C.c_int
    L.ellp_dual_solve_with_initial.argtypes = _PROBLEM_ARGS + [C.c_void_p, C.c_void_p] + tail
    L.ellp_engine_create.restype = C.c_int
    L.ellp_engine_create.argtypes = ([C.c_int] + _PROBLEM_ARGS + [C.c_void_p, C.c_void_p] +
                                     [C.POINTER(Opts), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t])
    L.ellp_engine_run.restype = C.c_int
    L.ellp_engine_run.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Stats), C.c_char_p, C.c_size_t]
    L.ellp_engine_read_point.restype = C.c_int
    L.ellp_engine_read_point.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_char_p, C.c_size_t]
    L.ellp_engine_tap.restype = C.c_int64
    L.ellp_engine_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.ellp_engine_refactor.restype = C.c_int
    L.ellp_engine_refactor.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.ellp_engine_refresh.restype = C.c_double
    L.ellp_engine_refresh.argtypes = [C.c_void_p]
    L.ellp_engine_inverse_residual.restype = C.c_double
    L.ellp_engine_inverse_residual.argtypes = [C.c_void_p]
    L.ellp_engine_destroy.argtypes = [C.c_void_p]
    L.ellp_engine_read_trace.restype = C.c_int64
    L.ellp_engine_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.ellp_engine_request_maintenance.restype = C.c_int
    L.ellp_engine_request_maintenance.argtypes = [C.c_void_p]
    L.ellp_engine_debug_scale_inverse.restype = C.c_int
    L.ellp_engine_debug_scale_inverse.argtypes = [C.c_void_p, C.c_double]
    L.ellp_engine_set_shard.restype = C.c_int
    L.ellp_engine_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
    L.ellp_engine_segment_doubles.restype = C.c_int64
    L.ellp_engine_segment_doubles.argtypes = [C.c_void_p, C.c_int]
    L.ellp_engine_exchange_info.restype = C.c_int
    L.ellp_engine_exchange_info.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ellp_engine_set_stream.restype = C.c_int
    L.ellp_engine_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.ellp_engine_step.restype = C.c_int
    L.ellp_engine_step.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.ellp_engine_poll.restype = C.c_int
    L.ellp_engine_poll.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_char_p, C.c_size_t]
    L.ellp_engine_rephase.restype = C.c_int
    L.ellp_engine_rephase.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.ellp_engine_create_primal_phase1.restype = C.c_int
    L.ellp_engine_create_primal_phase1.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 7 + [
        C.POINTER(Opts), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.ellp_engine_create_dual_phase1.restype = C.c_int
    L.ellp_engine_create_dual_phase1.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 8 + [
        C.POINTER(Opts), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.ellp_engine_dual_rephase.restype = C.c_int
    L.ellp_engine_dual_rephase.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_char_p, C.c_size_t]
    L.ellp_hip_qr_transposed.restype = C.c_int
    L.ellp_hip_qr_transposed.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.ellp_comm_unique_id.restype = C.c_int
    L.ellp_comm_unique_id.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.ellp_engine_comm_init.restype = C.c_int
    L.ellp_engine_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.ellp_engine_run_sharded.restype = C.c_int
    L.ellp_engine_run_sharded.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Stats), C.c_char_p, C.c_size_t]
    L.ellp_engine_shard_columns.restype = C.c_int
    L.ellp_engine_shard_columns.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.ellp_engine_set_exchange_callback.restype = C.c_int
    L.ellp_engine_set_exchange_callback.argtypes = [C.c_void_p, EXCHANGE_FN, C.c_void_p]
    L.ellp_engine_mailbox_export.restype = C.c_int
    L.ellp_engine_mailbox_export.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.ellp_engine_mailbox_connect.restype = C.c_int
    L.ellp_engine_mailbox_connect.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.ellp_engine_mailbox_selftest.restype = C.c_int
    L.ellp_engine_mailbox_selftest.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.ellp_shard_select_compact.restype = C.c_int
    L.ellp_shard_select_compact.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ellp_shard_pack_doubles.restype = C.c_int64
    L.ellp_shard_pack_doubles.argtypes = [C.c_int64]
    L.ellp_hip_lu_transposed.restype = C.c_int
    L.ellp_hip_lu_transposed.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.ellp_engine_shard_info.restype = C.c_int
    L.ellp_engine_shard_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    _lib = L
    return L


def qr_transposed(A, device=-1):
    """Column-pivoted Householder QR of A^T on the device (standard_form.rs:142).  A: (m, nv) array.
    Returns (pivots, |R_ii|), both of length min(m, nv)."""
    A = np.asfortranarray(A, dtype=np.float64)
    m, nv = A.shape
    mn = min(m, nv)
    piv = np.zeros(mn, dtype=np.int64)
    rd = np.zeros(mn, dtype=np.float64)
    err = C.create_string_buffer(512)
    s = lib().ellp_hip_qr_transposed(m, nv, A.ctypes.data_as(C.c_void_p), _p(piv), _p(rd), int(device), err, 512)
    if s != OPTIMAL:
        raise EllpHipError(s, err.value.decode())
    return piv, rd


def lu_transposed(A, device=-1):
    """LU with partial pivoting of A^T on the device (dual_problem.rs:141).  A: (m, nv) array, nv >= m.
    Returns (pivots, U_ii), both of length m."""
    A = np.asfortranarray(A, dtype=np.float64)
    m, nv = A.shape
    piv = np.zeros(m, dtype=np.int64)
    ud = np.zeros(m, dtype=np.float64)
    err = C.create_string_buffer(512)
    s = lib().ellp_hip_lu_transposed(m, nv, A.ctypes.data_as(C.c_void_p), _p(piv), _p(ud), int(device), err, 512)
    if s != OPTIMAL:
        raise EllpHipError(s, err.value.decode())
    return piv, ud


def shard_pack_doubles(ld):
    return int(lib().ellp_shard_pack_doubles(int(ld)))


def shard_select_compact(packs, world, ld, eps=1e-10):
    """The ranks' selection on gathered packs (host build of the kernel's code).  Returns (verdict, q,
    src_rank, src_slot): verdict 1 = the full exchange is needed."""
    packs = np.ascontiguousarray(packs, dtype=np.float64)
    q, sr, sc = C.c_int64(-1), C.c_int(-1), C.c_int(0)
    v = lib().ellp_shard_select_compact(_p(packs), int(world), int(ld), float(eps), C.byref(q), C.byref(sr), C.byref(sc))
    return int(v), int(q.value), int(sr.value), int(sc.value)


def comm_unique_id(rccl_path=None):
    """128-byte RCCL unique id (rank 0 creates it, the other ranks receive it)."""
    buf = C.create_string_buffer(128)
    err = C.create_string_buffer(512)
    path = rccl_path.encode() if rccl_path else None
    s = lib().ellp_comm_unique_id(path, buf, err, 512)
    if s != OPTIMAL:
        raise EllpHipError(s, err.value.decode())
    return buf.raw


def default_opts(**kw):
    o = Opts()
    lib().ellp_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "max_iter" and v is None:
            v = MAX_ITER_NONE
        setattr(o, k, v)
    return o


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class FlatProblem:
    """The arrays solve_with_initial has in hand (standard_form.rs:21-34), C-ABI shaped.
    A: (m*n,) column-major.  x/B/N/Nb (and y/d) are mutated in place by the solve calls."""

    def __init__(self, m, n, n_c, A, c, b, kind, lb, ub, x, B, N, Nb, y=None, d=None, nN=None):
        self.m, self.n, self.n_c = int(m), int(n), int(n_c)
        self.A, self.c, self.b = _f64(A).reshape(-1), _f64(c), _f64(b)
        self.kind = np.ascontiguousarray(kind, dtype=np.uint8)
        self.lb, self.ub = _f64(lb), _f64(ub)
        self.x = _f64(x).copy()
        self.B = np.ascontiguousarray(B, dtype=np.int64).copy()
        self.N = np.ascontiguousarray(N, dtype=np.int64).copy()
        self.Nb = np.ascontiguousarray(Nb, dtype=np.uint8).copy()
        self.nB = len(self.B)
        self.nN = len(self.N) if nN is None else int(nN)
        self.y = None if y is None else _f64(y).copy()
        self.d = None if d is None else _f64(d).copy()

    def _args(self):
        return [self.m, self.n, self.n_c, _p(self.A), _p(self.c), _p(self.b), _p(self.kind),
                _p(self.lb), _p(self.ub), _p(self.x), _p(self.B), self.nB, _p(self.N), _p(self.Nb),
                self.nN]

    def obj(self):
        return float(np.dot(self.c, self.x))


def primal_solve_with_initial(fp, opts=None):
    """PrimalSimplexSolver::solve_with_initial on the GPU. Returns (status, Stats, errmsg)."""
    o = opts or default_opts()
    st = Stats()
    err = C.create_string_buffer(512)
    s = lib().ellp_primal_solve_with_initial(*fp._args(), C.byref(o), C.byref(st), err, 512)
    return s, st, err.value.decode()


def dual_solve_with_initial(fp, opts=None):
    o = opts or default_opts()
    st = Stats()
    err = C.create_string_buffer(512)
    s = lib().ellp_dual_solve_with_initial(*fp._args(), _p(fp.y), _p(fp.d), C.byref(o), C.byref(st),
                                           err, 512)
    return s, st, err.value.decode()


class Engine:
    """Resident engine: tableau stays in HBM between run() slices."""

    def __init__(self, kind, fp, opts=None):
        self.fp = fp
        self.kind = kind
        self._h = C.c_void_p()
        o = opts or default_opts()
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_create(kind, *fp._args(), _p(fp.y), _p(fp.d), C.byref(o),
                                     C.byref(self._h), err, 512)
        if s != OPTIMAL:
            self._h = C.c_void_p()
            raise EllpHipError(s, err.value.decode())

    @classmethod
    def primal_phase1(cls, m, n, A, b, kind, lb, ub, x, Nb, opts=None):
        """Primal phase 1 built on the device (ellp_engine_create_primal_phase1): the standard form and the
        nonbasic start of the n original variables in, an engine with n + m columns out.  Its FlatProblem holds
        the phase-1 arrays as the engine defines them (A is not materialised on the host: fp.A is empty)."""
        m, n = int(m), int(n)
        nt = n + m
        A_ = _f64(A).reshape(-1)
        assert A_.size == m * n
        kind_ = np.concatenate([np.ascontiguousarray(kind, dtype=np.uint8)[:n], np.full(m, 1, dtype=np.uint8)])
        fp = FlatProblem(m, nt, nt, np.zeros(0), np.concatenate([np.zeros(n), np.ones(m)]), b, kind_,
                         np.concatenate([_f64(lb)[:n], np.zeros(m)]), np.concatenate([_f64(ub)[:n], np.zeros(m)]),
                         np.concatenate([_f64(x)[:n], np.zeros(m)]), np.arange(n, nt, dtype=np.int64),
                         np.arange(n, dtype=np.int64), np.ascontiguousarray(Nb, dtype=np.uint8)[:n])
        self = cls.__new__(cls)
        self.fp, self.kind, self._h = fp, ENGINE_PRIMAL, C.c_void_p()
        o = opts or default_opts()
        err = C.create_string_buffer(512)
        xs, Nbs = _f64(x)[:n].copy(), np.ascontiguousarray(Nb, dtype=np.uint8)[:n].copy()
        s = lib().ellp_engine_create_primal_phase1(m, n, _p(A_), _p(fp.b), _p(kind_), _p(fp.lb), _p(fp.ub), _p(xs), _p(Nbs),
                                                   C.byref(o), C.byref(self._h), err, 512)
        if s != OPTIMAL:
            self._h = C.c_void_p()
            raise EllpHipError(s, err.value.decode())
        return self

    @classmethod
    def dual_phase1(cls, m, n, A, c, b, kind, lb, ub, B, N, opts=None):
        """DualPhase1::new's point made on the device (ellp_engine_create_dual_phase1): the box problem's standard
        form and the basis the LU of A^T picked in, a dual engine with y, d, x and the nonbasic labels out
        (read_point brings them into self.fp)."""
        m, n = int(m), int(n)
        fp = FlatProblem(m, n, n, A, c, b, kind, lb, ub, np.zeros(n), B, N, np.zeros(n - m, dtype=np.uint8),
                         y=np.zeros(m), d=np.zeros(n))
        self = cls.__new__(cls)
        self.fp, self.kind, self._h = fp, ENGINE_DUAL, C.c_void_p()
        o = opts or default_opts()
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_create_dual_phase1(m, n, _p(fp.A), _p(fp.c), _p(fp.b), _p(fp.kind), _p(fp.lb), _p(fp.ub),
                                                 _p(fp.B), _p(fp.N), C.byref(o), C.byref(self._h), err, 512)
        if s != OPTIMAL:
            self._h = C.c_void_p()
            raise EllpHipError(s, err.value.decode())
        return self

    def run(self, max_iters):
        st = Stats()
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_run(self._h, int(max_iters), C.byref(st), err, 512)
        if s == ERR_DEVICE:
            raise EllpHipError(s, err.value.decode())
        return s, st, err.value.decode()

    def read_point(self):
        fp = self.fp
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_read_point(self._h, _p(fp.x), _p(fp.B), _p(fp.N), _p(fp.Nb), _p(fp.y),
                                         _p(fp.d), err, 512)
        # OPTIMAL: delivered.  UNBOUNDED / MAXITER: delivered, and completing the iteration the two-launch pipeline had left
        # open ended the solve that way (include/ellp_hip.h) — kept in closing_status.  Every error code raises (the arrays
        # are filled all the same: the caller may look at fp after catching): a point that came with ERR_NAN / ERR_PANIC /
        # ERR_SINGULAR or any code this binding does not know is never returned as a success.
        self.closing_status = s
        if s not in (OPTIMAL, UNBOUNDED, MAXITER):
            raise EllpHipError(s, err.value.decode())
        return fp

    def tap(self, what, count):
        out = np.zeros(int(count), dtype=np.float64)
        n = lib().ellp_engine_tap(self._h, what, _p(out), out.size)
        if n < 0:
            raise EllpHipError(int(n), "tap failed")
        return out[:n]

    def refactor(self):
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_refactor(self._h, err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def refresh(self):
        """One Newton-Schulz step on B^-1; returns the residual before the step."""
        return lib().ellp_engine_refresh(self._h)

    def inverse_residual(self):
        return lib().ellp_engine_inverse_residual(self._h)

    def read_trace(self, cap=1 << 16):
        """(iterations, objectives) of the ring buffer kept with opts.trace_len > 0, oldest first"""
        it = np.zeros(int(cap), dtype=np.uint64)
        ob = np.zeros(int(cap), dtype=np.float64)
        n = lib().ellp_engine_read_trace(self._h, _p(it), _p(ob), int(cap))
        if n < 0:
            raise EllpHipError(int(n), "read_trace failed")
        return it[:n], ob[:n]

    def request_maintenance(self):
        """test hook: what a kernel does after a tiny pivot (the next run()/poll() services it)"""
        s = lib().ellp_engine_request_maintenance(self._h)
        if s != OPTIMAL:
            raise EllpHipError(s, "request_maintenance failed")

    def debug_scale_inverse(self, factor):
        """test hook: B^-1 <- factor * B^-1"""
        s = lib().ellp_engine_debug_scale_inverse(self._h, float(factor))
        if s != OPTIMAL:
            raise EllpHipError(s, "debug_scale_inverse failed")

    def counters(self):
        """host-side maintenance counters (TAP_STATE tail)"""
        v = self.tap(TAP_STATE, 30)
        return dict(drift=v[12], drift_checks=int(v[13]), maint_requests=int(v[14]), refreshes=int(v[15]),
                    rebuilds=int(v[16]), resyncs=int(v[17]), last_refresh_residual=v[18], launches_per_iteration=int(v[19]),
                    rebuild_shortcuts=int(v[20]), t_setup_s=float(v[21]),
                    # certified hybrid: is it on, guarded pivots handed to the exact kernel, terminal statuses
                    # examined, of those not confirmed, loop bodies run by the exact kernel, rebuilds after a hand-over
                    hybrid=(int(v[22]) == 1), certified_by_exact_lu_iteration=(int(v[22]) == 2), hybrid_guards=int(v[23]), hybrid_certs=int(v[24]), hybrid_disagreed=int(v[25]),
                    hybrid_exact_iters=int(v[26]), hybrid_rebuilds=int(v[27]), hybrid_redos=int(v[28]),
                    # end points returned NOT certified (a redo above 1,024 rows ruled out by ELLP_REDO_MAX_SECONDS, a singular LU)
                    hybrid_uncertified=int(v[29]))

    # ---- sharded / stepped driving (see ellp_amd/dist.py)
    def segment_doubles(self, world):
        return int(lib().ellp_engine_segment_doubles(self._h, int(world)))

    def set_shard(self, rank, world, buffer_ptr=None):
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_set_shard(self._h, int(rank), int(world), C.c_void_p(buffer_ptr), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def exchange_info(self):
        base, seg, rank, world = C.c_void_p(), C.c_int64(), C.c_int(), C.c_int()
        lib().ellp_engine_exchange_info(self._h, C.byref(base), C.byref(seg), C.byref(rank), C.byref(world))
        return base.value, seg.value, rank.value, world.value

    def set_stream(self, stream_handle):
        lib().ellp_engine_set_stream(self._h, C.c_void_p(stream_handle))

    def step(self, phase):
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_step(self._h, int(phase), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def poll(self):
        st = Stats()
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_poll(self._h, C.byref(st), err, 512)
        if s == ERR_DEVICE:
            raise EllpHipError(s, err.value.decode())
        return s, st, err.value.decode()

    def rephase(self, c, kind, lb, ub):
        """Primal phase-1 -> phase-2 hand-off on the device (costs and bounds replaced; basis, point
        and B^-1 stay).  The FlatProblem's c/kind/lb/ub are updated too."""
        fp = self.fp
        fp.c = _f64(c)
        fp.kind = np.ascontiguousarray(kind, dtype=np.uint8)
        fp.lb, fp.ub = _f64(lb), _f64(ub)
        assert fp.c.size == fp.n_c == fp.kind.size == fp.lb.size == fp.ub.size
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_rephase(self._h, _p(fp.c), _p(fp.kind), _p(fp.lb), _p(fp.ub), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def dual_rephase(self, c, b, kind, lb, ub):
        """Dual phase-1 -> phase-2 hand-off on the device (same matrix in both phases).  The FlatProblem's
        c/b/kind/lb/ub are replaced; read_point() afterwards brings x, N (in variable order), y, d back."""
        fp = self.fp
        fp.c, fp.b = _f64(c), _f64(b)
        fp.kind = np.ascontiguousarray(kind, dtype=np.uint8)
        fp.lb, fp.ub = _f64(lb), _f64(ub)
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_dual_rephase(self._h, _p(fp.c), _p(fp.b), _p(fp.kind), _p(fp.lb), _p(fp.ub), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    # ---- sharded loop inside the library, exchange by RCCL on the engine's stream
    def comm_init(self, unique_id, rank, world, rccl_path=None):
        err = C.create_string_buffer(512)
        path = rccl_path.encode() if rccl_path else None
        s = lib().ellp_engine_comm_init(self._h, path, bytes(unique_id), int(rank), int(world), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def run_sharded(self, max_iters):
        st = Stats()
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_run_sharded(self._h, C.c_uint64(int(max_iters)), C.byref(st), err, 512)
        if s == ERR_DEVICE or s == ERR_ARG:
            raise EllpHipError(s, err.value.decode())
        return s, st, err.value.decode()

    # ---- column-sharded storage (include/ellp_hip.h, ellp_engine_shard_columns)
    def shard_columns(self, rank, world):
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_shard_columns(self._h, int(rank), int(world), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def set_exchange_callback(self, fn):
        """fn(host_buffer_address, segment_bytes, world) -> 0 on success; all-gathers in place"""
        def _cb(user, buf, seg, world):
            try:
                return int(fn(buf, seg, world))
            except Exception:  # never unwind into C
                import traceback
                traceback.print_exc()
                return 1
        self._xcb = EXCHANGE_FN(_cb)  # keep it alive as long as the engine
        lib().ellp_engine_set_exchange_callback(self._h, self._xcb, None)

    def mailbox_export(self):
        buf = C.create_string_buffer(128)
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_mailbox_export(self._h, buf, err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())
        return buf.raw

    def mailbox_connect(self, all_handles):
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_mailbox_connect(self._h, bytes(all_handles), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def mailbox_selftest(self, rounds=8):
        err = C.create_string_buffer(512)
        s = lib().ellp_engine_mailbox_selftest(self._h, int(rounds), err, 512)
        if s != OPTIMAL:
            raise EllpHipError(s, err.value.decode())

    def shard_info(self):
        out = (C.c_double * 6)()
        lib().ellp_engine_shard_info(self._h, out)
        return dict(full_exchanges=int(out[0]), column_requests=int(out[1]),
                    transport={0: "none", 1: "rccl", 2: "mailbox", 3: "callback"}[int(out[2])],
                    pack_doubles=int(out[3]), own=(int(out[4]), int(out[5])))

    def close(self):
        if self._h:
            lib().ellp_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
