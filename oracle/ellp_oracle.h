/*
 * ellp_oracle.h — CPU oracle for the ellp simplex hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm of
 * kehlert/ellp 0.2.0 (reference files cited per function in ellp_oracle.c).  It is
 * the checker for the HIP engine in ellp_amd/: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library
 * (libellp_hip.so / libellp_host.so) never links, loads or calls anything in oracle/.
 *
 * Parity pin: end-to-end only.  The reference's arithmetic lives in the un-vendored
 * crate nalgebra "^0" (Cargo.toml:16, no Cargo.lock); its published algorithms
 * (partial-pivot LU, column-oriented triangular solves, max-element column-pivoted
 * Householder QR, full-pivot LU) are restated here from their documented behaviour.
 * The oracle is pinned by every known answer the reference's tests hold
 * (tests/problems/mod.rs:130-674, README.md:88-106) for both solvers — see
 * tests/test_oracle_fixtures.py.  Per-step (bit-level) parity with nalgebra is
 * unpinned: no reference test pins it and no Rust toolchain exists here.
 */
#ifndef ELLP_ORACLE_H
#define ELLP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EO_EPS 1e-10 /* src/util.rs:1 */

/* problem.rs:191-197 */
enum { EO_FREE = 0, EO_LOWER = 1, EO_UPPER = 2, EO_TWOSIDED = 3, EO_FIXED = 4 };
/* standard_form.rs:206-210 */
enum { EO_NB_LOWER = 0, EO_NB_UPPER = 1, EO_NB_FREE = 2 };
/* problem.rs:299-303 */
enum { EO_LTE = 0, EO_EQ = 1, EO_GTE = 2 };
/* solver.rs:28-33 (+ error codes standing in for Err(EllPError) and panics) */
enum {
    EO_OPTIMAL = 0,
    EO_INFEASIBLE = 1,
    EO_UNBOUNDED = 2,
    EO_MAXITER = 3,
    EO_NEED_EXACT = 4,    /* eo_*_binv_* with eo_set_binv_guard: the pivot about to be used is suspicious (see ellp_oracle.c) */
    EO_ERR_BAD_DIMS = -1, /* EllPError "invalid B/N" */
    EO_ERR_SINGULAR = -2, /* EllPError "A_B is not invertible" */
    EO_ERR_NAN = -3,      /* panic "NaN detected" */
    EO_ERR_ARG = -5,
    EO_ERR_PANIC = -6     /* any other assert!/panic!/unwrap() of the reference */
};

#define EO_MAX_ITER_NONE UINT64_MAX /* Solver::new(None) */

typedef struct eo_problem eo_problem;

eo_problem *eo_problem_new(void);
void eo_problem_free(eo_problem *p);
/* Problem::add_var (problem.rs:24): returns the new id (= index) or -1 on validation error. */
int64_t eo_add_var(eo_problem *p, double obj_coeff, int kind, double lb, double ub);
/* Problem::add_var_with_id (problem.rs:35) */
int64_t eo_add_var_with_id(eo_problem *p, double obj_coeff, int kind, double lb, double ub,
                           int64_t id);
/* Problem::add_constraint (problem.rs:85): 0 ok, -1 unknown variable id. */
int eo_add_constraint(eo_problem *p, int64_t ncoef, const int64_t *ids, const double *coef,
                      int op, double rhs);
int64_t eo_num_vars(const eo_problem *p);
int64_t eo_num_constraints(const eo_problem *p);

/*
 * A "phase": StandardForm + feasible point, flattened exactly as the C ABI of the
 * HIP engine takes them (include/ellp_hip.h).  A is m x n column-major (ld = m);
 * c, x, kind/lb/ub have length n_c >= n (quirk Q5: primal phase 1 with free variables).
 */
typedef struct eo_phase {
    int64_t m, n, n_c;
    double *A, *c, *b;
    uint8_t *kind;
    double *lb, *ub;
    double *x;
    int64_t nB, nN;
    int64_t *B, *N;
    uint8_t *Nb;
    double *y, *d; /* dual phases only, else NULL (y: m, d: n_c) */
    /* private bookkeeping */
    int which;                /* 1 primal-1, 2 primal-2, 3 dual-1, 4 dual-2 */
    int64_t n_orig_vars;      /* prob.variables.len() of the user's problem */
    double *orig_obj;         /* user's objective coefficients (n_orig_vars) */
    uint8_t *orig_kind;       /* user's bounds */
    double *orig_lb, *orig_ub;
    int64_t n_p1vars;         /* primal: phase_1_vars */
    int64_t *p1vars;
    struct eo_phase *orig;    /* dual phase 1: orig_std_form as a phase without a point */
    int64_t *p1ids;           /* dual phase 1: id of box-problem variable k (orig column) */
} eo_phase;

void eo_phase_free(eo_phase *ph);
double eo_phase_obj(const eo_phase *ph);      /* standard_form.rs:48  c.dot(x) */
double eo_phase_dual_obj(const eo_phase *ph); /* standard_form.rs:52-68 */

/* primal_problem.rs:80-261; NULL => infeasible (None).  *err != 0 => reference would panic. */
eo_phase *eo_primal_phase1(const eo_problem *p, int *err);
/* primal_problem.rs:263-291 (does not consume its argument) */
eo_phase *eo_primal_phase2(const eo_phase *phase1);
/* dual_problem.rs:89-256 */
eo_phase *eo_dual_phase1(const eo_problem *p, int *err);
/* dual_problem.rs:258-404 */
eo_phase *eo_dual_phase2(const eo_phase *phase1, int *err);

/*
 * The hot loops.  primal_simplex_solver.rs:95-436 / dual_simplex_solver.rs:110-335,
 * LU refactorisation every iteration, same EPS rules and tie-breaks.
 * x, B, N, Nb (and y, d) are updated in place.  *iters = loop bodies entered.
 * If max_pivots_window != 0 the loop also stops (status EO_MAXITER) after that many
 * iterations — used for windowed state parity and the timed CPU baseline.
 */
int eo_primal_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                                 const double *c, const double *b, const uint8_t *kind,
                                 const double *lb, const double *ub, double *x, int64_t *B,
                                 int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN,
                                 uint64_t max_iter, uint64_t *iters, char *err, size_t errlen);
int eo_dual_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                               const double *c, const double *b, const uint8_t *kind,
                               const double *lb, const double *ub, double *x, int64_t *B,
                               int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN, double *y,
                               double *d, uint64_t max_iter, uint64_t *iters, char *err,
                               size_t errlen);

/*
 * Same pivoting rules as eo_primal_solve_with_initial, but B^-1 is kept explicitly and updated
 * by the eta formula (what the HIP engine does) and the big passes run on `threads` OpenMP
 * threads.  The "same algorithm on the host cores" baseline of bench.py and the checker for
 * long pivot windows at full size; checked against the LU-per-iteration loop in
 * tests/test_oracle_binv.py.  refresh > 0: B^-1 is rebuilt from an LU every `refresh` basis
 * changes.  *loop_seconds = time spent in the loop (the initial inverse excluded).
 */
int eo_primal_binv_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                                      const double *c, const double *b, const uint8_t *kind,
                                      const double *lb, const double *ub, double *x, int64_t *B,
                                      int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN,
                                      uint64_t max_iter, uint64_t *iters, int threads, int refresh,
                                      double *loop_seconds, char *err, size_t errlen);

/* The dual counterpart (dual…:200-333 with an explicit B^-1). */
int eo_dual_binv_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                                    const double *c, const double *b, const uint8_t *kind,
                                    const double *lb, const double *ub, double *x, int64_t *B,
                                    int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN, double *y,
                                    double *d, uint64_t max_iter, uint64_t *iters, int threads,
                                    int refresh, double *loop_seconds, char *err, size_t errlen);

/* The CERTIFIED HYBRID (round 4; not the reference's loop — the engine's default policy above 128 rows, restated; see
 * ellp_oracle.c): the explicit-inverse loop with a pivot guard (|pivot| < guard_abs stops it before the iteration is
 * committed), every terminal status and every guarded iteration handed to the LU-per-iteration loop for up to K iterations.
 * counters5: [0] hand-overs after a guard stop, [1] terminal statuses examined, [2] of those not confirmed, [3] iterations
 * of the exact loop, [4] solves repeated from their start by the exact loop because the end point violated an invariant of
 * the reference's loop ("certify or redo"). */
int eo_primal_hybrid_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                                        const double *b, const uint8_t *kind, const double *lb, const double *ub,
                                        double *x, int64_t *B, int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN,
                                        uint64_t max_iter, uint64_t *iters, int K, double guard_abs, int refresh,
                                        int threads, uint64_t *counters5, char *err, size_t errlen);
int eo_dual_hybrid_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                                      const double *b, const uint8_t *kind, const double *lb, const double *ub, double *x,
                                      int64_t *B, int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN, double *y, double *d,
                                      uint64_t max_iter, uint64_t *iters, int K, double guard_abs, int refresh,
                                      int threads, uint64_t *counters5, char *err, size_t errlen);

/* Optional per-iteration trace for pivot-sequence parity (entering position, leaving
 * position or -1, objective). Set to NULL to disable.  Not thread-safe (test use only). */
typedef void (*eo_trace_fn)(void *user, uint64_t iter, int64_t entering_pos,
                            int64_t leaving_pos, int64_t entering_var, int64_t leaving_var);
void eo_set_trace(eo_trace_fn fn, void *user);

/* 1: LU/solves do nalgebra's full dense work (no zero-multiplier skip) — used when the oracle
 * is TIMED as the CPU baseline; results are identical either way. */
void eo_set_dense_lu(int on);
/* threads for the once-per-solve setup factorizations (bitwise the one-thread results); default 1 */
void eo_set_setup_threads(int n);
/* dual extensions for eo_dual_solve_with_initial (see ellp_oracle.c): bit 0 bound-flipping ratio test, bit 1 leaving
 * row of largest violation; 0 = the reference's rules */
void eo_set_dual_rule(int bits);
/* primal extension for eo_primal_solve_with_initial (see ellp_oracle.c): 1 = steepest-edge pricing; 0 = the reference's rule */
void eo_set_primal_rule(int rule);
/* pivot guard of the certified hybrid in the explicit-inverse loops (see ellp_oracle.c); 0, 0 = off */
void eo_set_binv_guard(double rel, double abs_);
void eo_set_continuation(int on);
/* the explicit-inverse loops take entries of B^-1 a_q / rho . a_j below t to be zero (certified hybrid, see ellp_oracle.c); 0 = off */
void eo_set_binv_zero_tol(double t);
/* partial pricing for eo_primal_solve_with_initial (an extension, see ellp_oracle.c); P <= 1: off */
void eo_set_partial_segments(int P);

typedef struct eo_result {
    int status;       /* EO_OPTIMAL.. or error */
    double obj;       /* Optimal: c.x ; MaxIter: obj field of SolverResult::MaxIter */
    int64_t nx;       /* prob.variables.len() */
    double *x;        /* malloc'ed, nx entries (Optimal only) */
    uint64_t iters1, iters2;
    char err[256];
} eo_result;

/* PrimalSimplexSolver::solve (primal_simplex_solver.rs:32-93) when solver == 0,
 * DualSimplexSolver::solve (dual_simplex_solver.rs:33-108) when solver == 1. */
int eo_solve(const eo_problem *p, int solver, uint64_t max_iter, eo_result *out);
void eo_result_free(eo_result *r);

/* Deterministic synthetic LP family of SURVEY.md §8d (splitmix64): fills A (m x n,
 * column-major, ld = m), b (m), c (n). */
void eo_synth_dense_lp(uint64_t seed, int64_t m, int64_t n, double *A, double *b, double *c);

#ifdef __cplusplus
}
#endif
#endif
