/*
 * ellp_host.h — C view of the C++ host mirror (ellp_amd/csrc/host/ellp.h) so that Python
 * (ctypes) and C callers can drive the same Problem -> solve() flow the reference exposes
 * (src/lib.rs:109-129).  The simplex loops behind solve() run on the GPU via ellp_hip.h.
 */
#ifndef ELLP_HOST_H
#define ELLP_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "ellp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ellp_problem ellp_problem;

/* ConstraintOp (problem.rs:299-303) */
enum { ELLP_OP_LTE = 0, ELLP_OP_EQ = 1, ELLP_OP_GTE = 2 };
enum { ELLP_SOLVER_PRIMAL = 0, ELLP_SOLVER_DUAL = 1 };

ellp_problem *ellp_problem_new(void);                      /* Problem::new         problem.rs:20 */
ellp_problem *ellp_problem_clone(const ellp_problem *p);
void ellp_problem_free(ellp_problem *p);
/* Problem::add_var (problem.rs:24): returns the id or -1 (message in errbuf). name may be NULL. */
int64_t ellp_problem_add_var(ellp_problem *p, double obj_coeff, int bound_kind, double lb, double ub,
                             const char *name, char *errbuf, size_t errbuf_len);
int64_t ellp_problem_add_var_with_id(ellp_problem *p, double obj_coeff, int bound_kind, double lb, double ub,
                                     int64_t id, const char *name, char *errbuf, size_t errbuf_len);
/* Problem::add_constraint (problem.rs:85): 0 ok, -1 error. */
int ellp_problem_add_constraint(ellp_problem *p, int64_t n, const int64_t *var_ids, const double *coeffs,
                                int op, double rhs, char *errbuf, size_t errbuf_len);
int64_t ellp_problem_num_vars(const ellp_problem *p);
int64_t ellp_problem_num_constraints(const ellp_problem *p);
/* Problem::is_feasible (problem.rs:108) */
int ellp_problem_is_feasible(const ellp_problem *p, const double *x, int64_t n);
/* parse_mps (parse_mps.rs:23): NULL on error (message in errbuf). */
ellp_problem *ellp_parse_mps(const char *text, char *errbuf, size_t errbuf_len);

/* SolverResult (solver.rs:6-12) flattened. */
typedef struct ellp_result {
    int status;        /* ellp_status: OPTIMAL / INFEASIBLE / UNBOUNDED / MAXITER or an error */
    double obj;        /* Optimal: sol.obj(); MaxIter: the `obj` field */
    int64_t nx;        /* prob.variables.len() */
    double *x;         /* Optimal: sol.x(), owned by the result (ellp_result_free) */
    uint64_t iters_phase1, iters_phase2;
    char err[512];
} ellp_result;

/* PrimalSimplexSolver / DualSimplexSolver ::solve (primal…:32, dual…:33).
 * max_iter: ELLP_MAX_ITER_NONE = ::new(None); 1000 = ::default().  opts may be NULL; only its
 * engine fields (device, refactor_period, btran_mode, poll_interval) are used. */
int ellp_solve(const ellp_problem *p, int solver, uint64_t max_iter, const ellp_opts *opts, ellp_result *out);
void ellp_result_free(ellp_result *r);

/* Test/diagnostic tap: the flattened phase-1 problem (exactly the arrays that solve() hands to
 * ellp_*_solve_with_initial) so the host setup can be checked without a GPU.  All arrays are
 * owned by the struct (ellp_flat_phase_free).  Returns 0, 1 if the setup already proves
 * infeasibility (None), or a negative ellp_status. */
typedef struct ellp_flat_phase {
    int64_t m, n, n_c, n_B, n_N;
    double *A, *c, *b, *lb, *ub, *x, *y, *d;
    uint8_t *bound_kind, *N_bound;
    int64_t *B_index, *N_index;
} ellp_flat_phase;
int ellp_debug_phase1(const ellp_problem *p, int solver, ellp_flat_phase *out, char *errbuf, size_t errbuf_len);
void ellp_flat_phase_free(ellp_flat_phase *f);

#ifdef __cplusplus
}
#endif
#endif
