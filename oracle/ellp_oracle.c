/*
 * ellp_oracle.c — CPU oracle (TEST INFRASTRUCTURE, see ellp_oracle.h).
 *
 * Plain-C restatement of kehlert/ellp 0.2.0.  File:line citations are relative to the
 * reference tree.  Single-threaded on purpose: nalgebra, which carries all of the
 * reference's arithmetic, is single-threaded.
 *
 * Build with -ffp-contract=off: Rust never contracts a*b+c into an FMA.
 */
#include "ellp_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EPS EO_EPS

/* ------------------------------------------------------------------ utilities */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "ellp_oracle: out of memory (%zu bytes)\n", n);
        abort();
    }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) {
        fprintf(stderr, "ellp_oracle: out of memory\n");
        abort();
    }
    return p;
}
static double *dcopy(const double *s, int64_t n) {
    double *d = (double *)xmalloc(sizeof(double) * (size_t)n);
    if (n) memcpy(d, s, sizeof(double) * (size_t)n);
    return d;
}
static int64_t *icopy(const int64_t *s, int64_t n) {
    int64_t *d = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)n);
    if (n) memcpy(d, s, sizeof(int64_t) * (size_t)n);
    return d;
}
static uint8_t *bcopy8(const uint8_t *s, int64_t n) {
    uint8_t *d = (uint8_t *)xmalloc((size_t)n);
    if (n) memcpy(d, s, (size_t)n);
    return d;
}
static void set_err(char *err, size_t errlen, const char *msg) {
    if (err && errlen) {
        snprintf(err, errlen, "%s", msg);
    }
}

/* nalgebra's gauss_step runs its axpy for every trailing column, zero multiplier or not.
 * Skipping a zero multiplier is value-identical for finite data and keeps the test-suite fast
 * on slack/artificial-heavy bases; eo_set_dense_lu(1) switches the skip off so that the timed
 * CPU baseline pays the reference's full (2/3) m^3 per iteration. */
static int g_dense_lu = 0;
void eo_set_dense_lu(int on) { g_dense_lu = on; }

/* Threads for the ONCE-PER-SOLVE setup factorizations (the rank check's QR, the LU of A^T that picks
 * the dual phase-1 basis) when a test builds phase arrays at BASELINE.json's full sizes (2 x 56 GFLOP of
 * QR at config 3).  Only loops over independent columns are shared out and every column is still
 * processed by one thread in the sequential order, so every number is bit for bit the one-thread
 * result.  1 (the default) everywhere else — in particular for the timed CPU baseline of bench.py, which
 * is the reference's single-threaded algorithm. */
/* Partial pricing (NOT the reference's behaviour: an extension its README lists as future work),
 * restated here so that the engine's opt-in implementation (ellp_opts.partial_segments) has a checker: the nonbasic
 * positions are cut into P segments of S = ceil(|N| / P) positions; a loop body prices the current segment with the
 * reference's entering rule; a body that finds no candidate moves to the next segment, and P such bodies in a row
 * are the optimality test.  P <= 1: off.  Used by eo_primal_solve_with_initial only. */
static int g_partial_segments = 0;
void eo_set_partial_segments(int P) { g_partial_segments = P > 1 ? P : 0; }

/* Dual extensions (NOT the reference's rules, off by default; eo_set_dual_rule):
 *   bit 0  bound-flipping ("long-step") ratio test: the dual step goes past the breakpoints of BOXED nonbasic
 *          variables — each is moved to its other bound instead of entering — for as long as the dual objective
 *          keeps growing, i.e. while the infeasibility |delta| of the leaving row minus the sum of
 *          |alpha_j| (ub_j - lb_j) over the passed breakpoints stays positive; the breakpoint at which it
 *          would turn non-positive (or the first one of a variable that has no other bound) enters.
 *          Breakpoints are taken in the order (ratio, position).
 *   bit 1  the leaving row is the one with the LARGEST bound violation (first of equals) instead of the first
 *          violated one. */
static int g_dual_rule = 0;
void eo_set_dual_rule(int bits) { g_dual_rule = bits; }

/* Primal extension (NOT the reference's rule, off by default; eo_set_primal_rule(1)): steepest-edge
 * pricing.  Every nonbasic position carries gamma_j = 1 + |B^-1 a_j|^2 (exact at the start, whatever the basis); the entering candidates are the reference's
 * (primal…:253-270) but the key that goes through the reference's fold (:271-287) is |r_j| / sqrt(gamma_j) instead of |r_j|.
 * gamma is exact at the start at EVERY basis: 1 + |a_j|^2 read off the columns at a signed permutation (every phase-1 start),
 * 1 + |B^-1 a_j|^2 from one LU of the basis otherwise (the engine: from the inverse it has just built); after a pivot (entering position q, leaving row r, alpha_q = B^-1 a_q, rho = row r of
 * B^-1, v = B^-T alpha_q) it is updated exactly (Goldfarb & Reid): with abar_j = (rho . a_j) / alpha_q[r],
 *   gamma_j <- max(gamma_j - 2 abar_j (a_j . v) + abar_j^2 gamma_q, 1 + abar_j^2),   gamma_leaving <- max(gamma_q / alpha_q[r]^2, 1)
 * with gamma_q = 1 + |alpha_q|^2 taken exactly; a bound flip leaves the weights alone.
 * (rule 2: Devex without framework resets, gamma_j <- max(gamma_j, abar_j^2 gamma_q) from all ones — kept as a measured
 * negative result: 6 x MORE iterations than Dantzig in phase 2.) */
static int g_primal_rule = 0;
void eo_set_primal_rule(int rule) { g_primal_rule = rule; }

/* Pivot guard of the CERTIFIED HYBRID (round 4; NOT the reference's behaviour — the policy of the engine's default above
 * 128 rows, restated here so that it has a checker; off by default; eo_set_binv_guard).  The explicit-inverse loops
 * (eo_*_binv_*) stop with EO_NEED_EXACT, BEFORE anything of the iteration is committed (x, the basis and the iteration
 * counter stand as the previous iteration left them), when the pivot element they are about to use is suspicious:
 *   |pivot| < abs   or   |pivot| < rel * max_i |column_i|      (column = B^-1 a_q)
 * An explicit inverse is good to cond(A_B) * 2^-53; an entry of B^-1 a_q that is a structural zero comes out of it as
 * ~1e-9 on the ill-conditioned bases of real LPs, passes the reference's |.| >= EPS test, wins a degenerate ratio test
 * and makes the basis singular.  The caller then runs the LU-per-iteration loop (the reference's arithmetic, which keeps
 * such entries at 0) for a few iterations from the same arrays and hands back. */
static double g_guard_rel = 0.0, g_guard_abs = 0.0;
void eo_set_binv_guard(double rel, double abs_) {
    g_guard_rel = rel > 0.0 ? rel : 0.0;
    g_guard_abs = abs_ > 0.0 ? abs_ : 0.0;
}
/* Second half of the certified hybrid's fast loop (eo_set_binv_zero_tol; 0 = off): an entry of B^-1 a_q, or of the dual's
 * pricing row rho . a_j, whose modulus is below zero_tol is taken to BE zero — in the ratio tests and in the updates of
 * x_B and d alike.  An explicit inverse is good to cond(A_B) * 2^-53; on the sparse bases of real LPs most entries of these
 * vectors are structural zeros that come out of it as +-1e-9, and where the LU solves of the reference return exact zeros
 * for them and leave x_B / d untouched, every pivot of the explicit-inverse loop adds theta * 1e-9 to each: the carried
 * d drifts until a ratio test picks the wrong column and the phase ends at a basis that is dual infeasible by 1e-9
 * (ADLITTLE x 18: 8 of 30 dual phase-1 runs).  A genuine entry below zero_tol would be refused as a pivot by the guard
 * anyway (zero_tol < guard_abs). */
static double g_zero_tol = 0.0;
void eo_set_binv_zero_tol(double t) { g_zero_tol = t > 0.0 ? t : 0.0; }

static int guard_trips(double pivot, const double *col, int64_t m) {
    if (g_guard_rel <= 0.0 && g_guard_abs <= 0.0) return 0;
    const double ap = fabs(pivot);
    if (ap < g_guard_abs) return 1;
    if (g_guard_rel > 0.0) {
        double mx = 0.0;
        for (int64_t i = 0; i < m; ++i)
            if (fabs(col[i]) > mx) mx = fabs(col[i]);
        if (ap < g_guard_rel * mx) return 1;
    }
    return 0;
}

/* certified hybrid: the dual loops are being CONTINUED from a point another loop of the same solve left (a hand-over in
 * the middle of solve_with_initial), so the entry assertion on the initial point (dual…:139-151) does not apply */
static int g_continuation = 0;
void eo_set_continuation(int on) { g_continuation = on ? 1 : 0; }

static int g_setup_threads = 1;
void eo_set_setup_threads(int n) { g_setup_threads = n > 1 ? n : 1; }

static eo_trace_fn g_trace = NULL;
static void *g_trace_user = NULL;
void eo_set_trace(eo_trace_fn fn, void *user) {
    g_trace = fn;
    g_trace_user = user;
}

/* f64::signum (Rust): 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN */
static double rust_signum(double v) {
    if (isnan(v)) return v;
    return signbit(v) ? -1.0 : 1.0;
}

/* ------------------------------------------------------------------ permutation sequences
 * nalgebra PermutationSequence: a list of row transpositions (i, j).
 * permute_rows applies them in recording order, inv_permute_rows in reverse order. */
typedef struct {
    int64_t len;
    int64_t *a, *b;
} perm_t;

static void perm_init(perm_t *p, int64_t cap) {
    p->len = 0;
    p->a = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(cap > 0 ? cap : 1));
    p->b = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(cap > 0 ? cap : 1));
}
static void perm_free(perm_t *p) {
    free(p->a);
    free(p->b);
    p->a = p->b = NULL;
    p->len = 0;
}
static void perm_push(perm_t *p, int64_t i, int64_t j) {
    p->a[p->len] = i;
    p->b[p->len] = j;
    p->len++;
}
static void perm_rows_d(const perm_t *p, double *v) {
    for (int64_t k = 0; k < p->len; ++k) {
        double t = v[p->a[k]];
        v[p->a[k]] = v[p->b[k]];
        v[p->b[k]] = t;
    }
}
static void inv_perm_rows_d(const perm_t *p, double *v) {
    for (int64_t k = p->len - 1; k >= 0; --k) {
        double t = v[p->a[k]];
        v[p->a[k]] = v[p->b[k]];
        v[p->b[k]] = t;
    }
}
static void perm_rows_i(const perm_t *p, int64_t *v) {
    for (int64_t k = 0; k < p->len; ++k) {
        int64_t t = v[p->a[k]];
        v[p->a[k]] = v[p->b[k]];
        v[p->b[k]] = t;
    }
}

/* ------------------------------------------------------------------ LU (nalgebra linalg::LU)
 * Partial (row) pivoting, P*A = L*U, unit-diagonal L stored below the diagonal of `lu`.
 * Pivot = FIRST entry of maximal |.| in the column (iamax uses a strict '>').
 * A zero pivot column is skipped (no elimination), as nalgebra does. */
typedef struct {
    int64_t nr, nc;
    double *lu; /* nr x nc, col-major, ld = nr */
    perm_t p;
} lu_t;

static void lu_free(lu_t *f) {
    free(f->lu);
    f->lu = NULL;
    perm_free(&f->p);
}

/* gauss_step / gauss_step_swap: multipliers are a * (1/diag) (two roundings), the trailing
 * update is down[:,k] += (-pivot_row[k]) * coeffs (mul then add). */
static void lu_factor_inplace(lu_t *f) {
    const int64_t nr = f->nr, nc = f->nc;
    const int64_t mn = nr < nc ? nr : nc;
    double *M = f->lu;
    for (int64_t i = 0; i < mn; ++i) {
        double *ci = M + i * nr;
        int64_t piv = i;
        double best = fabs(ci[i]);
        for (int64_t r = i + 1; r < nr; ++r) {
            double v = fabs(ci[r]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        double diag = ci[piv];
        if (diag == 0.0) continue;
        if (piv != i) {
            perm_push(&f->p, i, piv);
            /* swap rows i, piv in columns [0, i) and in column i .. nc (gauss_step_swap) */
#pragma omp parallel for if (g_setup_threads > 1 && nc > 256) num_threads(g_setup_threads) schedule(static)
            for (int64_t k = 0; k < nc; ++k) {
                double *ck = M + k * nr;
                double t = ck[i];
                ck[i] = ck[piv];
                ck[piv] = t;
            }
        }
        double inv_diag = 1.0 / diag;
        for (int64_t r = i + 1; r < nr; ++r) ci[r] *= inv_diag;
#pragma omp parallel for if (g_setup_threads > 1 && nc - i > 64) num_threads(g_setup_threads) schedule(static)
        for (int64_t k = i + 1; k < nc; ++k) {
            double *ck = M + k * nr;
            double a = -ck[i];
            if (a == 0.0 && !g_dense_lu) continue; /* see eo_set_dense_lu */
            for (int64_t r = i + 1; r < nr; ++r) ck[r] = a * ci[r] + ck[r];
        }
    }
}

static void lu_factor(lu_t *f, const double *A, int64_t nr, int64_t nc) {
    f->nr = nr;
    f->nc = nc;
    f->lu = dcopy(A, nr * nc);
    perm_init(&f->p, (nr < nc ? nr : nc) + 1);
    lu_factor_inplace(f);
}

/* any |U_ii| < EPS over the min(nr,nc) diagonal */
static int lu_small_diag(const lu_t *f) {
    int64_t mn = f->nr < f->nc ? f->nr : f->nc;
    for (int64_t i = 0; i < mn; ++i)
        if (fabs(f->lu[i + i * f->nr]) < EPS) return 1;
    return 0;
}

/* U^T x = b  (tr_solve_upper_triangular): x_i = (b_i - U[0..i,i].x[0..i]) / U_ii */
static int lu_tr_solve_upper(const lu_t *f, double *b) {
    const int64_t n = f->nr;
    for (int64_t i = 0; i < n; ++i) {
        const double *ci = f->lu + i * n;
        double dot = 0.0;
        for (int64_t k = 0; k < i; ++k) dot += ci[k] * b[k];
        b[i] -= dot;
        double diag = ci[i];
        if (diag == 0.0) return 0;
        b[i] /= diag;
    }
    return 1;
}
/* L^T x = b, unit diagonal (tr_solve_lower_triangular on l()) */
static int lu_tr_solve_lower_unit(const lu_t *f, double *b) {
    const int64_t n = f->nr;
    for (int64_t i = n - 1; i >= 0; --i) {
        const double *ci = f->lu + i * n;
        double dot = 0.0;
        for (int64_t k = i + 1; k < n; ++k) dot += ci[k] * b[k];
        b[i] -= dot;
        b[i] /= 1.0;
    }
    return 1;
}
/* LU::solve: x = P b ; L y = x (column-oriented, unit diag) ; U z = y (column-oriented) */
static int lu_solve(const lu_t *f, double *b) {
    const int64_t n = f->nr;
    perm_rows_d(&f->p, b);
    for (int64_t i = 0; i + 1 < n; ++i) {
        double coeff = b[i];
        if (coeff == 0.0 && !g_dense_lu) continue;
        const double *ci = f->lu + i * n;
        double a = -coeff;
        for (int64_t r = i + 1; r < n; ++r) b[r] = a * ci[r] + b[r];
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        const double *ci = f->lu + i * n;
        double diag = ci[i];
        if (diag == 0.0) return 0;
        double coeff = b[i] / diag;
        b[i] = coeff;
        if (coeff == 0.0 && !g_dense_lu) continue;
        double a = -coeff;
        for (int64_t r = 0; r < i; ++r) b[r] = a * ci[r] + b[r];
    }
    return 1;
}
/* u = A_B^{-T} c_B :  U^T ut = c ; L^T w = ut ; u = P^{-1} w  (primal…:184-187) */
static int lu_btran(const lu_t *f, double *v) {
    if (!lu_tr_solve_upper(f, v)) return 0;
    if (!lu_tr_solve_lower_unit(f, v)) return 0;
    inv_perm_rows_d(&f->p, v);
    return 1;
}

/* ------------------------------------------------------------------ Problem (problem.rs) */

typedef struct {
    int64_t id;
    double obj;
    int kind;
    double lb, ub;
} var_t;
typedef struct {
    int64_t n;
    int64_t *ids;
    double *coef;
    int op;
    double rhs;
} con_t;

struct eo_problem {
    var_t *vars;
    int64_t nvars, capv;
    con_t *cons;
    int64_t ncons, capc;
    uint8_t *idset; /* idset[id] = 1 if used */
    int64_t idcap;
};

eo_problem *eo_problem_new(void) { return (eo_problem *)xcalloc(1, sizeof(eo_problem)); }

void eo_problem_free(eo_problem *p) {
    if (!p) return;
    for (int64_t i = 0; i < p->ncons; ++i) {
        free(p->cons[i].ids);
        free(p->cons[i].coef);
    }
    free(p->cons);
    free(p->vars);
    free(p->idset);
    free(p);
}

int64_t eo_num_vars(const eo_problem *p) { return p->nvars; }
int64_t eo_num_constraints(const eo_problem *p) { return p->ncons; }

/* problem.rs:35-83 */
int64_t eo_add_var_with_id(eo_problem *p, double obj, int kind, double lb, double ub, int64_t id) {
    if (id < 0) return -1;
    switch (kind) {
    case EO_FREE: break;
    case EO_LOWER:
        if (!isfinite(lb)) return -1;
        break;
    case EO_UPPER:
        if (!isfinite(ub)) return -1;
        break;
    case EO_TWOSIDED:
        if (lb > ub) return -1;
        if (!isfinite(lb) || !isfinite(ub)) return -1;
        break;
    case EO_FIXED:
        if (!isfinite(lb)) return -1;
        ub = lb;
        break;
    default: return -1;
    }
    if (id >= p->idcap) {
        int64_t nc = p->idcap ? p->idcap : 64;
        while (nc <= id) nc *= 2;
        p->idset = (uint8_t *)realloc(p->idset, (size_t)nc);
        memset(p->idset + p->idcap, 0, (size_t)(nc - p->idcap));
        p->idcap = nc;
    }
    if (p->idset[id]) return -1;
    p->idset[id] = 1;
    if (p->nvars == p->capv) {
        p->capv = p->capv ? 2 * p->capv : 64;
        p->vars = (var_t *)realloc(p->vars, sizeof(var_t) * (size_t)p->capv);
    }
    var_t *v = &p->vars[p->nvars++];
    v->id = id;
    v->obj = obj;
    v->kind = kind;
    v->lb = lb;
    v->ub = ub;
    return id;
}
/* problem.rs:24-33 */
int64_t eo_add_var(eo_problem *p, double obj, int kind, double lb, double ub) {
    return eo_add_var_with_id(p, obj, kind, lb, ub, p->nvars);
}
/* problem.rs:85-106 */
int eo_add_constraint(eo_problem *p, int64_t n, const int64_t *ids, const double *coef, int op,
                      double rhs) {
    for (int64_t k = 0; k < n; ++k) {
        if (ids[k] < 0 || ids[k] >= p->idcap || !p->idset[ids[k]]) return -1;
    }
    if (p->ncons == p->capc) {
        p->capc = p->capc ? 2 * p->capc : 64;
        p->cons = (con_t *)realloc(p->cons, sizeof(con_t) * (size_t)p->capc);
    }
    con_t *c = &p->cons[p->ncons++];
    c->n = n;
    c->ids = icopy(ids, n);
    c->coef = dcopy(coef, n);
    c->op = op;
    c->rhs = rhs;
    return 0;
}

/* ------------------------------------------------------------------ phases */

static eo_phase *phase_alloc(int64_t m, int64_t n, int64_t n_c) {
    eo_phase *ph = (eo_phase *)xcalloc(1, sizeof(eo_phase));
    ph->m = m;
    ph->n = n;
    ph->n_c = n_c;
    ph->A = (double *)xcalloc((size_t)(m * n), sizeof(double));
    ph->c = (double *)xcalloc((size_t)n_c, sizeof(double));
    ph->b = (double *)xcalloc((size_t)m, sizeof(double));
    ph->kind = (uint8_t *)xcalloc((size_t)n_c, 1);
    ph->lb = (double *)xcalloc((size_t)n_c, sizeof(double));
    ph->ub = (double *)xcalloc((size_t)n_c, sizeof(double));
    ph->x = (double *)xcalloc((size_t)n_c, sizeof(double));
    ph->B = (int64_t *)xcalloc((size_t)(n_c + m + 1), sizeof(int64_t));
    ph->N = (int64_t *)xcalloc((size_t)(n_c + m + 1), sizeof(int64_t));
    ph->Nb = (uint8_t *)xcalloc((size_t)(n_c + m + 1), 1);
    return ph;
}

void eo_phase_free(eo_phase *ph) {
    if (!ph) return;
    free(ph->A);
    free(ph->c);
    free(ph->b);
    free(ph->kind);
    free(ph->lb);
    free(ph->ub);
    free(ph->x);
    free(ph->B);
    free(ph->N);
    free(ph->Nb);
    free(ph->y);
    free(ph->d);
    free(ph->orig_obj);
    free(ph->orig_kind);
    free(ph->orig_lb);
    free(ph->orig_ub);
    free(ph->p1vars);
    free(ph->p1ids);
    if (ph->orig) eo_phase_free(ph->orig);
    free(ph);
}

static void phase_keep_orig(eo_phase *ph, const eo_problem *p) {
    ph->n_orig_vars = p->nvars;
    ph->orig_obj = (double *)xmalloc(sizeof(double) * (size_t)p->nvars);
    ph->orig_kind = (uint8_t *)xmalloc((size_t)p->nvars);
    ph->orig_lb = (double *)xmalloc(sizeof(double) * (size_t)p->nvars);
    ph->orig_ub = (double *)xmalloc(sizeof(double) * (size_t)p->nvars);
    for (int64_t i = 0; i < p->nvars; ++i) {
        ph->orig_obj[i] = p->vars[i].obj;
        ph->orig_kind[i] = (uint8_t)p->vars[i].kind;
        ph->orig_lb[i] = p->vars[i].lb;
        ph->orig_ub[i] = p->vars[i].ub;
    }
}
static void phase_copy_orig(eo_phase *dst, const eo_phase *src) {
    dst->n_orig_vars = src->n_orig_vars;
    dst->orig_obj = dcopy(src->orig_obj, src->n_orig_vars);
    dst->orig_kind = bcopy8(src->orig_kind, src->n_orig_vars);
    dst->orig_lb = dcopy(src->orig_lb, src->n_orig_vars);
    dst->orig_ub = dcopy(src->orig_ub, src->n_orig_vars);
}

/* standard_form.rs:48 */
double eo_phase_obj(const eo_phase *ph) {
    double s = 0.0;
    for (int64_t i = 0; i < ph->n_c; ++i) s += ph->c[i] * ph->x[i];
    return s;
}
/* standard_form.rs:52-68 */
static double dual_obj(int64_t m, int64_t n_c, const double *b, const uint8_t *kind,
                       const double *lb, const double *ub, const double *y, const double *d) {
    double obj = 0.0;
    for (int64_t i = 0; i < m; ++i) obj += b[i] * y[i];
    for (int64_t i = 0; i < n_c; ++i) {
        switch (kind[i]) {
        case EO_FREE: break;
        case EO_LOWER: obj += lb[i] * d[i]; break;
        case EO_UPPER: obj += ub[i] * d[i]; break;
        case EO_TWOSIDED: obj += (d[i] > 0.0) ? lb[i] * d[i] : ub[i] * d[i]; break;
        case EO_FIXED: obj += lb[i] * d[i]; break;
        }
    }
    return obj;
}
double eo_phase_dual_obj(const eo_phase *ph) {
    if (!ph->y || !ph->d) return NAN;
    return dual_obj(ph->m, ph->n_c, ph->b, ph->kind, ph->lb, ph->ub, ph->y, ph->d);
}

/* ------------------------------------------------------------------ StandardForm
 * standard_form.rs:78-191.  Returns a phase without a point (nB = nN = 0), or NULL if None. */

/* Column-pivoted Householder QR of M (nr x nc, col-major, destroyed), nalgebra ColPivQR:
 * at step i the pivot column is the column holding the entry of maximal |.| of the
 * trailing block (icamax_full: column-major scan, strict '>'), not the largest-norm column.
 * Outputs the swap sequence and |R_ii|. */
static void col_piv_qr(double *M, int64_t nr, int64_t nc, perm_t *p, double *rdiag) {
    const int64_t mn = nr < nc ? nr : nc;
    for (int64_t i = 0; i < mn; ++i) {
        int64_t pj = i;
        double best = fabs(M[i + i * nr]);
        if (g_setup_threads > 1 && (nc - i) * (nr - i) > 65536) {
            /* the column-major scan with a strict '>' ends at the FIRST column that holds the maximum:
             * per-thread (maximum, first column holding it) over contiguous column ranges, combined in
             * thread order — the same column */
            const int nt = g_setup_threads;
            double tb[256];
            int64_t tj[256];
            const int ntc = nt < 256 ? nt : 256;
#pragma omp parallel num_threads(ntc)
            {
#ifdef _OPENMP
                const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
                const int t = 0, T = 1;
#endif
                const int64_t span = nc - i, lo = i + span * t / T, hi = i + span * (t + 1) / T;
                double b = -1.0;
                int64_t bj = -1;
                for (int64_t j = lo; j < hi; ++j) {
                    const double *cj = M + j * nr;
                    for (int64_t r = i; r < nr; ++r) {
                        double v = fabs(cj[r]);
                        if (v > b) {
                            b = v;
                            bj = j;
                        }
                    }
                }
                tb[t] = b;
                tj[t] = bj;
#pragma omp single
                { tb[255] = (double)T; }
            }
            const int T = (int)tb[255] < ntc ? (int)tb[255] : ntc;
            for (int t = 0; t < T; ++t)
                if (tj[t] >= 0 && tb[t] > best) {
                    best = tb[t];
                    pj = tj[t];
                }
        } else
        for (int64_t j = i; j < nc; ++j) {
            const double *cj = M + j * nr;
            for (int64_t r = i; r < nr; ++r) {
                double v = fabs(cj[r]);
                if (v > best) {
                    best = v;
                    pj = j;
                }
            }
        }
        if (pj != i) {
            double *a = M + i * nr, *b = M + pj * nr;
            for (int64_t r = 0; r < nr; ++r) {
                double t = a[r];
                a[r] = b[r];
                b[r] = t;
            }
        }
        perm_push(p, i, pj);
        /* householder::reflection_axis_mut on M[i.., i] */
        double *v = M + i + i * nr;
        const int64_t len = nr - i;
        double sqn = 0.0;
        for (int64_t r = 0; r < len; ++r) sqn += v[r] * v[r];
        double norm = sqrt(sqn);
        double modulus = fabs(v[0]);
        double sign = (v[0] < 0.0) ? -1.0 : 1.0;
        double signed_norm = sign * norm;
        double factor = (sqn + modulus * norm) * 2.0;
        v[0] += signed_norm;
        if (factor != 0.0) {
            double sf = sqrt(factor);
            for (int64_t r = 0; r < len; ++r) v[r] /= sf;
            double n2 = 0.0;
            for (int64_t r = 0; r < len; ++r) n2 += v[r] * v[r];
            n2 = sqrt(n2);
            if (n2 != 0.0)
                for (int64_t r = 0; r < len; ++r) v[r] /= n2;
            rdiag[i] = fabs(signed_norm);
            /* reflect the trailing columns: col -= 2 (v.col) v  (overall sign immaterial:
             * only |entries| steer later pivots and only |R_ii| is consumed) */
#pragma omp parallel for if (g_setup_threads > 1 && (nc - i) * len > 65536) num_threads(g_setup_threads) schedule(static)
            for (int64_t j = i + 1; j < nc; ++j) {
                double *cj = M + i + j * nr;
                double dot = 0.0;
                for (int64_t r = 0; r < len; ++r) dot += v[r] * cj[r];
                double f2 = -2.0 * dot;
                for (int64_t r = 0; r < len; ++r) cj[r] = f2 * v[r] + cj[r];
            }
        } else {
            rdiag[i] = fabs(signed_norm);
        }
    }
}

static eo_phase *standard_form(const eo_problem *prob) {
    const int64_t n = prob->nvars;
    const int64_t m = prob->ncons;
    int64_t num_slack = 0;
    for (int64_t i = 0; i < m; ++i)
        if (prob->cons[i].op != EO_EQ) num_slack++;
    const int64_t tv = n + num_slack;

    double *A = (double *)xcalloc((size_t)(m * tv), sizeof(double)); /* m x tv, ld m */
    double *b = (double *)xcalloc((size_t)m, sizeof(double));
    double *c = (double *)xcalloc((size_t)tv, sizeof(double));
    uint8_t *kind = (uint8_t *)xmalloc((size_t)tv);
    double *lb = (double *)xcalloc((size_t)tv, sizeof(double));
    double *ub = (double *)xcalloc((size_t)tv, sizeof(double));
    for (int64_t i = 0; i < tv; ++i) kind[i] = EO_LOWER; /* Lower(0.) default :106 */

    int64_t maxid = -1;
    for (int64_t i = 0; i < n; ++i)
        if (prob->vars[i].id > maxid) maxid = prob->vars[i].id;
    int64_t *id_to_index = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(maxid + 2));
    for (int64_t i = 0; i < n; ++i) {
        c[i] = prob->vars[i].obj;
        kind[i] = (uint8_t)prob->vars[i].kind;
        lb[i] = prob->vars[i].lb;
        ub[i] = prob->vars[i].ub;
        id_to_index[prob->vars[i].id] = i;
    }

    int64_t cur_slack_col = tv > 0 ? tv - 1 : 0; /* saturating_sub :115 */
    int infeasible = 0;
    for (int64_t i = 0; i < m && !infeasible; ++i) {
        const con_t *cn = &prob->cons[i];
        b[i] = cn->rhs;
        if (cn->n == 0 && b[i] != 0.0) {
            infeasible = 1;
            break;
        }
        for (int64_t k = 0; k < cn->n; ++k) A[i + id_to_index[cn->ids[k]] * m] = cn->coef[k];
        if (cn->op != EO_EQ) {
            A[i + cur_slack_col * m] = (cn->op == EO_LTE) ? 1.0 : -1.0;
            cur_slack_col -= 1;
        }
    }
    free(id_to_index);
    if (infeasible) {
        free(A); free(b); free(c); free(kind); free(lb); free(ub);
        return NULL;
    }

    /* remove redundant rows: A^T.col_piv_qr()  :142-181 */
    const int64_t mn = tv < m ? tv : m;
    double *At = (double *)xmalloc(sizeof(double) * (size_t)(tv * m + 1)); /* tv x m */
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < tv; ++j) At[j + i * tv] = A[i + j * m];
    perm_t p;
    perm_init(&p, mn + 1);
    double *rdiag = (double *)xcalloc((size_t)(mn + 1), sizeof(double));
    col_piv_qr(At, tv, m, &p, rdiag);
    free(At);

    inv_perm_rows_d(&p, b);
    for (int64_t i = 0; i < mn; ++i)
        if (fabs(rdiag[i]) < EPS) rdiag[i] = 0.0;
    const int r_nonempty = (mn > 0 && m > 0);
    const int is_trivial = r_nonempty && fabs(rdiag[0]) < EPS && fabs(b[0]) < EPS;
    if (!is_trivial) {
        /* R.tr_solve_upper_triangular(&b)? : fails iff a diagonal entry is exactly zero */
        for (int64_t i = 0; i < mn; ++i) {
            if (rdiag[i] == 0.0) {
                infeasible = 1;
                break;
            }
        }
    }
    perm_rows_d(&p, b);
    if (infeasible) {
        perm_free(&p);
        free(rdiag); free(A); free(b); free(c); free(kind); free(lb); free(ub);
        return NULL;
    }
    int64_t num_indep = mn;
    for (int64_t i = 0; i < mn; ++i) {
        if (fabs(rdiag[i]) < EPS) {
            num_indep = i;
            break;
        }
    }
    int64_t *rows = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(m + 1));
    for (int64_t i = 0; i < m; ++i) rows[i] = i;
    perm_rows_i(&p, rows);
    perm_free(&p);
    free(rdiag);

    eo_phase *sf = phase_alloc(num_indep, tv, tv);
    for (int64_t k = 0; k < num_indep; ++k) {
        const int64_t src = rows[k];
        for (int64_t j = 0; j < tv; ++j) sf->A[k + j * num_indep] = A[src + j * m];
        sf->b[k] = b[src];
    }
    memcpy(sf->c, c, sizeof(double) * (size_t)tv);
    memcpy(sf->kind, kind, (size_t)tv);
    memcpy(sf->lb, lb, sizeof(double) * (size_t)tv);
    memcpy(sf->ub, ub, sizeof(double) * (size_t)tv);
    phase_keep_orig(sf, prob);
    free(rows); free(A); free(b); free(c); free(kind); free(lb); free(ub);
    return sf;
}

/* ------------------------------------------------------------------ full-pivot LU (nalgebra FullPivLU) */
typedef struct {
    int64_t nr, nc;
    double *lu;
    perm_t p, q;
} fplu_t;

static void fplu_factor(fplu_t *f, const double *A, int64_t nr, int64_t nc) {
    f->nr = nr;
    f->nc = nc;
    f->lu = dcopy(A, nr * nc);
    const int64_t mn = nr < nc ? nr : nc;
    perm_init(&f->p, mn + 1);
    perm_init(&f->q, mn + 1);
    double *M = f->lu;
    for (int64_t i = 0; i < mn; ++i) {
        int64_t pr = i, pc = i;
        double best = fabs(M[i + i * nr]);
        for (int64_t j = i; j < nc; ++j) {
            for (int64_t r = i; r < nr; ++r) {
                double v = fabs(M[r + j * nr]);
                if (v > best) {
                    best = v;
                    pr = r;
                    pc = j;
                }
            }
        }
        double diag = M[pr + pc * nr];
        if (diag == 0.0) break;
        if (pc != i) {
            double *a = M + i * nr, *b = M + pc * nr;
            for (int64_t r = 0; r < nr; ++r) {
                double t = a[r];
                a[r] = b[r];
                b[r] = t;
            }
        }
        perm_push(&f->q, i, pc);
        if (pr != i) {
            perm_push(&f->p, i, pr);
            for (int64_t k = 0; k < nc; ++k) {
                double t = M[i + k * nr];
                M[i + k * nr] = M[pr + k * nr];
                M[pr + k * nr] = t;
            }
        }
        double *ci = M + i * nr;
        double inv_diag = 1.0 / diag;
        for (int64_t r = i + 1; r < nr; ++r) ci[r] *= inv_diag;
#pragma omp parallel for if (g_setup_threads > 1 && nc - i > 64) num_threads(g_setup_threads) schedule(static)
        for (int64_t k = i + 1; k < nc; ++k) {
            double *ck = M + k * nr;
            double a = -ck[i];
            for (int64_t r = i + 1; r < nr; ++r) ck[r] = a * ci[r] + ck[r];
        }
    }
}
static void fplu_free(fplu_t *f) {
    free(f->lu);
    perm_free(&f->p);
    perm_free(&f->q);
}

/* ------------------------------------------------------------------ PrimalPhase1  (primal_problem.rs:80-261) */

static void matvec_sub(const double *A, int64_t m, int64_t n, const double *v, const double *b,
                       double *out) {
    /* out = b - A v  (A*v accumulated column by column as nalgebra's gemv does) */
    double *Av = (double *)xcalloc((size_t)m, sizeof(double));
    for (int64_t j = 0; j < n; ++j) {
        double vj = v[j];
        if (vj == 0.0) continue;
        const double *cj = A + j * m;
        for (int64_t i = 0; i < m; ++i) Av[i] += cj[i] * vj;
    }
    for (int64_t i = 0; i < m; ++i) out[i] = b[i] - Av[i];
    free(Av);
}

eo_phase *eo_primal_phase1(const eo_problem *prob, int *err) {
    if (err) *err = 0;
    eo_phase *sf = standard_form(prob);
    if (!sf) return NULL;
    const int64_t n = sf->n, m = sf->m;

    int64_t *N = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(n + m + 1));
    uint8_t *Nb = (uint8_t *)xmalloc((size_t)(n + m + 1));
    int64_t *B = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(n + m + 1));
    int64_t nN = 0, nB = 0;
    double *v = (double *)xcalloc((size_t)(n + m + 1), sizeof(double));

    for (int64_t i = 0; i < n; ++i) {
        switch (sf->kind[i]) {
        case EO_FREE: break;
        case EO_LOWER: v[i] = sf->lb[i]; N[nN] = i; Nb[nN++] = EO_NB_LOWER; break;
        case EO_UPPER: v[i] = sf->ub[i]; N[nN] = i; Nb[nN++] = EO_NB_UPPER; break;
        case EO_TWOSIDED: v[i] = sf->lb[i]; N[nN] = i; Nb[nN++] = EO_NB_LOWER; break;
        case EO_FIXED: v[i] = sf->lb[i]; N[nN] = i; Nb[nN++] = EO_NB_LOWER; break;
        }
    }

    int64_t nfree = 0;
    int64_t *free_vars = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < n; ++i)
        if (sf->kind[i] == EO_FREE) free_vars[nfree++] = i;

    /* kind/lb/ub of the phase (copy; free vars beyond the rank become Fixed(0)) */
    uint8_t *kind = (uint8_t *)xmalloc((size_t)(n + m + 1));
    double *lb = (double *)xcalloc((size_t)(n + m + 1), sizeof(double));
    double *ub = (double *)xcalloc((size_t)(n + m + 1), sizeof(double));
    memcpy(kind, sf->kind, (size_t)n);
    memcpy(lb, sf->lb, sizeof(double) * (size_t)n);
    memcpy(ub, sf->ub, sizeof(double) * (size_t)n);

    eo_phase *ph = NULL;
    const int A_empty = (m == 0 || n == 0);

    if (nfree > 0 && !A_empty) {
        double *AF = (double *)xmalloc(sizeof(double) * (size_t)(m * nfree));
        for (int64_t k = 0; k < nfree; ++k)
            memcpy(AF + k * m, sf->A + free_vars[k] * m, sizeof(double) * (size_t)m);
        fplu_t f;
        fplu_factor(&f, AF, m, nfree);
        free(AF);
        const int64_t max_rank = m < nfree ? m : nfree;
        int64_t rank = nfree; /* unwrap_or_else(|| free_vars.len()) :175 */
        for (int64_t i = 0; i < max_rank; ++i) {
            if (fabs(f.lu[i + i * m]) < EPS) {
                rank = i;
                break;
            }
        }
        if (rank > max_rank) {
            /* the reference would panic slicing L/U (rank x rank) :202-205 */
            if (err) *err = EO_ERR_PANIC;
            fplu_free(&f);
            goto fail;
        }
        perm_rows_i(&f.q, free_vars);
        for (int64_t k = 0; k < rank; ++k) B[nB++] = free_vars[k];
        for (int64_t k = rank; k < nfree; ++k) {
            int64_t i = free_vars[k];
            kind[i] = EO_FIXED;
            lb[i] = 0.0;
            ub[i] = 0.0;
            N[nN] = i;
            Nb[nN++] = EO_NB_LOWER;
        }
        double *bt = (double *)xmalloc(sizeof(double) * (size_t)(m + 1));
        matvec_sub(sf->A, m, n, v, sf->b, bt);
        perm_rows_d(&f.p, bt);
        /* L (unit lower) y = bt[0..rank] ; U z = y, both column-oriented, rank x rank slices */
        for (int64_t i = 0; i < rank; ++i) {
            double coeff = bt[i] / 1.0;
            bt[i] = coeff;
            for (int64_t r = i + 1; r < rank; ++r) bt[r] = (-coeff) * f.lu[r + i * m] + bt[r];
        }
        for (int64_t i = rank - 1; i >= 0; --i) {
            double diag = f.lu[i + i * m];
            if (diag == 0.0) {
                if (err) *err = EO_ERR_PANIC; /* unwrap() on None :210 */
                free(bt);
                fplu_free(&f);
                goto fail;
            }
            double coeff = bt[i] / diag;
            bt[i] = coeff;
            for (int64_t r = 0; r < i; ++r) bt[r] = (-coeff) * f.lu[r + i * m] + bt[r];
        }
        for (int64_t k = 0; k < rank; ++k) v[free_vars[k]] = bt[k];

        int64_t *rows = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(m + 1));
        for (int64_t i = 0; i < m; ++i) rows[i] = i;
        perm_rows_i(&f.p, rows);
        const int64_t nart = m - rank;

        matvec_sub(sf->A, m, n, v, sf->b, bt); /* b_tilde with the free basics set :222 */

        ph = phase_alloc(m, n + nart, n + m);
        memcpy(ph->A, sf->A, sizeof(double) * (size_t)(m * n));
        int64_t cur_col = n + nart - 1;
        for (int64_t k = rank; k < m; ++k) {
            int64_t i = rows[k];
            v[cur_col] = fabs(bt[i]);
            ph->A[i + cur_col * m] = rust_signum(bt[i]);
            B[nB++] = cur_col;
            cur_col -= 1;
        }
        free(rows);
        free(bt);
        fplu_free(&f);
    } else {
        double *bt = (double *)xmalloc(sizeof(double) * (size_t)(m + 1));
        matvec_sub(sf->A, m, n, v, sf->b, bt);
        ph = phase_alloc(m, n + m, n + m);
        memcpy(ph->A, sf->A, sizeof(double) * (size_t)(m * n));
        for (int64_t i = 0; i < m; ++i) {
            int64_t index = n + i;
            v[index] = fabs(bt[i]);
            ph->A[i + index * m] = rust_signum(bt[i]);
            B[nB++] = index;
        }
        free(bt);
    }

    /* c = 0 on the first n, 1 on the m appended :137-141 */
    for (int64_t i = 0; i < n; ++i) ph->c[i] = 0.0;
    for (int64_t i = n; i < n + m; ++i) ph->c[i] = 1.0;
    memcpy(ph->b, sf->b, sizeof(double) * (size_t)m);
    memcpy(ph->kind, kind, (size_t)n);
    memcpy(ph->lb, lb, sizeof(double) * (size_t)n);
    memcpy(ph->ub, ub, sizeof(double) * (size_t)n);
    ph->n_p1vars = m;
    ph->p1vars = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(m + 1));
    for (int64_t i = 0; i < m; ++i) { /* :250-253 */
        ph->p1vars[i] = n + i;
        ph->kind[n + i] = EO_LOWER;
        ph->lb[n + i] = 0.0;
        ph->ub[n + i] = 0.0;
    }
    memcpy(ph->x, v, sizeof(double) * (size_t)(n + m));
    memcpy(ph->B, B, sizeof(int64_t) * (size_t)nB);
    memcpy(ph->N, N, sizeof(int64_t) * (size_t)nN);
    memcpy(ph->Nb, Nb, (size_t)nN);
    ph->nB = nB;
    ph->nN = nN;
    ph->which = 1;
    phase_copy_orig(ph, sf);

fail:
    free(N); free(Nb); free(B); free(v); free(free_vars); free(kind); free(lb); free(ub);
    eo_phase_free(sf);
    return ph;
}

static eo_phase *phase_clone_core(const eo_phase *s) {
    eo_phase *ph = phase_alloc(s->m, s->n, s->n_c);
    memcpy(ph->A, s->A, sizeof(double) * (size_t)(s->m * s->n));
    memcpy(ph->c, s->c, sizeof(double) * (size_t)s->n_c);
    memcpy(ph->b, s->b, sizeof(double) * (size_t)s->m);
    memcpy(ph->kind, s->kind, (size_t)s->n_c);
    memcpy(ph->lb, s->lb, sizeof(double) * (size_t)s->n_c);
    memcpy(ph->ub, s->ub, sizeof(double) * (size_t)s->n_c);
    memcpy(ph->x, s->x, sizeof(double) * (size_t)s->n_c);
    memcpy(ph->B, s->B, sizeof(int64_t) * (size_t)s->nB);
    memcpy(ph->N, s->N, sizeof(int64_t) * (size_t)s->nN);
    memcpy(ph->Nb, s->Nb, (size_t)s->nN);
    ph->nB = s->nB;
    ph->nN = s->nN;
    phase_copy_orig(ph, s);
    return ph;
}

/* primal_problem.rs:263-291 */
eo_phase *eo_primal_phase2(const eo_phase *p1) {
    eo_phase *ph = phase_clone_core(p1);
    for (int64_t k = 0; k < p1->n_p1vars; ++k) {
        int64_t i = p1->p1vars[k];
        ph->c[i] = 0.0;
        ph->kind[i] = EO_FIXED;
        ph->lb[i] = 0.0;
        ph->ub[i] = 0.0;
    }
    for (int64_t i = 0; i < p1->n_orig_vars; ++i) {
        ph->c[i] = p1->orig_obj[i];
        ph->kind[i] = p1->orig_kind[i];
        ph->lb[i] = p1->orig_lb[i];
        ph->ub[i] = p1->orig_ub[i];
    }
    for (int64_t k = 0; k < ph->nN; ++k)
        if (ph->kind[ph->N[k]] == EO_FREE) ph->Nb[k] = EO_NB_FREE;
    ph->which = 2;
    return ph;
}

/* ------------------------------------------------------------------ trivial solver
 * solvers/trivial/solve_trivial_problem.rs:5-96 (quirk Q8 kept) */
static int solve_trivial(int64_t n_c, const double *c, const uint8_t *kind, const double *lb,
                         const double *ub, double *x, int64_t *N, uint8_t *Nb, int64_t *nN,
                         int minimize) {
    int64_t k = 0;
    for (int64_t i = 0; i < n_c; ++i) {
        double ci = c[i];
        switch (kind[i]) {
        case EO_FREE:
            N[k] = i; Nb[k++] = EO_NB_FREE;
            if (ci != 0.0) { *nN = k; return EO_UNBOUNDED; }
            x[i] = 0.0;
            break;
        case EO_LOWER:
            N[k] = i; Nb[k++] = EO_NB_LOWER;
            if (ci > 0.0) {
                if (minimize) x[i] = lb[i];
                else { *nN = k; return EO_UNBOUNDED; }
            } else {
                if (minimize) x[i] = lb[i];
                else if (ci != 0.0) { *nN = k; return EO_UNBOUNDED; }
                else x[i] = lb[i];
            }
            break;
        case EO_UPPER:
            N[k] = i; Nb[k++] = EO_NB_UPPER;
            if (ci > 0.0) {
                if (minimize) { *nN = k; return EO_UNBOUNDED; }
                else x[i] = ub[i];
            } else {
                if (minimize) {
                    if (ci != 0.0) { *nN = k; return EO_UNBOUNDED; }
                    else x[i] = ub[i];
                } else x[i] = ub[i];
            }
            break;
        case EO_TWOSIDED:
            if ((ci > 0.0) == (minimize != 0)) {
                N[k] = i; Nb[k++] = EO_NB_LOWER; x[i] = lb[i];
            } else {
                N[k] = i; Nb[k++] = EO_NB_UPPER; x[i] = ub[i];
            }
            break;
        case EO_FIXED:
            N[k] = i; Nb[k++] = EO_NB_LOWER; x[i] = lb[i];
            break;
        }
    }
    *nN = k;
    return EO_OPTIMAL;
}

/* ------------------------------------------------------------------ primal hot loop
 * primal_simplex_solver.rs:95-436 */
int eo_primal_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                                 const double *c, const double *b, const uint8_t *kind,
                                 const double *lb, const double *ub, double *x, int64_t *B,
                                 int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN,
                                 uint64_t max_iter, uint64_t *iters_out, char *err,
                                 size_t errlen) {
    (void)b;
    if (iters_out) *iters_out = 0;
    if (m == 0) { /* :118-122 */
        if (nB != 0) {
            set_err(err, errlen, "assertion failed: B.is_empty()");
            return EO_ERR_PANIC;
        }
        int64_t k = 0;
        /* the caller's N buffer must hold n_c entries in this case */
        return solve_trivial(n_c, c, kind, lb, ub, x, N, Nb, &k, 1);
    }
    if (nB != m) { /* :124-130 */
        if (err && errlen)
            snprintf(err, errlen, "invalid B, has %lld elements but %lld expected", (long long)nB,
                     (long long)m);
        return EO_ERR_BAD_DIMS;
    }
    if (n < m) {
        set_err(err, errlen, "checked_sub overflow: cols < rows");
        return EO_ERR_PANIC;
    }
    if (nN != n - m) { /* :132-140 */
        if (err && errlen)
            snprintf(err, errlen, "invalid N, has %lld elements but %lld expected", (long long)nN,
                     (long long)(n - m));
        return EO_ERR_BAD_DIMS;
    }
    if (nN == 0) return EO_OPTIMAL; /* :149-151 */

    double *A_B = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *c_B = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *A_N = (double *)xmalloc(sizeof(double) * (size_t)(m * nN));
    double *c_N = (double *)xmalloc(sizeof(double) * (size_t)nN);
    for (int64_t i = 0; i < m; ++i) {
        memcpy(A_B + i * m, A + B[i] * m, sizeof(double) * (size_t)m);
        c_B[i] = c[B[i]];
    }
    for (int64_t j = 0; j < nN; ++j) {
        memcpy(A_N + j * m, A + N[j] * m, sizeof(double) * (size_t)m);
        c_N[j] = c[N[j]];
    }
    double *u = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *r = (double *)xmalloc(sizeof(double) * (size_t)nN);
    double *d = (double *)xmalloc(sizeof(double) * (size_t)m);
    lu_t f;
    f.lu = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    f.nr = f.nc = m;
    perm_init(&f.p, m + 1);

    double *gamma = NULL, *se_rho = NULL, *se_v = NULL;
    if (g_primal_rule == 1 || g_primal_rule == 2) {
        gamma = (double *)xmalloc(sizeof(double) * (size_t)(nN > 0 ? nN : 1));
        se_rho = (double *)xmalloc(sizeof(double) * (size_t)m);
        se_v = (double *)xmalloc(sizeof(double) * (size_t)m);
        int perm = 1; /* is A_B a signed permutation? */
        for (int64_t i = 0; i < m && perm; ++i) {
            int cnt = 0;
            for (int64_t k2 = 0; k2 < m; ++k2) {
                const double a = A_B[i * m + k2];
                if (a != 0.0) {
                    cnt += 1;
                    if (fabs(a) != 1.0) perm = 0;
                }
            }
            if (cnt != 1) perm = 0;
        }
        int have_lu = 0;
        if (!perm && g_primal_rule == 1) { /* a general starting basis: exact weights 1 + |B^-1 a_j|^2 from one factorisation */
            memcpy(f.lu, A_B, sizeof(double) * (size_t)(m * m));
            f.p.len = 0;
            lu_factor_inplace(&f);
            have_lu = 1;
        }
        for (int64_t j = 0; j < nN; ++j) {
            double g = 1.0;
            const double *cj = A_N + j * m;
            if (perm && g_primal_rule == 1) {
                for (int64_t i = 0; i < m; ++i) g += cj[i] * cj[i];
            } else if (have_lu) {
                memcpy(se_v, cj, sizeof(double) * (size_t)m);
                if (lu_solve(&f, se_v))
                    for (int64_t i = 0; i < m; ++i) g += se_v[i] * se_v[i];
            }
            gamma[j] = g;
        }
        if (have_lu) memset(se_v, 0, sizeof(double) * (size_t)m);
    }
    int status = EO_ERR_PANIC;
    uint64_t iter = 1;
    uint64_t entered = 0;
    /* partial pricing (see g_partial_segments): the same normalisation of P and S as the engine's */
    int pp_P = 0, pp_seg = 0, pp_empty = 0;
    int64_t pp_S = nN;
    if (g_partial_segments > 1 && nN > 0) {
        int P = g_partial_segments;
        if ((int64_t)P > nN) P = (int)nN;
        pp_S = (nN + P - 1) / P;
        P = (int)((nN + pp_S - 1) / pp_S);
        if (P > 1) pp_P = P;
        else pp_S = nN;
    }
    for (;;) {
        if (iter > max_iter) { /* :163-166 */
            status = EO_MAXITER;
            break;
        }
        iter += 1;
        entered += 1;

        /* :173  lu = A_B.clone().lu() */
        memcpy(f.lu, A_B, sizeof(double) * (size_t)(m * m));
        f.p.len = 0;
        lu_factor_inplace(&f);
        if (lu_small_diag(&f)) { /* :175-179 */
            set_err(err, errlen, "invalid B, A_B is not invertible");
            status = EO_ERR_SINGULAR;
            break;
        }
        /* :184-187 BTRAN */
        memcpy(u, c_B, sizeof(double) * (size_t)m);
        if (!lu_btran(&f, u)) {
            set_err(err, errlen, "unwrap() on None in BTRAN");
            status = EO_ERR_PANIC;
            break;
        }
        /* :189 pricing r = c_N - A_N^T u */
        for (int64_t j = 0; j < nN; ++j) {
            const double *cj = A_N + j * m;
            double dot = 0.0;
            for (int64_t i = 0; i < m; ++i) dot += cj[i] * u[i];
            r[j] = c_N[j] - dot;
        }

        /* ---- pivot() :238-435 ---- entering: sequential max_by fold :253-287 */
        int have = 0;
        double r1 = 0.0;
        int64_t q = -1;
        const int64_t pp_lo = pp_P > 1 ? (int64_t)pp_seg * pp_S : 0;
        const int64_t pp_hi = pp_P > 1 ? (pp_lo + pp_S < nN ? pp_lo + pp_S : nN) : nN;
        for (int64_t j = pp_lo; j < pp_hi; ++j) {
            double rj = r[j];
            if (fabs(rj) < EPS) continue;
            double key;
            int pos = rj > 0.0;
            if (pos && Nb[j] == EO_NB_UPPER) key = rj;
            else if (!pos && Nb[j] == EO_NB_LOWER) key = -rj;
            else if (Nb[j] == EO_NB_FREE) key = fabs(rj);
            else continue;
            if (gamma) key = fabs(rj) / sqrt(gamma[j]); /* extension: steepest edge.  |r_j| / sqrt(gamma_j), not its square: the fold
                                                           * compares keys to within EPS, and squares of small reduced costs
                                                           * all fall below it long before the optimum */
            if (!have) {
                have = 1;
                r1 = key;
                q = j;
                continue;
            }
            /* max_by keeps acc only if compare(acc, new) == Greater */
            int acc_greater;
            if (fabs(r1 - key) >= EPS) {
                /* partial_cmp(..).expect("NaN detected") can never see a NaN here: a NaN
                 * operand makes the '>= EPS' test false and falls to the index rule. */
                acc_greater = r1 > key;
            } else {
                acc_greater = N[q] > N[j];
            }
            if (!acc_greater) {
                r1 = key;
                q = j;
            }
        }
        /* No "NaN detected" exit here: the expect() of :282 sits behind the '>= EPS' test, which is false for a NaN
         * operand, so the reference orders a NaN key by variable index and goes on (the engine stops with
         * ELLP_ERR_NAN instead — a documented deviation, pinned by tests/test_gpu_abi_edge.py). */
        if (!have) { /* :289-292 */
            if (pp_P > 1 && pp_empty + 1 < pp_P) { /* nothing in this segment: on to the next one */
                pp_seg = (pp_seg + 1) % pp_P;
                pp_empty += 1;
                continue;
            }
            status = EO_OPTIMAL;
            break;
        }
        pp_empty = 0;
        const int64_t jq = N[q];
        /* :295-300 FTRAN */
        memcpy(d, A + jq * m, sizeof(double) * (size_t)m);
        if (!lu_solve(&f, d)) {
            set_err(err, errlen, "unwrap() on None in FTRAN");
            status = EO_ERR_PANIC;
            break;
        }
        const int at_lower = (Nb[q] == EO_NB_LOWER);
        if (at_lower)
            for (int64_t i = 0; i < m; ++i) d[i] = -d[i];

        double lambda; /* :305-311 */
        switch (kind[jq]) {
        case EO_TWOSIDED: lambda = ub[jq] - lb[jq]; break;
        case EO_FIXED: lambda = 0.0; break;
        default: lambda = INFINITY; break;
        }
        int64_t new_basic = -1;
        int new_side = EO_NB_LOWER;
        int have_nbi = 0;
        int64_t nbi = 0;
        for (int64_t i = 0; i < m; ++i) { /* :320-400 */
            double di = d[i];
            if (fabs(di) < EPS) continue;
            const int64_t bi = B[i];
            double xi = x[bi];
            double li;
            switch (kind[bi]) {
            case EO_FREE: li = INFINITY; break;
            case EO_LOWER:
                if (di > 0.0) li = INFINITY;
                else if (xi > lb[bi]) li = (lb[bi] - xi) / di;
                else li = 0.0;
                break;
            case EO_UPPER:
                if (di > 0.0) li = (xi < ub[bi]) ? (ub[bi] - xi) / di : 0.0;
                else li = INFINITY;
                break;
            case EO_TWOSIDED:
                if (di > 0.0) li = (xi < ub[bi]) ? (ub[bi] - xi) / di : 0.0;
                else if (xi < lb[bi]) li = (lb[bi] - xi) / di; /* quirk Q1 :359 */
                else li = 0.0;
                break;
            default: li = 0.0; break; /* Fixed */
            }
            if (li < lambda - EPS) {
                lambda = li;
                new_basic = i;
                new_side = (di > 0.0) ? EO_NB_UPPER : EO_NB_LOWER;
            } else if (fabs(li - lambda) < EPS) {
                if (!have_nbi || bi < nbi) {
                    have_nbi = 1;
                    nbi = bi;
                    lambda = li;
                    new_basic = i;
                    new_side = (di > 0.0) ? EO_NB_UPPER : EO_NB_LOWER;
                }
            }
        }
        if (!(lambda >= 0.0)) { /* :402 */
            set_err(err, errlen, "assertion failed: lambda >= 0.");
            status = EO_ERR_PANIC;
            break;
        }
        if (isinf(lambda)) { /* :404-406 */
            status = EO_UNBOUNDED;
            break;
        }
        if (lambda > 0.0) { /* :408-417 */
            for (int64_t i = 0; i < m; ++i) x[B[i]] += lambda * d[i];
            if (at_lower) x[jq] += lambda;
            else x[jq] -= lambda;
        }
        if (g_trace)
            g_trace(g_trace_user, entered, q, new_basic, jq, new_basic >= 0 ? B[new_basic] : -1);
        if (gamma && new_basic >= 0) { /* extension: steepest-edge weights, with the OLD basis' factors */
            const int64_t rr = new_basic;
            const double arq = at_lower ? -d[rr] : d[rr];
            memset(se_rho, 0, sizeof(double) * (size_t)m);
            se_rho[rr] = 1.0;
            for (int64_t i = 0; i < m; ++i) se_v[i] = at_lower ? -d[i] : d[i];
            if (!lu_btran(&f, se_rho) || !lu_btran(&f, se_v)) {
                set_err(err, errlen, "unwrap() on None in the steepest-edge BTRAN");
                status = EO_ERR_PANIC;
                break;
            }
            /* the entering variable's weight is taken EXACTLY, 1 + |alpha_q|^2 (alpha_q is in hand): an error in the
             * stored gamma_q would spread to every other weight through abar_j^2 gamma_q and from there into the next
             * gamma_q — measured on the engine, whose inverse is only good to 1e-12: a factor 1.8 per iteration */
            double gq = 1.0;
            for (int64_t i = 0; i < m; ++i) gq += d[i] * d[i];
            if (g_primal_rule == 2) gq = gamma[q];
            for (int64_t j = 0; j < nN; ++j) {
                if (j == q) continue;
                const double *cj = A_N + j * m;
                double ar = 0.0, tv = 0.0;
                for (int64_t i = 0; i < m; ++i) {
                    ar += cj[i] * se_rho[i];
                    tv += cj[i] * se_v[i];
                }
                const double ab = ar / arq;
                if (g_primal_rule == 2) { /* Devex: the reference-framework estimate only ever grows */
                    const double g = ab * ab * gq;
                    if (g > gamma[j]) gamma[j] = g;
                    continue;
                }
                double g = gamma[j] - 2.0 * ab * tv + ab * ab * gq;
                const double lo = 1.0 + ab * ab;
                gamma[j] = g > lo ? g : lo;
            }
            const double gl = gq / (arq * arq);
            gamma[q] = gl > 1.0 ? gl : 1.0;
        }
        /* :205-232 apply the pivot */
        if (new_basic >= 0) {
            int64_t t = B[new_basic];
            B[new_basic] = N[q];
            N[q] = t;
            double *cn = A_N + q * m, *cb = A_B + new_basic * m;
            for (int64_t i = 0; i < m; ++i) {
                double tt = cn[i];
                cn[i] = cb[i];
                cb[i] = tt;
            }
            double tc = c_N[q];
            c_N[q] = c_B[new_basic];
            c_B[new_basic] = tc;
            Nb[q] = (uint8_t)new_side;
        } else {
            if (Nb[q] == EO_NB_LOWER) Nb[q] = EO_NB_UPPER;
            else if (Nb[q] == EO_NB_UPPER) Nb[q] = EO_NB_LOWER;
            else {
                set_err(err, errlen, "pivot should have been unbounded");
                status = EO_ERR_PANIC;
                break;
            }
        }
    }
    if (iters_out) *iters_out = entered;
    lu_free(&f);
    free(A_B); free(c_B); free(A_N); free(c_N); free(u); free(r); free(d); free(gamma); free(se_rho); free(se_v);
    return status;
}

/* ------------------------------------------------------------------ primal loop, explicit B^-1
 * NOT the reference's linear algebra: the same pivoting rules as the loop above
 * (primal_simplex_solver.rs:253-292 entering fold, :305-400 ratio test, :205-232 swap), but
 * B^-1 is kept explicitly and updated by the rank-1 (eta) formula, as the HIP engine does,
 * and the O(m|N|) / O(m^2) passes run on `threads` OpenMP threads.  Two uses, both test
 * infrastructure: (1) the "same algorithm on the host cores" CPU baseline that bench.py reports
 * next to the reference-algorithm baseline; (2) pivot-sequence parity with
 * the engine over windows of thousands of pivots at full size, where the LU-per-iteration loop
 * would take hours.  It is itself checked against the loop above (tests/test_oracle_binv.py).
 * refresh > 0: B^-1 is recomputed from an LU of A_B every `refresh` basis changes. */
#include <time.h>

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* four independent partial sums: the order of additions differs from the reference loop's
 * single chain (last-bit differences only), and the adds no longer wait on each other */
static double dot4(const double *a, const double *v, int64_t n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        s0 += a[i] * v[i];
        s1 += a[i + 1] * v[i + 1];
        s2 += a[i + 2] * v[i + 2];
        s3 += a[i + 3] * v[i + 3];
    }
    for (; i < n; ++i) s0 += a[i] * v[i];
    return (s0 + s1) + (s2 + s3);
}

/* W (row-major m x m) = A_B^-1 ; 0 if A_B fails the reference's singularity guard (:175-179) */
static int binv_from_lu(const double *A_B, int64_t m, double *W, int threads) {
    lu_t f;
    lu_factor(&f, A_B, m, m);
    int ok = !lu_small_diag(&f);
    if (ok) {
        int bad = 0;
        (void)threads;
#pragma omp parallel num_threads(threads)
        {
            double *e = (double *)xmalloc(sizeof(double) * (size_t)m);
#pragma omp for schedule(static)
            for (int64_t k = 0; k < m; ++k) {
                memset(e, 0, sizeof(double) * (size_t)m);
                e[k] = 1.0;
                if (!lu_solve(&f, e)) {
#pragma omp atomic write
                    bad = 1;
                }
                for (int64_t i = 0; i < m; ++i) W[i * m + k] = e[i];
            }
            free(e);
        }
        if (bad) ok = 0;
    }
    lu_free(&f);
    return ok;
}

int eo_primal_binv_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                                      const double *c, const double *b, const uint8_t *kind,
                                      const double *lb, const double *ub, double *x, int64_t *B,
                                      int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN,
                                      uint64_t max_iter, uint64_t *iters_out, int threads,
                                      int refresh, double *loop_seconds, char *err, size_t errlen) {
    (void)b;
    (void)n_c;
    if (iters_out) *iters_out = 0;
    if (loop_seconds) *loop_seconds = 0.0;
    if (m <= 0 || nB != m || n < m || nN != n - m) {
        set_err(err, errlen, "bad dimensions");
        return EO_ERR_BAD_DIMS;
    }
    if (nN == 0) return EO_OPTIMAL;
    if (threads < 1) threads = 1;

    double *A_B = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *c_B = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *A_N = (double *)xmalloc(sizeof(double) * (size_t)(m * nN));
    double *c_N = (double *)xmalloc(sizeof(double) * (size_t)nN);
    for (int64_t i = 0; i < m; ++i) {
        memcpy(A_B + i * m, A + B[i] * m, sizeof(double) * (size_t)m);
        c_B[i] = c[B[i]];
    }
    for (int64_t j = 0; j < nN; ++j) {
        memcpy(A_N + j * m, A + N[j] * m, sizeof(double) * (size_t)m);
        c_N[j] = c[N[j]];
    }
    double *W = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *u = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *r = (double *)xmalloc(sizeof(double) * (size_t)nN);
    double *d = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *rho = (double *)xmalloc(sizeof(double) * (size_t)m);

    int status = EO_ERR_PANIC;
    uint64_t iter = 1, entered = 0, since_refresh = 0, since_btran = 0;
    int u_valid = 0;
    if (!binv_from_lu(A_B, m, W, threads)) {
        set_err(err, errlen, "invalid B, A_B is not invertible");
        status = EO_ERR_SINGULAR;
        goto done;
    }
    const double t0 = now_seconds();
    for (;;) {
        if (iter > max_iter) {
            status = EO_MAXITER;
            break;
        }
        iter += 1;
        entered += 1;
        if (refresh > 0 && since_refresh >= (uint64_t)refresh) {
            if (!binv_from_lu(A_B, m, W, threads)) {
                set_err(err, errlen, "invalid B, A_B is not invertible");
                status = EO_ERR_SINGULAR;
                break;
            }
            since_refresh = 0;
            u_valid = 0;
        }
        if (!u_valid || since_btran >= 32) { /* u = B^-T c_B in full */
#pragma omp parallel for schedule(static) num_threads(threads)
            for (int64_t k = 0; k < m; ++k) {
                double acc = 0.0;
                for (int64_t i = 0; i < m; ++i) acc += c_B[i] * W[i * m + k];
                u[k] = acc;
            }
            u_valid = 1;
            since_btran = 0;
        }
        /* :189 pricing */
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t j = 0; j < nN; ++j) {
            r[j] = c_N[j] - dot4(A_N + j * m, u, m);
        }
        /* entering: the sequential max_by fold :253-287 */
        int have = 0;
        double r1 = 0.0;
        int64_t q = -1;
        for (int64_t j = 0; j < nN; ++j) {
            double rj = r[j];
            if (rj != rj) {
                set_err(err, errlen, "NaN detected");
                status = EO_ERR_NAN;
                goto timed_done;
            }
            if (fabs(rj) < EPS) continue;
            double key;
            int pos = rj > 0.0;
            if (pos && Nb[j] == EO_NB_UPPER) key = rj;
            else if (!pos && Nb[j] == EO_NB_LOWER) key = -rj;
            else if (Nb[j] == EO_NB_FREE) key = fabs(rj);
            else continue;
            if (!have) {
                have = 1;
                r1 = key;
                q = j;
                continue;
            }
            int acc_greater = (fabs(r1 - key) >= EPS) ? (r1 > key) : (N[q] > N[j]);
            if (!acc_greater) {
                r1 = key;
                q = j;
            }
        }
        if (!have) {
            status = EO_OPTIMAL;
            break;
        }
        const int64_t jq = N[q];
        const int at_lower = (Nb[q] == EO_NB_LOWER);
        /* FTRAN d = +-B^-1 a_q */
        {
            const double *aq = A_N + q * m;
            const double sg = at_lower ? -1.0 : 1.0;
#pragma omp parallel for schedule(static) num_threads(threads)
            for (int64_t i = 0; i < m; ++i) {
                d[i] = sg * dot4(W + i * m, aq, m);
                if (fabs(d[i]) < g_zero_tol) d[i] = 0.0;
            }
        }
        double lambda;
        switch (kind[jq]) {
        case EO_TWOSIDED: lambda = ub[jq] - lb[jq]; break;
        case EO_FIXED: lambda = 0.0; break;
        default: lambda = INFINITY; break;
        }
        int64_t new_basic = -1;
        int new_side = EO_NB_LOWER;
        int have_nbi = 0;
        int64_t nbi = 0;
        for (int64_t i = 0; i < m; ++i) { /* :320-400 */
            double di = d[i];
            if (fabs(di) < EPS) continue;
            const int64_t bi = B[i];
            double xi = x[bi];
            double li;
            switch (kind[bi]) {
            case EO_FREE: li = INFINITY; break;
            case EO_LOWER:
                if (di > 0.0) li = INFINITY;
                else if (xi > lb[bi]) li = (lb[bi] - xi) / di;
                else li = 0.0;
                break;
            case EO_UPPER:
                if (di > 0.0) li = (xi < ub[bi]) ? (ub[bi] - xi) / di : 0.0;
                else li = INFINITY;
                break;
            case EO_TWOSIDED:
                if (di > 0.0) li = (xi < ub[bi]) ? (ub[bi] - xi) / di : 0.0;
                else if (xi < lb[bi]) li = (lb[bi] - xi) / di; /* quirk Q1 :359 */
                else li = 0.0;
                break;
            default: li = 0.0; break;
            }
            if (li < lambda - EPS) {
                lambda = li;
                new_basic = i;
                new_side = (di > 0.0) ? EO_NB_UPPER : EO_NB_LOWER;
            } else if (fabs(li - lambda) < EPS) {
                if (!have_nbi || bi < nbi) {
                    have_nbi = 1;
                    nbi = bi;
                    lambda = li;
                    new_basic = i;
                    new_side = (di > 0.0) ? EO_NB_UPPER : EO_NB_LOWER;
                }
            }
        }
        if (!(lambda >= 0.0)) {
            set_err(err, errlen, "assertion failed: lambda >= 0.");
            status = EO_ERR_PANIC;
            break;
        }
        if (isinf(lambda)) {
            status = EO_UNBOUNDED;
            break;
        }
        if (new_basic >= 0 && guard_trips(d[new_basic], d, m)) { /* certified hybrid: nothing of this iteration is committed */
            status = EO_NEED_EXACT;
            entered -= 1;
            break;
        }
        if (lambda > 0.0) {
            for (int64_t i = 0; i < m; ++i) x[B[i]] += lambda * d[i];
            if (at_lower) x[jq] += lambda;
            else x[jq] -= lambda;
        }
        if (g_trace)
            g_trace(g_trace_user, entered, q, new_basic, jq, new_basic >= 0 ? B[new_basic] : -1);
        if (new_basic >= 0) {
            const int64_t rr = new_basic;
            const double d_r = d[rr];
            const double alpha_r = at_lower ? -d_r : d_r;
            memcpy(rho, W + rr * m, sizeof(double) * (size_t)m);
            { /* u += (r_q / alpha_r) * row_r(B^-1) */
                const double cf = r[q] / alpha_r;
                for (int64_t k = 0; k < m; ++k) u[k] = cf * rho[k] + u[k];
                since_btran += 1;
            }
            /* B^-1 <- E B^-1 : row_i -= (d_i/d_r) row_r ; row_r /= alpha_r */
#pragma omp parallel for schedule(static) num_threads(threads)
            for (int64_t i = 0; i < m; ++i) {
                double *wi = W + i * m;
                if (i == rr) {
                    for (int64_t k = 0; k < m; ++k) wi[k] = rho[k] / alpha_r;
                } else {
                    const double f = -(d[i] / d_r);
                    if (f != 0.0)
                        for (int64_t k = 0; k < m; ++k) wi[k] = f * rho[k] + wi[k];
                }
            }
            int64_t t = B[rr];
            B[rr] = N[q];
            N[q] = t;
            double *cn = A_N + q * m, *cb = A_B + rr * m;
            for (int64_t i = 0; i < m; ++i) {
                double tt = cn[i];
                cn[i] = cb[i];
                cb[i] = tt;
            }
            double tc = c_N[q];
            c_N[q] = c_B[rr];
            c_B[rr] = tc;
            Nb[q] = (uint8_t)new_side;
            since_refresh += 1;
        } else {
            if (Nb[q] == EO_NB_LOWER) Nb[q] = EO_NB_UPPER;
            else if (Nb[q] == EO_NB_UPPER) Nb[q] = EO_NB_LOWER;
            else {
                set_err(err, errlen, "pivot should have been unbounded");
                status = EO_ERR_PANIC;
                break;
            }
        }
    }
timed_done:
    if (loop_seconds) *loop_seconds = now_seconds() - t0;
done:
    if (iters_out) *iters_out = entered;
    free(A_B); free(c_B); free(A_N); free(c_N); free(W); free(u); free(r); free(d); free(rho);
    return status;
}

/* ------------------------------------------------------------------ dual hot loop
 * dual_simplex_solver.rs:110-335 */
int eo_dual_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                               const double *c, const double *b, const uint8_t *kind,
                               const double *lb, const double *ub, double *x, int64_t *B,
                               int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN, double *y,
                               double *d, uint64_t max_iter, uint64_t *iters_out, char *err,
                               size_t errlen) {
    if (iters_out) *iters_out = 0;
    if (m == 0) { /* :132-136 */
        if (nB != 0) {
            set_err(err, errlen, "assertion failed: B.is_empty()");
            return EO_ERR_PANIC;
        }
        int64_t k = 0;
        return solve_trivial(n_c, c, kind, lb, ub, x, N, Nb, &k, 1);
    }
    for (int64_t j = 0; j < nN && !g_continuation; ++j) { /* :139-151 */
        double di = d[N[j]];
        int infeasible;
        if (Nb[j] == EO_NB_LOWER) infeasible = di < -EPS;
        else if (Nb[j] == EO_NB_UPPER) infeasible = di > EPS;
        else infeasible = fabs(di) > EPS;
        if (infeasible) {
            set_err(err, errlen, "initial point of dual phase 2 is dual infeasible");
            return EO_ERR_PANIC;
        }
    }
    if (nB != m) {
        if (err && errlen)
            snprintf(err, errlen, "invalid B, has %lld elements but %lld expected", (long long)nB,
                     (long long)m);
        return EO_ERR_BAD_DIMS;
    }
    if (n < m) {
        set_err(err, errlen, "checked_sub overflow: cols < rows");
        return EO_ERR_PANIC;
    }
    if (nN != n - m) {
        if (err && errlen)
            snprintf(err, errlen, "invalid N, has %lld elements but %lld expected", (long long)nN,
                     (long long)(n - m));
        return EO_ERR_BAD_DIMS;
    }
    if (nN == 0) return EO_OPTIMAL; /* :175-177 */

    double *A_B = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *A_N = (double *)xmalloc(sizeof(double) * (size_t)(m * nN));
    for (int64_t i = 0; i < m; ++i)
        memcpy(A_B + i * m, A + B[i] * m, sizeof(double) * (size_t)m);
    for (int64_t j = 0; j < nN; ++j)
        memcpy(A_N + j * m, A + N[j] * m, sizeof(double) * (size_t)m);
    double *rho = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *alpha = (double *)xmalloc(sizeof(double) * (size_t)nN);
    double *alpha_q = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *flipcol = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *bf_ratio = (double *)xmalloc(sizeof(double) * (size_t)(nN + 1));
    int64_t *bf_pos = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(nN + 1));
    lu_t f;
    f.lu = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    f.nr = f.nc = m;
    perm_init(&f.p, m + 1);

    int status = EO_ERR_PANIC;
    uint64_t iter = 0, entered = 0;
    double obj = dual_obj(m, n_c, b, kind, lb, ub, y, d); /* :184 */
    (void)obj;
    for (;;) {
        if (iter >= max_iter) { /* :191-194 */
            status = EO_MAXITER;
            break;
        }
        iter += 1;
        entered += 1;
        /* leaving: first violated basic in B order :200-236 */
        int64_t r = -1;
        double delta = 0.0;
        int side = EO_NB_LOWER;
        for (int64_t i = 0; i < m && (r < 0 || (g_dual_rule & 2)); ++i) {
            const int64_t bi = B[i];
            double xi = x[bi];
            int64_t ri = -1;
            double di = 0.0;
            int si = EO_NB_LOWER;
            switch (kind[bi]) {
            case EO_LOWER:
                if (xi < lb[bi] - EPS) { ri = i; di = xi - lb[bi]; si = EO_NB_LOWER; }
                break;
            case EO_UPPER:
                if (xi > ub[bi] + EPS) { ri = i; di = xi - ub[bi]; si = EO_NB_UPPER; }
                break;
            case EO_TWOSIDED:
                if (xi > ub[bi] + EPS) { ri = i; di = xi - ub[bi]; si = EO_NB_UPPER; }
                else if (xi < lb[bi] - EPS) { ri = i; di = xi - lb[bi]; si = EO_NB_LOWER; }
                break;
            default: break; /* Free, Fixed never leave (quirk Q3) */
            }
            if (ri >= 0 && (r < 0 || fabs(di) > fabs(delta))) { /* extension bit 1: the largest violation, first of equals */
                r = ri;
                delta = di;
                side = si;
            }
        }
        /* :241 LU before the optimality test (quirk Q4) */
        memcpy(f.lu, A_B, sizeof(double) * (size_t)(m * m));
        f.p.len = 0;
        lu_factor_inplace(&f);
        if (r < 0) { /* :243-246 */
            status = EO_OPTIMAL;
            break;
        }
        /* :248-253 rho = row r of A_B^{-1} */
        memset(rho, 0, sizeof(double) * (size_t)m);
        rho[r] = 1.0;
        if (!lu_btran(&f, rho)) {
            set_err(err, errlen, "unwrap() on None in dual BTRAN");
            status = EO_ERR_PANIC;
            break;
        }
        /* :255-259 alpha = A_N^T rho, negated if delta < 0 */
        for (int64_t j = 0; j < nN; ++j) {
            const double *cj = A_N + j * m;
            double dot = 0.0;
            for (int64_t i = 0; i < m; ++i) dot += cj[i] * rho[i];
            alpha[j] = (delta < 0.0) ? -dot : dot;
        }
        /* :263-279 entering: first minimum of d/alpha over eligible */
        int64_t q = -1;
        double theta_dual = 0.0;
        int nan_seen = 0;
        for (int64_t j = 0; j < nN; ++j) {
            int keep;
            if (Nb[j] == EO_NB_LOWER) keep = alpha[j] > EPS;
            else if (Nb[j] == EO_NB_UPPER) keep = alpha[j] < -EPS;
            else keep = 1;
            if (!keep) continue;
            double ratio = d[N[j]] / alpha[j];
            if (q < 0) {
                q = j;
                theta_dual = ratio;
            } else {
                if (isnan(ratio) || isnan(theta_dual)) {
                    nan_seen = 1;
                    break;
                }
                /* min_by keeps acc unless compare(acc,new) == Greater */
                if (theta_dual > ratio) {
                    q = j;
                    theta_dual = ratio;
                }
            }
        }
        if (nan_seen) {
            set_err(err, errlen, "unwrap() on None: NaN in dual ratio test");
            status = EO_ERR_NAN;
            break;
        }
        if (q < 0) { /* :281-284 dual unbounded */
            status = EO_INFEASIBLE;
            break;
        }
        int64_t nflip = 0;
        if (g_dual_rule & 1) {
            /* extension bit 0: walk the breakpoints in (ratio, position) order */
            int64_t nc = 0;
            for (int64_t j = 0; j < nN; ++j) {
                int keep;
                if (Nb[j] == EO_NB_LOWER) keep = alpha[j] > EPS;
                else if (Nb[j] == EO_NB_UPPER) keep = alpha[j] < -EPS;
                else keep = 1;
                if (keep) bf_pos[nc++] = j;
            }
            for (int64_t k = 0; k < nc; ++k) bf_ratio[k] = d[N[bf_pos[k]]] / alpha[bf_pos[k]];
            /* insertion into sorted order by (ratio, position) as far as needed: selection, one breakpoint at a time */
            double slope = fabs(delta);
            int64_t taken = 0;
            for (;;) {
                int64_t best = -1;
                for (int64_t k = taken; k < nc; ++k)
                    if (best < 0 || bf_ratio[k] < bf_ratio[best] || (bf_ratio[k] == bf_ratio[best] && bf_pos[k] < bf_pos[best])) best = k;
                /* move it to slot `taken` */
                const int64_t pj = bf_pos[best];
                const double pr = bf_ratio[best];
                bf_pos[best] = bf_pos[taken];
                bf_ratio[best] = bf_ratio[taken];
                bf_pos[taken] = pj;
                bf_ratio[taken] = pr;
                const int64_t vj = N[pj];
                const int boxed = kind[vj] == EO_TWOSIDED;
                const double drop = boxed ? fabs(alpha[pj]) * (ub[vj] - lb[vj]) : INFINITY;
                if (!boxed || !(slope - drop > 0.0) || taken + 1 == nc) { /* this one enters */
                    q = pj;
                    theta_dual = pr;
                    break;
                }
                slope -= drop;
                taken += 1;
            }
            nflip = taken;
        }
        if (delta < 0.0) { /* :286-289 */
            for (int64_t j = 0; j < nN; ++j) alpha[j] = -alpha[j];
            theta_dual = -theta_dual;
        }
        const int64_t leaving_var = B[r];
        const int64_t entering_var = N[q];
        if (nflip > 0) {
            /* the passed variables change bound: x_N moves, x_B follows by B^-1 (sum of a_j dx_j); the dual objective
             * gains what the passed stretches contribute (recomputed below from scratch would be the same) */
            memset(flipcol, 0, sizeof(double) * (size_t)m);
            for (int64_t k = 0; k < nflip; ++k) {
                const int64_t pj = bf_pos[k];
                const int64_t vj = N[pj];
                const double dx = (Nb[pj] == EO_NB_LOWER) ? (ub[vj] - lb[vj]) : (lb[vj] - ub[vj]);
                const double *cj = A_N + pj * m;
                for (int64_t i = 0; i < m; ++i) flipcol[i] += cj[i] * dx;
                x[vj] = (Nb[pj] == EO_NB_LOWER) ? ub[vj] : lb[vj];
                Nb[pj] = (Nb[pj] == EO_NB_LOWER) ? EO_NB_UPPER : EO_NB_LOWER;
            }
            if (!lu_solve(&f, flipcol)) {
                set_err(err, errlen, "unwrap() on None in the bound-flip FTRAN");
                status = EO_ERR_PANIC;
                break;
            }
            for (int64_t i = 0; i < m; ++i) x[B[i]] -= flipcol[i];
            /* the leaving row's violation after the flips */
            {
                const int64_t bi = B[r];
                delta = (side == EO_NB_UPPER) ? x[bi] - ub[bi] : x[bi] - lb[bi];
            }
        }
        memcpy(alpha_q, A + entering_var * m, sizeof(double) * (size_t)m);
        if (!lu_solve(&f, alpha_q)) { /* :294 */
            set_err(err, errlen, "unwrap() on None in dual FTRAN");
            status = EO_ERR_PANIC;
            break;
        }
        d[leaving_var] = -theta_dual; /* :296-304 */
        for (int64_t j = 0; j < nN; ++j) d[N[j]] -= theta_dual * alpha[j];
        d[entering_var] = 0.0;
        for (int64_t i = 0; i < m; ++i) y[i] += theta_dual * rho[i];
        double theta_primal = delta / alpha_q[r]; /* :306-316 */
        for (int64_t i = 0; i < m; ++i) x[B[i]] -= theta_primal * alpha_q[i];
        x[entering_var] += theta_primal;
        obj += theta_dual * delta;
        if (g_trace) g_trace(g_trace_user, entered, q, r, entering_var, leaving_var);
        /* :322-333 */
        B[r] = entering_var;
        N[q] = leaving_var;
        Nb[q] = (uint8_t)side;
        double *cb = A_B + r * m, *cn = A_N + q * m;
        for (int64_t i = 0; i < m; ++i) {
            double t = cb[i];
            cb[i] = cn[i];
            cn[i] = t;
        }
    }
    if (iters_out) *iters_out = entered;
    lu_free(&f);
    free(A_B); free(A_N); free(rho); free(alpha); free(alpha_q); free(flipcol); free(bf_ratio); free(bf_pos);
    return status;
}

/* ------------------------------------------------------------------ dual loop, explicit B^-1
 * As eo_primal_binv_solve_with_initial: the reference's dual pivot rules (dual…:200-333) with B^-1
 * kept explicitly (eta updates, what the HIP engine does) and the big passes on OpenMP threads.
 * Checked against eo_dual_solve_with_initial in tests/test_oracle_binv.py; used for long dual
 * windows at full size. */
int eo_dual_binv_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A,
                                    const double *c, const double *b, const uint8_t *kind,
                                    const double *lb, const double *ub, double *x, int64_t *B,
                                    int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN, double *y,
                                    double *d, uint64_t max_iter, uint64_t *iters_out, int threads,
                                    int refresh, double *loop_seconds, char *err, size_t errlen) {
    (void)c;
    (void)b;
    (void)n_c;
    if (iters_out) *iters_out = 0;
    if (loop_seconds) *loop_seconds = 0.0;
    if (m <= 0 || nB != m || n < m || nN != n - m) {
        set_err(err, errlen, "bad dimensions");
        return EO_ERR_BAD_DIMS;
    }
    for (int64_t j = 0; j < nN && !g_continuation; ++j) { /* :139-151 */
        double di = d[N[j]];
        int infeasible;
        if (Nb[j] == EO_NB_LOWER) infeasible = di < -EPS;
        else if (Nb[j] == EO_NB_UPPER) infeasible = di > EPS;
        else infeasible = fabs(di) > EPS;
        if (infeasible) {
            set_err(err, errlen, "initial point of dual phase 2 is dual infeasible");
            return EO_ERR_PANIC;
        }
    }
    if (nN == 0) return EO_OPTIMAL;
    if (threads < 1) threads = 1;
    double *A_B = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *A_N = (double *)xmalloc(sizeof(double) * (size_t)(m * nN));
    for (int64_t i = 0; i < m; ++i) memcpy(A_B + i * m, A + B[i] * m, sizeof(double) * (size_t)m);
    for (int64_t j = 0; j < nN; ++j) memcpy(A_N + j * m, A + N[j] * m, sizeof(double) * (size_t)m);
    double *W = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *rho = (double *)xmalloc(sizeof(double) * (size_t)m);
    double *alpha = (double *)xmalloc(sizeof(double) * (size_t)nN);
    double *alpha_q = (double *)xmalloc(sizeof(double) * (size_t)m);
    int status = EO_ERR_PANIC;
    uint64_t iter = 0, entered = 0, since_refresh = 0;
    double t0 = 0.0;
    /* the dual loop has no singularity guard (dual…:241-253 unwrap): only an exactly singular basis fails */
    {
        lu_t f;
        lu_factor(&f, A_B, m, m);
        int bad = 0;
#pragma omp parallel num_threads(threads)
        {
            double *e = (double *)xmalloc(sizeof(double) * (size_t)m);
#pragma omp for schedule(static)
            for (int64_t k = 0; k < m; ++k) {
                memset(e, 0, sizeof(double) * (size_t)m);
                e[k] = 1.0;
                if (!lu_solve(&f, e)) {
#pragma omp atomic write
                    bad = 1;
                }
                for (int64_t i = 0; i < m; ++i) W[i * m + k] = e[i];
            }
            free(e);
        }
        lu_free(&f);
        if (bad) {
            set_err(err, errlen, "unwrap() on None: singular basis");
            goto done;
        }
    }
    t0 = now_seconds();
    for (;;) {
        if (iter >= max_iter) {
            status = EO_MAXITER;
            break;
        }
        iter += 1;
        entered += 1;
        if (refresh > 0 && since_refresh >= (uint64_t)refresh) {
            lu_t f;
            lu_factor(&f, A_B, m, m);
            double *e = (double *)xmalloc(sizeof(double) * (size_t)m);
            int bad = 0;
            for (int64_t k = 0; k < m && !bad; ++k) {
                memset(e, 0, sizeof(double) * (size_t)m);
                e[k] = 1.0;
                if (!lu_solve(&f, e)) bad = 1;
                for (int64_t i = 0; i < m; ++i) W[i * m + k] = e[i];
            }
            free(e);
            lu_free(&f);
            if (bad) {
                set_err(err, errlen, "unwrap() on None: singular basis");
                break;
            }
            since_refresh = 0;
        }
        int64_t r = -1;
        double delta = 0.0;
        int side = EO_NB_LOWER;
        for (int64_t i = 0; i < m && r < 0; ++i) { /* :200-236 */
            const int64_t bi = B[i];
            double xi = x[bi];
            switch (kind[bi]) {
            case EO_LOWER:
                if (xi < lb[bi] - EPS) { r = i; delta = xi - lb[bi]; side = EO_NB_LOWER; }
                break;
            case EO_UPPER:
                if (xi > ub[bi] + EPS) { r = i; delta = xi - ub[bi]; side = EO_NB_UPPER; }
                break;
            case EO_TWOSIDED:
                if (xi > ub[bi] + EPS) { r = i; delta = xi - ub[bi]; side = EO_NB_UPPER; }
                else if (xi < lb[bi] - EPS) { r = i; delta = xi - lb[bi]; side = EO_NB_LOWER; }
                break;
            default: break;
            }
        }
        if (r < 0) {
            status = EO_OPTIMAL;
            break;
        }
        memcpy(rho, W + r * m, sizeof(double) * (size_t)m);
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t j = 0; j < nN; ++j) {
            double dot = dot4(A_N + j * m, rho, m);
            if (fabs(dot) < g_zero_tol) dot = 0.0;
            alpha[j] = (delta < 0.0) ? -dot : dot;
        }
        int64_t q = -1;
        double theta_dual = 0.0;
        int nan_seen = 0;
        for (int64_t j = 0; j < nN; ++j) { /* :263-279 */
            int keep;
            if (Nb[j] == EO_NB_LOWER) keep = alpha[j] > EPS;
            else if (Nb[j] == EO_NB_UPPER) keep = alpha[j] < -EPS;
            else keep = 1;
            if (!keep) continue;
            double ratio = d[N[j]] / alpha[j];
            if (q < 0) {
                q = j;
                theta_dual = ratio;
            } else {
                if (isnan(ratio) || isnan(theta_dual)) {
                    nan_seen = 1;
                    break;
                }
                if (theta_dual > ratio) {
                    q = j;
                    theta_dual = ratio;
                }
            }
        }
        if (nan_seen) {
            set_err(err, errlen, "unwrap() on None: NaN in dual ratio test");
            status = EO_ERR_NAN;
            break;
        }
        if (q < 0) {
            status = EO_INFEASIBLE;
            break;
        }
        if (delta < 0.0) {
            for (int64_t j = 0; j < nN; ++j) alpha[j] = -alpha[j];
            theta_dual = -theta_dual;
        }
        const int64_t leaving_var = B[r];
        const int64_t entering_var = N[q];
        {
            const double *aq = A_N + q * m;
#pragma omp parallel for schedule(static) num_threads(threads)
            for (int64_t i = 0; i < m; ++i) {
                alpha_q[i] = dot4(W + i * m, aq, m);
                if (fabs(alpha_q[i]) < g_zero_tol) alpha_q[i] = 0.0;
            }
        }
        if (guard_trips(alpha_q[r], alpha_q, m)) { /* certified hybrid: nothing of this iteration is committed */
            status = EO_NEED_EXACT;
            entered -= 1;
            break;
        }
        if (alpha_q[r] == 0.0) {
            set_err(err, errlen, "unwrap() on None in dual FTRAN");
            break;
        }
        d[leaving_var] = -theta_dual;
        for (int64_t j = 0; j < nN; ++j) d[N[j]] -= theta_dual * alpha[j];
        d[entering_var] = 0.0;
        for (int64_t i = 0; i < m; ++i) y[i] += theta_dual * rho[i];
        double theta_primal = delta / alpha_q[r];
        for (int64_t i = 0; i < m; ++i) x[B[i]] -= theta_primal * alpha_q[i];
        x[entering_var] += theta_primal;
        if (g_trace) g_trace(g_trace_user, entered, q, r, entering_var, leaving_var);
        { /* B^-1 <- E B^-1 */
            const double ar = alpha_q[r];
#pragma omp parallel for schedule(static) num_threads(threads)
            for (int64_t i = 0; i < m; ++i) {
                double *wi = W + i * m;
                if (i == r) {
                    for (int64_t k = 0; k < m; ++k) wi[k] = rho[k] / ar;
                } else {
                    const double f = -(alpha_q[i] / ar);
                    if (f != 0.0)
                        for (int64_t k = 0; k < m; ++k) wi[k] = f * rho[k] + wi[k];
                }
            }
        }
        B[r] = entering_var;
        N[q] = leaving_var;
        Nb[q] = (uint8_t)side;
        double *cb = A_B + r * m, *cn = A_N + q * m;
        for (int64_t i = 0; i < m; ++i) {
            double t = cb[i];
            cb[i] = cn[i];
            cn[i] = t;
        }
        since_refresh += 1;
    }
    if (loop_seconds) *loop_seconds = now_seconds() - t0;
done:
    if (iters_out) *iters_out = entered;
    free(A_B); free(A_N); free(W); free(rho); free(alpha); free(alpha_q);
    return status;
}

/* ------------------------------------------------------------------ the certified hybrid (round 4)
 * NOT the reference's loop: the policy of the engine's default above 128 rows, restated here FIRST so that the device
 * implementation has a checker (as every extension: partial pricing, the dual rules, steepest edge).
 *
 *   fast:    the explicit-inverse loop (eo_*_binv_*; the reference's pivoting rules on a maintained B^-1) with the pivot
 *            guard on (eo_set_binv_guard: |pivot| < guard_abs stops it BEFORE the iteration is committed);
 *   exact:   the reference's own loop, a fresh LU every iteration (eo_primal/dual_solve_with_initial), continued from
 *            the arrays the fast loop left.
 *
 *   1. run fast until it stops;
 *   2. EO_MAXITER -> return it; an error of the fast loop (singular / NaN / panic) is treated like a terminal status;
 *   3. guard stop (EO_NEED_EXACT): run exact for up to K iterations (the suspicious iteration is its first one); a terminal
 *      status of the exact loop is the result; otherwise back to 1;
 *   4. terminal status S of fast (Optimal / Infeasible / Unbounded): run exact for up to K iterations.  If its FIRST
 *      iteration ends with S the status is certified and returned (the loop body that found it is counted once, as in the
 *      reference); if it ends otherwise, that is the result; if it pivots on, back to 1 with the counters advanced.
 *   Dual only: before every hand-over to the exact loop x_B is recomputed from a fresh LU of the basis,
 *   x_B = A_B^-1 (b - A_N x_N) — the leaving-row test (dual…:200-236) compares x_B with its bounds to within EPS = 1e-10,
 *   and what the inexact alpha_q of the fast loop has added to the carried x_B is far above that on ill-conditioned
 *   bases.  y and d stay the carried ones: recomputed from an LU of a basis of condition 1e7-1e9 they are themselves
 *   only good to 1e-9..1e-8 (two LU codes differ by that much), and the caller tests sums over d against EPS = 1e-10
 *   (dual…:45-50, dual_problem.rs:293-310) — measured both ways on ADLITTLE x 10 / x 18, the recomputed d fails those
 *   tests more often than the carried one (EO_HYBRID_RESYNC_YD=1 keeps the other variant for the record).  The primal keeps
 *   the carried x: its exact zeros at degenerate vertices are worth more than a recomputation
 *   that returns -3e-9 for them (measured: tests/campaign/hybrid_cpu.py, phase-1 objectives of -1e-8 and the reference's
 *   assert!(obj > -EPS) after a resync; none without).
 * Every decision that ends the solve is thus taken by the reference's arithmetic on a fresh factorisation
 * (primal…:173-189,289-292,404-406; dual…:241-246,281-284). */
static void hybrid_resync(int64_t m, int64_t n, const double *A, const double *c, const double *b, double *x, const int64_t *B,
                          const int64_t *N, int64_t nN, double *y, double *d) {
    double *A_B = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    double *t = (double *)xmalloc(sizeof(double) * (size_t)m);
    for (int64_t i = 0; i < m; ++i) memcpy(A_B + i * m, A + B[i] * m, sizeof(double) * (size_t)m);
    for (int64_t i = 0; i < m; ++i) t[i] = b[i];
    for (int64_t j = 0; j < nN; ++j) { /* t = b - A_N x_N, column by column in position order */
        const double xj = x[N[j]];
        if (xj == 0.0) continue;
        const double *cj = A + N[j] * m;
        for (int64_t i = 0; i < m; ++i) t[i] = t[i] - cj[i] * xj;
    }
    lu_t f;
    lu_factor(&f, A_B, m, m);
    if (lu_solve(&f, t)) { /* a singular basis keeps the carried vectors: the exact loop reports it */
        for (int64_t i = 0; i < m; ++i) x[B[i]] = t[i];
        for (int64_t i = 0; i < m; ++i) t[i] = c[B[i]];
        if (!getenv("EO_HYBRID_RESYNC_YD")) { /* default: x_B only — y and d are carried (see the header of this section) */
        } else if (lu_btran(&f, t)) { /* y = A_B^-T c_B, d = c - A^T y (dual_problem.rs:162-172 at this basis) */
            for (int64_t i = 0; i < m; ++i) y[i] = t[i];
            for (int64_t j = 0; j < nN; ++j) {
                const double *cj = A + N[j] * m;
                double dot = 0.0;
                for (int64_t i = 0; i < m; ++i) dot += cj[i] * y[i];
                d[N[j]] = c[N[j]] - dot;
            }
            for (int64_t i = 0; i < m; ++i) d[B[i]] = 0.0;
        }
    }
    (void)n;
    lu_free(&f);
    free(A_B);
    free(t);
}

/* counters[0] hand-overs after a guard stop, [1] terminal statuses examined, [2] of those: not confirmed,
 * [3] iterations done by the exact loop, [4] solves repeated from their start by the exact loop (certify or redo) */
static int hybrid_run(int dual, int64_t m, int64_t n, int64_t n_c, const double *A, const double *c, const double *b,
                      const uint8_t *kind, const double *lb, const double *ub, double *x, int64_t *B, int64_t nB,
                      int64_t *N, uint8_t *Nb, int64_t nN, double *y, double *d, uint64_t max_iter, uint64_t *iters_out,
                      int K, double guard_abs, int refresh, int threads, uint64_t *counters, char *err, size_t errlen) {
    uint64_t total = 0;
    int first = 1;
    int status = EO_MAXITER;
    const double save_rel = g_guard_rel, save_abs = g_guard_abs;
    const int save_cont = g_continuation;
    if (K < 1) K = 1;
    if (counters) counters[0] = counters[1] = counters[2] = counters[3] = counters[4] = 0;
    /* the start of the solve, for "certify or redo" (below) */
    double *x0 = NULL, *y0 = NULL, *d0 = NULL;
    int64_t *B0 = NULL, *N0 = NULL;
    uint8_t *Nb0 = NULL;
    if (m > 0 && nB == m && nN == n - m && nN > 0) {
        x0 = dcopy(x, n_c);
        B0 = icopy(B, nB);
        N0 = icopy(N, nN);
        Nb0 = bcopy8(Nb, nN);
        if (dual) {
            y0 = dcopy(y, m);
            d0 = dcopy(d, n_c);
        }
    }
    for (;;) {
        if (total >= max_iter) {
            status = EO_MAXITER;
            break;
        }
        uint64_t it = 0;
        int st;
        g_guard_rel = 0.0;
        g_guard_abs = guard_abs;
        g_continuation = first ? save_cont : 1;
        if (dual)
            st = eo_dual_binv_solve_with_initial(m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, y, d, max_iter - total,
                                                 &it, threads, refresh, NULL, err, errlen);
        else
            st = eo_primal_binv_solve_with_initial(m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, max_iter - total, &it,
                                                   threads, refresh, NULL, err, errlen);
        g_guard_rel = save_rel;
        g_guard_abs = save_abs;
        total += it;
        if (first && (st == EO_ERR_BAD_DIMS || (st == EO_ERR_PANIC && it == 0 && dual))) { /* entry checks of the seam */
            status = st;
            break;
        }
        first = 0;
        if (st == EO_MAXITER) {
            status = st;
            break;
        }
        const int guard_stop = st == EO_NEED_EXACT;
        if (dual) hybrid_resync(m, n, A, c, b, x, B, N, nN, y, d);
        uint64_t it2 = 0;
        int st2;
        uint64_t budget = max_iter - total;
        /* the loop body in which the fast loop found its terminal status is examined again, not counted twice */
        if (!guard_stop && total > 0) total -= 1, budget += 1;
        if (budget > (uint64_t)K) budget = (uint64_t)K;
        g_continuation = 1;
        if (dual)
            st2 = eo_dual_solve_with_initial(m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, y, d, budget, &it2, err, errlen);
        else
            st2 = eo_primal_solve_with_initial(m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, budget, &it2, err, errlen);
        g_continuation = save_cont;
        total += it2;
        if (counters) {
            counters[3] += it2;
            if (guard_stop) counters[0] += 1;
            else {
                counters[1] += 1;
                if (!(st2 == st && it2 <= 1)) counters[2] += 1;
            }
        }
        if (st2 != EO_MAXITER) {
            status = st2;
            break;
        }
    }
    g_continuation = save_cont;
    /* certify or redo: the invariants the reference's loop maintains, measured on the end point to within EPS — primal: x
     * within its bounds; dual: the loop's own entry assertion on d (dual…:139-151).  Violated: the explicit-inverse
     * stretch has let the point drift past EPS, and the solve is repeated from its start by the exact loop alone. */
    if (status == EO_OPTIMAL && x0) {
        double viol = 0.0; /* the SUM of the violations: the caller's phase-1 tests (primal…:42-50, dual…:45-50) are on sums */
        if (!dual) {
            for (int64_t i = 0; i < n_c; ++i) {
                double v = 0.0;
                switch (kind[i]) {
                case EO_LOWER: v = lb[i] - x[i]; break;
                case EO_UPPER: v = x[i] - ub[i]; break;
                case EO_TWOSIDED: v = fmax(lb[i] - x[i], x[i] - ub[i]); break;
                case EO_FIXED: v = fabs(x[i] - lb[i]); break;
                default: break;
                }
                if (v != v) v = INFINITY;
                if (v > 0.0) viol += v;
            }
        } else {
            for (int64_t j = 0; j < nN; ++j) {
                const double di = d[N[j]];
                double v = Nb[j] == EO_NB_LOWER ? -di : (Nb[j] == EO_NB_UPPER ? di : fabs(di));
                if (v != v) v = INFINITY;
                if (v > 0.0) viol += v;
            }
            /* a "box problem" (every bound TwoSided or Fixed, b = 0: DualPhase1's LP, dual_problem.rs:89-131): its dual
             * objective is minus the original problem's dual infeasibility, the number the caller tests against EPS
             * (dual…:45-50); below -EPS but within what the drift of the carried d can produce, the exact loop decides */
            int box = 1;
            for (int64_t i = 0; i < n_c && box; ++i) box = kind[i] == EO_TWOSIDED || kind[i] == EO_FIXED;
            for (int64_t i = 0; i < m && box; ++i) box = b[i] == 0.0;
            if (box) {
                const double o = dual_obj(m, n_c, b, kind, lb, ub, y, d);
                if (o <= -EPS && o > -1e-6) viol = INFINITY;
            }
        }
        if (!(viol <= EPS)) {
            memcpy(x, x0, sizeof(double) * (size_t)n_c);
            memcpy(B, B0, sizeof(int64_t) * (size_t)nB);
            memcpy(N, N0, sizeof(int64_t) * (size_t)nN);
            memcpy(Nb, Nb0, (size_t)nN);
            if (dual) {
                memcpy(y, y0, sizeof(double) * (size_t)m);
                memcpy(d, d0, sizeof(double) * (size_t)n_c);
            }
            uint64_t it3 = 0;
            if (dual)
                status = eo_dual_solve_with_initial(m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, y, d, max_iter, &it3, err, errlen);
            else
                status = eo_primal_solve_with_initial(m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, max_iter, &it3, err, errlen);
            total = it3;
            if (counters) counters[4] += 1;
        }
    }
    free(x0); free(B0); free(N0); free(Nb0); free(y0); free(d0);
    if (iters_out) *iters_out = total;
    return status;
}

int eo_primal_hybrid_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                                        const double *b, const uint8_t *kind, const double *lb, const double *ub,
                                        double *x, int64_t *B, int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN,
                                        uint64_t max_iter, uint64_t *iters, int K, double guard_abs, int refresh,
                                        int threads, uint64_t *counters5, char *err, size_t errlen) {
    return hybrid_run(0, m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, NULL, NULL, max_iter, iters, K, guard_abs,
                      refresh, threads, counters5, err, errlen);
}

int eo_dual_hybrid_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                                      const double *b, const uint8_t *kind, const double *lb, const double *ub, double *x,
                                      int64_t *B, int64_t nB, int64_t *N, uint8_t *Nb, int64_t nN, double *y, double *d,
                                      uint64_t max_iter, uint64_t *iters, int K, double guard_abs, int refresh,
                                      int threads, uint64_t *counters5, char *err, size_t errlen) {
    return hybrid_run(1, m, n, n_c, A, c, b, kind, lb, ub, x, B, nB, N, Nb, nN, y, d, max_iter, iters, K, guard_abs, refresh,
                      threads, counters5, err, errlen);
}

/* ------------------------------------------------------------------ DualPhase1 / DualPhase2 */

static eo_problem *problem_clone(const eo_problem *p) {
    eo_problem *q = eo_problem_new();
    for (int64_t i = 0; i < p->nvars; ++i)
        eo_add_var_with_id(q, p->vars[i].obj, p->vars[i].kind, p->vars[i].lb, p->vars[i].ub,
                           p->vars[i].id);
    for (int64_t i = 0; i < p->ncons; ++i)
        eo_add_constraint(q, p->cons[i].n, p->cons[i].ids, p->cons[i].coef, p->cons[i].op,
                          p->cons[i].rhs);
    return q;
}

/* y = A_B^{-T} c_B, d = c - A^T y  (dual_problem.rs:162-172, 276-283) */
static int dual_y_d(const eo_phase *sf, const int64_t *B, lu_t *f, double *y, double *d) {
    const int64_t m = sf->m, n = sf->n;
    double *A_B = (double *)xmalloc(sizeof(double) * (size_t)(m * m));
    for (int64_t i = 0; i < m; ++i) {
        memcpy(A_B + i * m, sf->A + B[i] * m, sizeof(double) * (size_t)m);
        y[i] = sf->c[B[i]];
    }
    lu_factor(f, A_B, m, m);
    free(A_B);
    if (!lu_btran(f, y)) return 0;
    for (int64_t j = 0; j < n; ++j) {
        const double *cj = sf->A + j * m;
        double dot = 0.0;
        for (int64_t i = 0; i < m; ++i) dot += cj[i] * y[i];
        d[j] = sf->c[j] - dot;
    }
    return 1;
}

/* dual_problem.rs:89-256 */
eo_phase *eo_dual_phase1(const eo_problem *prob, int *err) {
    if (err) *err = 0;
    eo_phase *orig = standard_form(prob);
    if (!orig) return NULL;
    const int64_t on = orig->n, om = orig->m;

    eo_problem *p1 = eo_problem_new();
    uint8_t *kept = (uint8_t *)xcalloc((size_t)(on + 1), 1);
    for (int64_t i = 0; i < on; ++i) { /* :99-112 */
        switch (orig->kind[i]) {
        case EO_FREE: kept[i] = 1; eo_add_var_with_id(p1, orig->c[i], EO_TWOSIDED, -1.0, 1.0, i); break;
        case EO_LOWER: kept[i] = 1; eo_add_var_with_id(p1, orig->c[i], EO_TWOSIDED, 0.0, 1.0, i); break;
        case EO_UPPER: kept[i] = 1; eo_add_var_with_id(p1, orig->c[i], EO_TWOSIDED, -1.0, 0.0, i); break;
        default: break;
        }
    }
    {
        int64_t *ids = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(on + 1));
        double *cf = (double *)xmalloc(sizeof(double) * (size_t)(on + 1));
        for (int64_t i = 0; i < om; ++i) { /* :114-134 */
            int64_t k = 0;
            for (int64_t j = 0; j < on; ++j) {
                if (kept[j]) {
                    ids[k] = j;
                    cf[k] = orig->A[i + j * om];
                    k++;
                }
            }
            if (k > 0) eo_add_constraint(p1, k, ids, cf, EO_EQ, 0.0);
        }
        free(ids);
        free(cf);
    }
    free(kept);

    eo_phase *sf = standard_form(p1); /* :136 */
    if (!sf) {
        eo_problem_free(p1);
        eo_phase_free(orig);
        return NULL;
    }
    const int64_t n = sf->n, m = sf->m;
    /* remember the id of each box-problem variable (phase_1_prob.variables[k].id) */
    sf->p1ids = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t k = 0; k < p1->nvars; ++k) sf->p1ids[k] = p1->vars[k].id;
    eo_problem_free(p1);

    /* :141-160 basis from LU(A^T) */
    double *At = (double *)xmalloc(sizeof(double) * (size_t)(n * m + 1));
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) At[j + i * n] = sf->A[i + j * m];
    lu_t ft;
    lu_factor(&ft, At, n, m);
    free(At);
    if (lu_small_diag(&ft) || n < m) {
        if (err) *err = EO_ERR_PANIC; /* "should always have a basis available" */
        lu_free(&ft);
        eo_phase_free(sf);
        eo_phase_free(orig);
        return NULL;
    }
    int64_t *perm_cols = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t j = 0; j < n; ++j) perm_cols[j] = j;
    perm_rows_i(&ft.p, perm_cols);
    lu_free(&ft);
    for (int64_t i = 0; i < m; ++i) sf->B[i] = perm_cols[i];
    sf->nB = m;
    for (int64_t k = m; k < n; ++k) {
        sf->N[k - m] = perm_cols[k];
        sf->Nb[k - m] = EO_NB_LOWER;
    }
    sf->nN = n - m;
    free(perm_cols);

    sf->y = (double *)xcalloc((size_t)(m + 1), sizeof(double));
    sf->d = (double *)xcalloc((size_t)(n + 1), sizeof(double));
    if (m > 0) { /* :166-226 */
        lu_t f;
        if (!dual_y_d(sf, sf->B, &f, sf->y, sf->d)) {
            if (err) *err = EO_ERR_PANIC;
            lu_free(&f);
            eo_phase_free(sf);
            eo_phase_free(orig);
            return NULL;
        }
        for (int64_t k = 0; k < sf->nN; ++k) {
            int64_t i = sf->N[k];
            if (sf->kind[i] == EO_TWOSIDED) {
                if (sf->d[i] >= 0.0) { sf->x[i] = sf->lb[i]; sf->Nb[k] = EO_NB_LOWER; }
                else { sf->x[i] = sf->ub[i]; sf->Nb[k] = EO_NB_UPPER; }
            } else if (sf->kind[i] == EO_FIXED) {
                sf->x[i] = sf->lb[i];
                sf->Nb[k] = (sf->d[i] >= 0.0) ? EO_NB_LOWER : EO_NB_UPPER;
            } else {
                if (err) *err = EO_ERR_PANIC;
                lu_free(&f);
                eo_phase_free(sf);
                eo_phase_free(orig);
                return NULL;
            }
        }
        double *bt = (double *)xmalloc(sizeof(double) * (size_t)(m + 1));
        matvec_sub(sf->A, m, n, sf->x, sf->b, bt);
        if (!lu_solve(&f, bt)) {
            if (err) *err = EO_ERR_PANIC;
            free(bt);
            lu_free(&f);
            eo_phase_free(sf);
            eo_phase_free(orig);
            return NULL;
        }
        for (int64_t i = 0; i < m; ++i) sf->x[sf->B[i]] = bt[i];
        free(bt);
        lu_free(&f);
    } else { /* :227-254 */
        for (int64_t k = 0; k < sf->nN; ++k) {
            int64_t i = sf->N[k];
            sf->Nb[k] = EO_NB_LOWER;
            if (sf->kind[i] == EO_TWOSIDED || sf->kind[i] == EO_FIXED) sf->x[i] = sf->lb[i];
            else {
                if (err) *err = EO_ERR_PANIC;
                eo_phase_free(sf);
                eo_phase_free(orig);
                return NULL;
            }
        }
        memcpy(sf->d, sf->c, sizeof(double) * (size_t)n);
    }
    sf->which = 3;
    sf->orig = orig;
    /* keep the user's problem data on the phase itself too (for phase 2 / fallback) */
    free(sf->orig_obj); free(sf->orig_kind); free(sf->orig_lb); free(sf->orig_ub);
    sf->orig_obj = NULL; sf->orig_kind = NULL; sf->orig_lb = sf->orig_ub = NULL;
    phase_copy_orig(sf, orig);
    return sf;
}

/* dual_problem.rs:258-404 */
eo_phase *eo_dual_phase2(const eo_phase *p1, int *err) {
    if (err) *err = 0;
    const eo_phase *o = p1->orig;
    const int64_t n = o->n, m = o->m;
    eo_phase *ph = phase_alloc(m, n, n);
    memcpy(ph->A, o->A, sizeof(double) * (size_t)(m * n));
    memcpy(ph->c, o->c, sizeof(double) * (size_t)n);
    memcpy(ph->b, o->b, sizeof(double) * (size_t)m);
    memcpy(ph->kind, o->kind, (size_t)n);
    memcpy(ph->lb, o->lb, sizeof(double) * (size_t)n);
    memcpy(ph->ub, o->ub, sizeof(double) * (size_t)n);
    phase_copy_orig(ph, o);
    ph->which = 4;

    uint8_t *is_basic = (uint8_t *)xcalloc((size_t)(n + 1), 1);
    for (int64_t i = 0; i < p1->nB; ++i) {
        int64_t index = p1->p1ids[p1->B[i]];
        is_basic[index] = 1;
        ph->B[i] = index;
    }
    ph->nB = p1->nB;
    ph->y = (double *)xcalloc((size_t)(m + 1), sizeof(double));
    ph->d = (double *)xcalloc((size_t)(n + 1), sizeof(double));

    if (ph->nB > 0) {
        if (ph->nB != m) { /* select_columns + lu().solve would panic on shape */
            if (err) *err = EO_ERR_PANIC;
            free(is_basic);
            eo_phase_free(ph);
            return NULL;
        }
        lu_t f;
        if (!dual_y_d(ph, ph->B, &f, ph->y, ph->d)) {
            if (err) *err = EO_ERR_PANIC;
            lu_free(&f);
            free(is_basic);
            eo_phase_free(ph);
            return NULL;
        }
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (is_basic[i]) continue;
            double di = ph->d[i];
            double xi = 0.0;
            int bound = EO_NB_LOWER;
            switch (ph->kind[i]) {
            case EO_FREE:
                if (!(fabs(di) < EPS)) { if (err) *err = EO_ERR_PANIC; }
                xi = 0.0; bound = EO_NB_FREE; break;
            case EO_LOWER:
                if (!(di > -EPS)) { if (err) *err = EO_ERR_PANIC; }
                xi = ph->lb[i]; bound = EO_NB_LOWER; break;
            case EO_UPPER:
                if (!(di < EPS)) { if (err) *err = EO_ERR_PANIC; }
                xi = ph->ub[i]; bound = EO_NB_UPPER; break;
            case EO_TWOSIDED:
                if (di >= 0.0) { xi = ph->lb[i]; bound = EO_NB_LOWER; }
                else { xi = ph->ub[i]; bound = EO_NB_UPPER; }
                break;
            default: xi = ph->lb[i]; bound = EO_NB_LOWER; break;
            }
            ph->N[k] = i;
            ph->Nb[k] = (uint8_t)bound;
            ph->x[i] = xi;
            k++;
        }
        ph->nN = k;
        if (err && *err) {
            lu_free(&f);
            free(is_basic);
            eo_phase_free(ph);
            return NULL;
        }
        /* x_B = A_B^{-1} (b - A_N x_N) :326-336 */
        double *xn = (double *)xcalloc((size_t)(n + 1), sizeof(double));
        for (int64_t kk = 0; kk < ph->nN; ++kk) xn[ph->N[kk]] = ph->x[ph->N[kk]];
        double *bt = (double *)xmalloc(sizeof(double) * (size_t)(m + 1));
        matvec_sub(ph->A, m, n, xn, ph->b, bt);
        free(xn);
        if (!lu_solve(&f, bt)) {
            if (err) *err = EO_ERR_PANIC;
            free(bt);
            lu_free(&f);
            free(is_basic);
            eo_phase_free(ph);
            return NULL;
        }
        for (int64_t i = 0; i < m; ++i) ph->x[ph->B[i]] = bt[i];
        free(bt);
        lu_free(&f);
    } else { /* :351-402 */
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (is_basic[i]) continue;
            int bound;
            switch (ph->kind[i]) {
            case EO_FREE: ph->x[i] = 0.0; bound = EO_NB_FREE; break;
            case EO_LOWER: ph->x[i] = ph->lb[i]; bound = EO_NB_LOWER; break;
            case EO_UPPER: ph->x[i] = ph->ub[i]; bound = EO_NB_UPPER; break;
            case EO_TWOSIDED: ph->x[i] = ph->lb[i]; bound = EO_NB_LOWER; break;
            default: ph->x[i] = ph->lb[i]; bound = EO_NB_LOWER; break;
            }
            ph->N[k] = i;
            ph->Nb[k] = (uint8_t)bound;
            k++;
        }
        ph->nN = k;
        memcpy(ph->d, ph->c, sizeof(double) * (size_t)n);
    }
    free(is_basic);
    return ph;
}

/* ------------------------------------------------------------------ solve() drivers */

void eo_result_free(eo_result *r) {
    if (r && r->x) {
        free(r->x);
        r->x = NULL;
    }
}

static int run_primal(eo_phase *ph, uint64_t max_iter, uint64_t *iters, char *err, size_t errlen) {
    int64_t nN = ph->nN;
    int st = eo_primal_solve_with_initial(ph->m, ph->n, ph->n_c, ph->A, ph->c, ph->b, ph->kind,
                                          ph->lb, ph->ub, ph->x, ph->B, ph->nB, ph->N, ph->Nb, nN,
                                          max_iter, iters, err, errlen);
    if (ph->m == 0) {
        /* the trivial solver rebuilt N: count what it wrote (every visited variable) */
        int64_t k = 0;
        solve_trivial(ph->n_c, ph->c, ph->kind, ph->lb, ph->ub, ph->x, ph->N, ph->Nb, &k, 1);
        ph->nN = k;
    }
    return st;
}
static int run_dual(eo_phase *ph, uint64_t max_iter, uint64_t *iters, char *err, size_t errlen) {
    int st = eo_dual_solve_with_initial(ph->m, ph->n, ph->n_c, ph->A, ph->c, ph->b, ph->kind,
                                        ph->lb, ph->ub, ph->x, ph->B, ph->nB, ph->N, ph->Nb,
                                        ph->nN, ph->y, ph->d, max_iter, iters, err, errlen);
    if (ph->m == 0) {
        int64_t k = 0;
        solve_trivial(ph->n_c, ph->c, ph->kind, ph->lb, ph->ub, ph->x, ph->N, ph->Nb, &k, 1);
        ph->nN = k;
    }
    return st;
}

static void fill_optimal(eo_result *out, const eo_phase *ph) {
    out->status = EO_OPTIMAL;
    out->obj = eo_phase_obj(ph); /* solver.rs:46-48 */
    out->nx = ph->n_orig_vars;   /* standard_form.rs:71-74 */
    out->x = dcopy(ph->x, ph->n_orig_vars);
}

/* primal_simplex_solver.rs:32-93 */
static int solve_primal(const eo_problem *p, uint64_t max_iter, eo_result *out) {
    int perr = 0;
    eo_phase *p1 = eo_primal_phase1(p, &perr);
    if (!p1) {
        out->status = perr ? perr : EO_INFEASIBLE;
        return out->status;
    }
    int st = run_primal(p1, max_iter, &out->iters1, out->err, sizeof(out->err));
    if (st < 0) { out->status = st; eo_phase_free(p1); return st; }
    if (st == EO_OPTIMAL) {
        double obj = eo_phase_obj(p1);
        if (!(obj > -EPS)) { /* :43 */
            out->status = EO_ERR_PANIC;
            snprintf(out->err, sizeof(out->err), "assertion failed: obj > -EPS");
            eo_phase_free(p1);
            return out->status;
        }
        if (!(obj < EPS)) { out->status = EO_INFEASIBLE; eo_phase_free(p1); return out->status; }
    } else if (st == EO_INFEASIBLE) {
        out->status = EO_INFEASIBLE; eo_phase_free(p1); return out->status;
    } else if (st == EO_UNBOUNDED) {
        out->status = EO_ERR_PANIC;
        snprintf(out->err, sizeof(out->err), "primal phase 1 should never be unbounded");
        eo_phase_free(p1);
        return out->status;
    } else { /* MaxIter :61-64 */
        out->status = EO_MAXITER; out->obj = INFINITY; eo_phase_free(p1); return out->status;
    }
    eo_phase *p2 = eo_primal_phase2(p1);
    eo_phase_free(p1);
    st = run_primal(p2, max_iter, &out->iters2, out->err, sizeof(out->err));
    if (st == EO_OPTIMAL) fill_optimal(out, p2);
    else if (st == EO_INFEASIBLE) {
        out->status = EO_ERR_PANIC;
        snprintf(out->err, sizeof(out->err), "primal phase 2 should never be infeasible");
    } else if (st == EO_MAXITER) { out->status = EO_MAXITER; out->obj = eo_phase_obj(p2); }
    else out->status = st;
    eo_phase_free(p2);
    return out->status;
}

/* rebuild the user's Problem from what a phase kept (only for the dual->primal fallback) */
/* dual_simplex_solver.rs:33-108 */
static int solve_dual(const eo_problem *p, uint64_t max_iter, eo_result *out) {
    int perr = 0;
    eo_phase *p1 = eo_dual_phase1(p, &perr);
    if (!p1) {
        out->status = perr ? perr : EO_INFEASIBLE;
        return out->status;
    }
    int st = run_dual(p1, max_iter, &out->iters1, out->err, sizeof(out->err));
    if (st < 0) { out->status = st; eo_phase_free(p1); return st; }
    if (st == EO_OPTIMAL) {
        double obj = eo_phase_dual_obj(p1);
        if (!(obj < EPS)) { /* :45 */
            out->status = EO_ERR_PANIC;
            snprintf(out->err, sizeof(out->err), "assertion failed: obj < EPS");
            eo_phase_free(p1);
            return out->status;
        }
        if (!(obj > -EPS)) {
            /* :51-66 classify with PrimalSimplexSolver::default() (max_iter 1000) */
            eo_phase_free(p1);
            eo_result r2;
            memset(&r2, 0, sizeof(r2));
            eo_problem *pc = problem_clone(p);
            solve_primal(pc, 1000, &r2);
            eo_problem_free(pc);
            if (r2.status == EO_OPTIMAL) {
                eo_result_free(&r2);
                out->status = EO_ERR_PANIC;
                snprintf(out->err, sizeof(out->err),
                         "dual infeasible but primal fallback found an optimum");
                return out->status;
            }
            out->status = r2.status;
            out->obj = r2.obj;
            memcpy(out->err, r2.err, sizeof(out->err));
            return out->status;
        }
    } else if (st == EO_INFEASIBLE || st == EO_UNBOUNDED) {
        out->status = EO_ERR_PANIC;
        snprintf(out->err, sizeof(out->err), "dual phase 1 should never be infeasible/unbounded");
        eo_phase_free(p1);
        return out->status;
    } else {
        out->status = EO_MAXITER; out->obj = INFINITY; eo_phase_free(p1); return out->status;
    }
    eo_phase *p2 = eo_dual_phase2(p1, &perr);
    eo_phase_free(p1);
    if (!p2) { out->status = perr ? perr : EO_ERR_PANIC; return out->status; }
    st = run_dual(p2, max_iter, &out->iters2, out->err, sizeof(out->err));
    if (st == EO_OPTIMAL) fill_optimal(out, p2);
    else if (st == EO_UNBOUNDED) {
        out->status = EO_ERR_PANIC;
        snprintf(out->err, sizeof(out->err), "dual phase 2 should never return unbounded");
    } else if (st == EO_MAXITER) { out->status = EO_MAXITER; out->obj = eo_phase_dual_obj(p2); }
    else out->status = st;
    eo_phase_free(p2);
    return out->status;
}

int eo_solve(const eo_problem *p, int solver, uint64_t max_iter, eo_result *out) {
    memset(out, 0, sizeof(*out));
    return solver == 0 ? solve_primal(p, max_iter, out) : solve_dual(p, max_iter, out);
}

/* ------------------------------------------------------------------ synthetic family */
static inline double sm64_next(uint64_t *s) {
    *s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
void eo_synth_dense_lp(uint64_t seed, int64_t m, int64_t n, double *A, double *b, double *c) {
    uint64_t s = seed;
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) A[i + j * m] = 0.1 + sm64_next(&s);
    double *x0 = (double *)xmalloc(sizeof(double) * (size_t)n);
    for (int64_t j = 0; j < n; ++j) x0[j] = sm64_next(&s);
    for (int64_t i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int64_t j = 0; j < n; ++j) acc += A[i + j * m] * x0[j];
        b[i] = acc;
    }
    for (int64_t j = 0; j < n; ++j) c[j] = -(0.1 + sm64_next(&s));
    free(x0);
}
