/*
 * ellp_hip.h — C ABI of the MI355X-native revised-simplex pivot engine.
 *
 * Drop-in boundary for kehlert/ellp 0.2.0's hot path.  The reference has no FFI; the seam
 * this ABI replaces is the generic method
 *
 *   PrimalSimplexSolver::solve_with_initial   src/solvers/primal/primal_simplex_solver.rs:95
 *   DualSimplexSolver::solve_with_initial     src/solvers/dual/dual_simplex_solver.rs:110
 *
 * i.e. everything between `prob.unpack()` and the returned SolutionStatus: the per-iteration
 * BTRAN, pricing, entering selection, FTRAN, ratio test, point update and basis swap
 * (primal…:160-235 + pivot() :238-435; dual…:188-334).  Problem / StandardForm / phase
 * construction / the m == 0 trivial solver stay on the host (INTEGRATION.md shows the Rust
 * binding a maintainer would add).
 *
 * Conventions (mirroring what solve_with_initial has in hand, standard_form.rs:21-34):
 *   - A is m x n, column-major, leading dimension m (nalgebra DMatrix storage), read-only.
 *   - c, x, bound_kind/lb/ub have length n_c >= n (n_c > n only in primal phase 1 with free
 *     variables, primal_problem.rs:141,223,250-253).  b has length m.
 *   - Bound (problem.rs:191-197) is flattened to kind + lb + ub:
 *       0 Free, 1 Lower(lb), 2 Upper(ub), 3 TwoSided(lb, ub), 4 Fixed(lb)
 *   - Point{x, N, B} (standard_form.rs:21-25) is flattened to x, B_index[n_B],
 *     N_index[n_N], N_bound[n_N] (0 Lower, 1 Upper, 2 Free; standard_form.rs:206-210) and is
 *     updated IN PLACE, so the warm-start contract (phase 1 -> phase 2) survives.
 *   - No unwinding across the boundary: Err(EllPError) and panics of the reference come back
 *     as negative status codes with a message in errbuf.
 *   - Re-entrant: no global mutable state; every call/engine owns its HIP stream.
 *   - There is NO CPU fallback: if no HIP device is usable the call returns ELLP_ERR_DEVICE.
 */
#ifndef ELLP_HIP_H
#define ELLP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELLP_HIP_ABI_VERSION 1

/* SolutionStatus (src/solver.rs:28-33) + error codes */
typedef enum ellp_status {
    ELLP_OPTIMAL = 0,
    ELLP_INFEASIBLE = 1,
    ELLP_UNBOUNDED = 2,
    ELLP_MAXITER = 3,
    ELLP_ERR_BAD_DIMS = -1, /* Err("invalid B/N, has .. elements but .. expected") primal…:124-140 */
    ELLP_ERR_SINGULAR = -2, /* Err("invalid B, A_B is not invertible")              primal…:175-179 */
    ELLP_ERR_NAN = -3,      /* NaN met in pricing / ratio keys                      primal…:282    */
    ELLP_ERR_DEVICE = -4,   /* HIP runtime error / no device / extension missing                   */
    ELLP_ERR_ARG = -5,      /* NULL pointer, index out of range, m == 0 (host handles that case)   */
    ELLP_ERR_PANIC = -6     /* an assert!/panic!/unwrap() of the reference would have fired        */
} ellp_status;

enum { ELLP_BOUND_FREE = 0, ELLP_BOUND_LOWER = 1, ELLP_BOUND_UPPER = 2, ELLP_BOUND_TWOSIDED = 3,
       ELLP_BOUND_FIXED = 4 };
enum { ELLP_NB_LOWER = 0, ELLP_NB_UPPER = 1, ELLP_NB_FREE = 2 };

#define ELLP_MAX_ITER_NONE UINT64_MAX /* PrimalSimplexSolver::new(None), primal…:26-30 */
#define ELLP_FLAG_DENSE_PRICING 1      /* ellp_opts.flags */
#define ELLP_FLAG_DUAL_MAX_VIOLATION 2 /* ellp_opts.flags */
#define ELLP_FLAG_PRIMAL_STEEPEST_EDGE 4 /* ellp_opts.flags */
#define ELLP_FLAG_NO_CERTIFY 8 /* ellp_opts.flags: the plain explicit-inverse engine where the default is the certified hybrid */
#define ELLP_FLAG_DUAL_BOUND_FLIPPING 16 /* ellp_opts.flags */

typedef struct ellp_opts {
    uint64_t max_iter;       /* self.max_iter (primal…:16, dual…:17); default 1000 (:21) */
    double eps;              /* EPS, src/util.rs:1; <= 0 selects 1e-10 */
    int32_t device;          /* HIP device ordinal; < 0 = current device */
    int32_t refactor_period; /* iterations between refactorisations of B^-1; <= 0 = default */
    int32_t btran_mode;      /* 0 default (incremental u, periodic refresh); 1 = u = B^-T c_B every iteration */
    int32_t poll_interval;   /* iterations enqueued between host polls of the status word; <= 0 = default */
    int32_t profile;         /* != 0: bracket every launch with HIP events (ellp_stats.kernel_ms) */
    int32_t use_graph;       /* reserved, must be 0 (hipGraph replay of the launch sequence is not implemented:
                                on gfx950 the per-iteration cost is GPU-side dispatch, not host launches) */
    int32_t pipeline;        /* launch structure of an iteration: 0 = engine default by size — m <= 128: 3; 128 < m <= 1024: the
                                CERTIFIED HYBRID: 1 with a pivot guard, every terminal status and every guarded iteration
                                re-examined by the LU-per-iteration kernel of 3 from the same arrays (DESIGN.md §3.1c;
                                ELLP_FLAG_NO_CERTIFY: plain 1 / 2 by size); m > 1024: 2 with the same guard in its kernels'
                                prologues, refused iterations and terminal statuses run on a fresh LU of the basis (§3.1d) —,
                                1 = three launches (pricing | FTRAN | eta update), 2 = two bandwidth
                                passes (primal: pricing | eta update of the previous pivot fused with this iteration's
                                FTRAN; dual: pricing | FTRAN fused with this iteration's eta update, + a closing block),
                                3 = the whole loop in one persistent workgroup with an LU per iteration (m <= 1024: factors in
                                LDS up to 128 rows, in global memory above) */
    int32_t trace_len;       /* > 0: keep the objective after each of the last `trace_len` iterations in a ring buffer on
                                the device (ellp_engine_read_trace) — what the reference's `debug!("{iter} | {obj}")` line
                                (primal…:161, dual…:189) prints; off by default, as the reference's logging is */
    int32_t partial_segments; /* > 1: partial pricing (an extension the reference's README lists as future work, not its
                                 behaviour): the nonbasic positions are cut into this many segments of
                                 ceil(|N| / segments) positions; an iteration prices one segment with the reference's
                                 entering rule; a pass that finds no candidate moves to the next segment (and counts
                                 as an iteration), `segments` such passes in a row are the optimality test.  Primal
                                 engines on one GPU, three-launch pipeline.  0 or 1: every column every iteration */
    int32_t flags;           /* bit 0 (ELLP_FLAG_DENSE_PRICING): stream every nonbasic column in the primal pricing pass,
                                also the unit columns (slacks, artificials), whose dot product the kernels otherwise
                                form from their single entry — same numbers, measured both ways by bench.py
                                bit 1 (ELLP_FLAG_DUAL_MAX_VIOLATION), an EXTENSION, dual engines: the leaving
                                row is the basic position with the LARGEST bound violation (first of equals) instead of the
                                first violated one (dual_simplex_solver.rs:200-236).  Not the reference's rule — restated in
                                the oracle (eo_set_dual_rule(2)) and checked against it; 24 x fewer iterations on the 200 x 500
                                LP of SURVEY.md §8d, and what makes a dual solve at config 4's size finish at all
                                bit 2 (ELLP_FLAG_PRIMAL_STEEPEST_EDGE), an EXTENSION, primal engines: steepest-edge pricing
                                (exact Goldfarb-Reid weights, ellp_se.inc) instead of the reference's Dantzig rule
                                (primal_simplex_solver.rs:253-287); restated in the oracle (eo_set_primal_rule(1)); runs on the
                                three-launch explicit-inverse engine at every size, with the reactive tiny-pivot maintenance
                                on.  Meant for RESIDENT solves (both phases on one engine, ellp_engine_rephase): a second
                                engine created with this flag at phase 1's end basis broke down at config 5's size (4000 x
                                40000) — the ratio test has no pivot-size safeguard, as the reference's has none
                                bit 3 (ELLP_FLAG_NO_CERTIFY): with pipeline 0, the plain explicit-inverse engine above 128 rows — no
                                pivot guard, no certificate behind a terminal status, no repeated solve (measurements, and the
                                replay side of a sharded run's self-check: sharded engines run without them)
                                bit 4 (ELLP_FLAG_DUAL_BOUND_FLIPPING), an EXTENSION (ellp's README.md:114-116),
                                dual engines: the long-step ("bound flipping") ratio test — the dual step goes past the
                                breakpoints of BOXED nonbasic variables, each moved to its other bound instead of entering, for as
                                long as the leaving row's infeasibility |delta| minus the sum of |alpha_j| (ub_j - lb_j) over the
                                passed breakpoints stays positive; breakpoints in (ratio, position) order; x_B follows the flips
                                by one more solve with the iteration's LU.  Restated in the oracle first (eo_set_dual_rule bit
                                0) and reproduced bit for bit.  Runs on the LU-per-iteration kernels only (pipeline 0 or 3, up
                                to 1,024 rows; with this flag pipeline 0 selects them at every such size): more rows, pipeline
                                1 / 2, partial pricing, or a call that needs the explicit-inverse engine (ellp_engine_step,
                                ellp_engine_refactor, sharding) return ELLP_ERR_ARG.  Combines with bit 1.  Ignored by primal
                                engines. */
} ellp_opts;

/* kernel ids for ellp_stats.kernel_ms / kernel_calls */
enum {
    ELLP_K_PRICE = 0,   /* r = c_N - A_N^T u (+ keys)          primal…:189, :253-270 */
    ELLP_K_SELECT = 1,  /* entering fold                        primal…:271-287       */
    ELLP_K_FTRAN = 2,   /* d = +-B^-1 a_q                       primal…:295-300, dual…:294 */
    ELLP_K_RATIO = 3,   /* ratio test + x update + swap         primal…:305-417, :205-232 */
    ELLP_K_UPDATE = 4,  /* rank-1 (eta) update of B^-1          (replaces primal…:173 / dual…:241) */
    ELLP_K_BTRAN = 5,   /* u = B^-T c_B                         primal…:184-187 */
    ELLP_K_REFACTOR = 6,/* B^-1 from A_B (all launches of one refactorisation) */
    ELLP_K_DLEAVE = 7,  /* dual leaving scan + rho               dual…:200-253 */
    ELLP_K_DPRICE = 8,  /* alpha = A_N^T rho (+ ratios)          dual…:255-278 */
    ELLP_K_DSELECT = 9, /* dual ratio argmin                     dual…:279-289 */
    ELLP_K_DUPDATE = 10,/* d, y, x updates + swap                dual…:296-333 */
    ELLP_K_EVENT_COST = 11, /* not a kernel: kernel_ms[11] = the event-bracket cost subtracted per launch */
    ELLP_K_COUNT = 12
};

typedef struct ellp_stats {
    uint64_t iters;       /* loop bodies entered (the reference's `iter` bookkeeping) */
    uint64_t pivots;      /* basis changes */
    uint64_t bound_flips; /* entering variable moved bound-to-bound (primal…:223-231) */
    uint64_t refactors;
    double obj;           /* c.x (primal) / dual objective (dual) at return */
    double t_loop_s;      /* wall time of the device loop */
    double t_setup_s;     /* upload + initial factorisation */
    double kernel_ms[ELLP_K_COUNT];      /* profile != 0 only: summed HIP-event time */
    uint64_t kernel_calls[ELLP_K_COUNT]; /* profile != 0 only */
} ellp_stats;

/* Fills *o with the defaults that reproduce reference behaviour (max_iter 1000, eps 1e-10). */
void ellp_default_opts(ellp_opts *o);

/* Number of usable HIP devices (0 if none / runtime unavailable). Never throws. */
int ellp_hip_device_count(void);
int ellp_hip_abi_version(void);

/*
 * PrimalSimplexSolver::solve_with_initial (primal_simplex_solver.rs:95-236).
 * Synchronous: uploads, runs the device loop, writes x / B_index / N_index / N_bound back.
 */
ellp_status ellp_primal_solve_with_initial(
    int64_t m, int64_t n, int64_t n_c,
    const double *A, const double *c, const double *b,
    const uint8_t *bound_kind, const double *lb, const double *ub,
    double *x,
    int64_t *B_index, int64_t n_B,
    int64_t *N_index, uint8_t *N_bound, int64_t n_N,
    const ellp_opts *opts, ellp_stats *stats, char *errbuf, size_t errbuf_len);

/*
 * DualSimplexSolver::solve_with_initial (dual_simplex_solver.rs:110-335).
 * y (m) and d (n_c) are the DualFeasiblePoint's vectors (dual_problem.rs:12-16), in/out.
 */
ellp_status ellp_dual_solve_with_initial(
    int64_t m, int64_t n, int64_t n_c,
    const double *A, const double *c, const double *b,
    const uint8_t *bound_kind, const double *lb, const double *ub,
    double *x,
    int64_t *B_index, int64_t n_B,
    int64_t *N_index, uint8_t *N_bound, int64_t n_N,
    double *y, double *d,
    const ellp_opts *opts, ellp_stats *stats, char *errbuf, size_t errbuf_len);

/*
 * Resident form of the same path: the tableau stays in HBM between calls, so a caller
 * (bench.py, a phase-1 -> phase-2 hand-off, a windowed parity test) can run the loop in
 * slices without re-uploading.  create = unpack + gather (primal…:99-155); run = the loop for
 * at most `max_iters` further iterations (returns ELLP_MAXITER when the slice is used up,
 * exactly as the loop does when iter > max_iter); read = copy the point back.
 */
typedef struct ellp_engine ellp_engine;
enum { ELLP_ENGINE_PRIMAL = 0, ELLP_ENGINE_DUAL = 1 };

ellp_status ellp_engine_create(
    int kind,
    int64_t m, int64_t n, int64_t n_c,
    const double *A, const double *c, const double *b,
    const uint8_t *bound_kind, const double *lb, const double *ub,
    const double *x,
    const int64_t *B_index, int64_t n_B,
    const int64_t *N_index, const uint8_t *N_bound, int64_t n_N,
    const double *y, const double *d, /* dual only, else NULL */
    const ellp_opts *opts, ellp_engine **out, char *errbuf, size_t errbuf_len);

/*
 * Primal phase 1 built ON THE DEVICE (primal_problem.rs:236-246, the branch without free
 * variables): the caller passes the standard form (A m x n, b, bounds) and the nonbasic start it has chosen
 * for the n original variables (x[j] = the bound value, N_bound[j] = its label, primal_problem.rs:95-135);
 * the engine appends the m artificial columns itself — b~ = b - A x (one pass over A in HBM), column n+i =
 * signum(b~_i) e_i, value |b~_i|, cost 1, Lower(0), basic — so neither the m x m block nor b~ is computed or
 * uploaded by the host.  The resulting engine has n + m columns (read_point sizes) and is what
 * ellp_engine_create would have built from the phase-1 arrays of the reference.
 */
ellp_status ellp_engine_create_primal_phase1(
    int64_t m, int64_t n,
    const double *A, const double *b,
    const uint8_t *bound_kind, const double *lb, const double *ub,
    const double *x, const uint8_t *N_bound,
    const ellp_opts *opts, ellp_engine **out, char *errbuf, size_t errbuf_len);

ellp_status ellp_engine_run(ellp_engine *e, uint64_t max_iters, ellp_stats *stats,
                            char *errbuf, size_t errbuf_len);

/* Certified endings (pipeline 0, m > 128): a status Optimal / Infeasible / Unbounded that ellp_engine_run
 * (and the one-shot solve_with_initial entry points) return has been decided by the reference's arithmetic on a fresh LU of the
 * final basis.  Between 129 and 1,024 rows a solve that ends Optimal on a point violating an invariant of the reference's loop by
 * more than EPS is repeated from the arrays the phase started with by the LU-per-iteration kernel alone; ellp_stats.iters is then
 * the repeated solve's count, and the engine keeps running on that kernel (no resident inverse: ellp_engine_dual_rephase returns
 * ELLP_ERR_ARG, as on any engine of that kind).  Above 1,024 rows (up to 8,192) the same repetition runs every loop body on a
 * fresh LU of the basis over all CUs (about 18 us x m per iteration), and is taken only when the iterations the phase needed,
 * at that price, stay within ELLP_REDO_MAX_SECONDS (environment, default 900); otherwise the point is returned as it stands
 * and counted (ELLP_TAP_STATE, "not certified").  The resident inverse follows that loop: the next phase runs fast again.
 *
 * ellp_engine_run(e, K) runs up to K loop bodies.  On the two-launch primal pipeline (m >= 384) a slice leaves the
 * ratio test of its last iteration to the next slice's first kernel; when the slice spends the caller's whole budget
 * (ellp_opts.max_iter loop bodies since the engine was made or re-phased) that iteration is completed before the
 * status is taken, so that — as in the reference, which runs max_iter FULL loop bodies (primal…:162-202) — an
 * unbounded ray found in the last permitted iteration is reported as ELLP_UNBOUNDED, not ELLP_MAXITER.
 * ellp_engine_read_point completes an open iteration too; it returns ELLP_OPTIMAL when the point was delivered and the
 * loop can go on (or had ended), otherwise the status that completing the open iteration produced (ELLP_UNBOUNDED,
 * ELLP_ERR_PANIC, ELLP_ERR_NAN): the arrays are filled in either case. */
ellp_status ellp_engine_read_point(ellp_engine *e, double *x, int64_t *B_index,
                                   int64_t *N_index, uint8_t *N_bound, double *y, double *d,
                                   char *errbuf, size_t errbuf_len);

/*
 * Column-block sharding of the pricing pass over several GPUs (one process per GPU).
 * The nonbasic positions are split into `world` contiguous blocks; rank k prices block k and
 * writes (per-block maxima / argmins, keys, reduced costs) into segment k of one exchange
 * buffer of world*seg doubles.  Between ellp_engine_step(e, 0) [pricing] and
 * ellp_engine_step(e, 1) [FTRAN + ratio test + eta update] the caller all-gathers that buffer
 * (RCCL over xGMI via torch.distributed, see ellp_amd/dist.py); everything else is replicated on
 * every rank and deterministic, so all ranks take the same pivots.
 *   segment_doubles: doubles per rank segment for a given world size
 *   set_shard   : choose (rank, world) before the first step; exchange_buffer = caller-owned
 *                 device memory of world*segment_doubles doubles (e.g. a torch tensor's data_ptr,
 *                 so the collective library sees its own allocation), or NULL to let the engine
 *                 allocate it
 *   exchange_info: device pointer of the buffer, doubles per segment, rank, world
 *   set_stream  : run the engine's launches on a caller-owned HIP stream (NULL = its own)
 *   step        : enqueue one half-iteration (never blocks): 0 = pricing, 1 = the rest,
 *                 2 = the rest of this iteration followed by the next iteration's pricing
 *   poll        : copy the status word back; ELLP_MAXITER = still running
 */
int64_t ellp_engine_segment_doubles(ellp_engine *e, int world);
ellp_status ellp_engine_set_shard(ellp_engine *e, int rank, int world, void *exchange_buffer, char *errbuf,
                                  size_t errbuf_len);
ellp_status ellp_engine_exchange_info(ellp_engine *e, void **base, int64_t *seg_doubles, int *rank, int *world);
ellp_status ellp_engine_set_stream(ellp_engine *e, void *hip_stream);
ellp_status ellp_engine_step(ellp_engine *e, int phase, char *errbuf, size_t errbuf_len);
ellp_status ellp_engine_poll(ellp_engine *e, ellp_stats *stats, char *errbuf, size_t errbuf_len);

/*
 * Phase-1 -> phase-2 hand-off of the primal method without leaving HBM (SURVEY.md §8 f2;
 * primal_problem.rs:263-291): the matrix, the basis, the point and B^-1 stay where they are; the
 * costs and bounds (n_c entries each) are replaced, and a nonbasic variable whose new bound is Free
 * gets the label Free (:285-289).  Status and counters start afresh, as for a new
 * solve_with_initial call.  Primal engines only.
 */
ellp_status ellp_engine_rephase(ellp_engine *e, const double *c, const uint8_t *bound_kind, const double *lb,
                                const double *ub, char *errbuf, size_t errbuf_len);

/* DualPhase1::new's starting point made on the device (src/solvers/dual/dual_problem.rs:162-214).
 * In: the box problem's standard form (A m x n column-major, c, b, bounds: TwoSided or Fixed only) and the basis
 * the LU of A^T picked (dual_problem.rs:139-160): B_index[m] and N_index[n-m] in the order of the permutation.
 * The engine builds B^-1 and from it y = B^-T c_B, d = c - A^T y, each nonbasic variable's label and value by
 * the sign of d_i (:177-203), b~ = b - A x and x_B = B^-1 b~ (:206-213); read them back with
 * ellp_engine_read_point.  The result is a dual engine ready for ellp_engine_run (phase 1) and, after it,
 * ellp_engine_dual_rephase (phase 2).  A bound of another kind is the reference's panic (ELLP_ERR_PANIC). */
ellp_status ellp_engine_create_dual_phase1(int64_t m, int64_t n, const double *A, const double *c, const double *b,
                                           const uint8_t *bound_kind, const double *lb, const double *ub,
                                           const int64_t *B_index, const int64_t *N_index, const ellp_opts *opts,
                                           ellp_engine **out, char *errbuf, size_t errlen);

/*
 * The dual method's phase-1 -> phase-2 hand-off without leaving HBM (SURVEY.md §8 f2;
 * DualPhase2::from(phase_1), dual_problem.rs:258-404) for the common case that the box problem of phase 1
 * has the same matrix as the original standard form (no TwoSided / Fixed variable was dropped,
 * dual_problem.rs:96-112 — the caller checks that).  With the basis phase 1 ended on and the resident
 * B^-1: y = B^-T c_B, d = c - A^T y, every nonbasic variable to the bound its kind and the sign of d_i
 * name (with the reference's assertions), N re-listed in variable order (columns moved along on the
 * device), x_B = B^-1 (b - A_N x_N), the dual objective, status and counters afresh.  c, kind, lb, ub:
 * n_c entries of the ORIGINAL standard form; b: m entries.  Engines of the explicit-inverse kind only
 * (m > 128): the persistent small-LP kernel carries no inverse — re-create the engine there.
 */
ellp_status ellp_engine_dual_rephase(ellp_engine *e, const double *c, const double *b, const uint8_t *bound_kind,
                                     const double *lb, const double *ub, char *errbuf, size_t errbuf_len);

/*
 * The same sharded loop driven from inside the library, with the exchange done by RCCL directly
 * on the engine's stream (ncclAllGather, in place, seg doubles per rank) — no host language in the
 * per-iteration path.  RCCL is bound at run time (dlopen; `rccl_path` may name the library the
 * process already uses, e.g. torch's copy; NULL = "librccl.so.1"), so a single-GPU user of this
 * library needs no RCCL.  Rank 0 calls ellp_comm_unique_id(), the 128 bytes travel to the other
 * ranks by any means (torch.distributed broadcast in ellp_amd/dist.py), every rank calls
 * ellp_engine_comm_init() (collective), then ellp_engine_run_sharded() (collective: every rank
 * must pass the same max_iters; the ranks take identical decisions, so they issue identical
 * sequences of collectives).  Returns like ellp_engine_run.
 */
#define ELLP_COMM_ID_BYTES 128
ellp_status ellp_comm_unique_id(const char *rccl_path, void *id_out, char *errbuf, size_t errbuf_len);
ellp_status ellp_engine_comm_init(ellp_engine *e, const char *rccl_path, const void *id, int rank, int world,
                                  char *errbuf, size_t errbuf_len);
ellp_status ellp_engine_run_sharded(ellp_engine *e, uint64_t max_iters, ellp_stats *stats, char *errbuf,
                                    size_t errbuf_len);

/*
 * Column-block sharding with SHARDED STORAGE (primal engines): after
 * ellp_engine_shard_columns(e, rank, world) the engine keeps only the nonbasic columns of its own block
 * (the rest of A_N is released) and ellp_engine_run_sharded() exchanges, once per iteration, one small
 * "pack" per rank — its maximal Dantzig key and the at most two candidates within 6 EPS of it, each with
 * its column (64 KB at m = 4000) — instead of the whole pricing output; when ties reach further the loop
 * falls back, for that iteration, to gathering the complete pricing output (exact in every case, see
 * ellp_amd/csrc/engine/ellp_shard.inc).  Transports for the exchange, chosen by what has been set up:
 *   ellp_engine_comm_init          RCCL all-gather on the engine's stream
 *   ellp_engine_mailbox_*          peer-to-peer mailbox: every rank stores its pack directly into every
 *                                  peer's memory (hipIpc-mapped, over xGMI) and raises a flag there; export
 *                                  gives this rank's two IPC handles (2 x 64 bytes: slots, flags), connect
 *                                  takes those of all ranks in rank order (world x 128 bytes)
 *   ellp_engine_set_exchange_callback   the library stages the segments through host memory and calls
 *                                  fn(user, host_buffer, segment_bytes, world) to all-gather them in place
 *                                  (own segment at rank * segment_bytes); for tests and gloo
 * All of them are collective in the usual sense: every rank makes the same calls in the same order.
 */
typedef int (*ellp_exchange_fn)(void *user, void *host_buffer, int64_t segment_bytes, int world);
ellp_status ellp_engine_shard_columns(ellp_engine *e, int rank, int world, char *errbuf, size_t errbuf_len);
ellp_status ellp_engine_set_exchange_callback(ellp_engine *e, ellp_exchange_fn fn, void *user);
#define ELLP_IPC_HANDLE_BYTES 64
ellp_status ellp_engine_mailbox_export(ellp_engine *e, void *handles_out /* 2 x 64 bytes */, char *errbuf, size_t errbuf_len);
ellp_status ellp_engine_mailbox_connect(ellp_engine *e, const void *all_handles /* world x 128 bytes */, char *errbuf,
                                        size_t errbuf_len);
/* one exchange of a test pattern through the mailbox, every word checked; ELLP_OPTIMAL if it arrived intact */
ellp_status ellp_engine_mailbox_selftest(ellp_engine *e, int rounds, char *errbuf, size_t errbuf_len);
/* counters of the sharded loop: [0] iterations that needed the full exchange, [1] column requests,
 * [2] transport (1 RCCL, 2 mailbox, 3 callback), [3] doubles per pack, [4] first own position, [5] one past the last */
ellp_status ellp_engine_shard_info(ellp_engine *e, double *out6);
/* The selection the ranks run on the gathered packs, as a host function (the same source the kernel
 * k_sh_select compiles; no device needed): packs = world * ellp_shard_pack_doubles(ld) doubles, each pack
 * [M_s, count, overflow, 0, then `count` records of (key, N.index, position, r_j, column[ld])].
 * Returns 1 if the packs are not conclusive (the full exchange is needed), 0 with *q = entering position
 * (-1: none) and the (rank, slot) of the pack that holds its column, -1 on bad arguments. */
int ellp_shard_select_compact(const double *packs, int world, int64_t ld, double eps, int64_t *q, int *src_rank,
                              int *src_slot);
int64_t ellp_shard_pack_doubles(int64_t ld);

/*
 * The optional per-iteration objective trace (ellp_opts.trace_len > 0): the last entries of
 * the ring, oldest first: iters_out[k] = the loop body just completed, obj_out[k] = the objective after it
 * (primal: c.x carried by obj += +-lambda r_q per pivot from c.x at engine creation / hand-off; dual: the dual
 * objective as the loop itself carries it, dual…:316).  Returns the number of entries written (<= cap).
 */
int64_t ellp_engine_read_trace(ellp_engine *e, uint64_t *iters_out, double *obj_out, int64_t cap);

/* Debug/parity taps: copy an internal device vector to host. what: see ELLP_TAP_*. Returns
 * the number of doubles written (<= cap) or a negative ellp_status. */
enum { ELLP_TAP_U = 0, ELLP_TAP_R = 1, ELLP_TAP_D = 2, ELLP_TAP_BINV = 3, ELLP_TAP_KEY = 4,
       ELLP_TAP_ALPHA = 5, ELLP_TAP_STATE = 6 /* 12 doubles: status,cur,s_q,s_r,theta_d,delta,lr,ldelta,iters,pivots,lambda,rq;
                                                  cap >= 14: + drift, drift checks; cap >= 20: + maintenance requests
                                                  serviced, Newton-Schulz refreshes, rebuilds, x_B resyncs, last
                                                  refresh residual, launches per primal iteration;
                                                  cap >= 22: + rebuilds settled by the permutation shortcut, setup seconds;
                                                  cap >= 28 (29, 30): + certified hybrid: on?, guarded pivots handed to the exact kernel,
                                                  terminal statuses examined, of those not confirmed, loop bodies run by the
                                                  exact kernel, rebuilds of B^-1 after a hand-over (, solves repeated by the exact
                                                  kernel)(, end points returned NOT certified: a repetition above 1,024 rows that
                                                  ELLP_REDO_MAX_SECONDS ruled out, a singular LU in the certificate) */ };
int64_t ellp_engine_tap(ellp_engine *e, int what, double *dst, int64_t cap);

/* One Newton-Schulz step W <- W + W (I - A_B W) on the resident inverse (two f64 GEMMs); this
 * is what the engine does every `refactor_period` iterations.  Returns max|I - A_B W| measured
 * before the step (NaN on error); if that is >= 1e-4 nothing is changed (rebuild instead). */
double ellp_engine_refresh(ellp_engine *e);

/* Forces a full rebuild of B^-1 from A_B now (used by tests and when a refresh is not safe). */
ellp_status ellp_engine_refactor(ellp_engine *e, char *errbuf, size_t errbuf_len);

/* Test / diagnostic hook: raises the maintenance request a kernel raises after a tiny pivot or a
 * drift-monitor hit (DevState::tiny), as if the last iteration had asked for it.  The next run()/poll()
 * services it: B^-1 is refreshed (and x_B re-checked) before any further iteration. */
ellp_status ellp_engine_request_maintenance(ellp_engine *e);

/* Test hook: multiplies the resident B^-1 by `factor` (a damaged inverse, to exercise the path on
 * which a refresh is refused and the host rebuilds from A_B). */
ellp_status ellp_engine_debug_scale_inverse(ellp_engine *e, double factor);

/* max_ij |(B^-1 A_B - I)_ij| computed on device (drift monitor; tests). */
double ellp_engine_inverse_residual(ellp_engine *e);

void ellp_engine_destroy(ellp_engine *e);

/*
 * SURVEY.md §8 f3 — the rank check of the standard form on the device: column-pivoted Householder
 * QR of A^T (src/standard_form.rs:142 `A.transpose().col_piv_qr()`), reduced to what :143-181
 * consume.  A: m x nv column-major (ld = m) in HOST memory, not modified.  pivot_out[i] = the
 * column of A^T (row of A) swapped into position i at step i (the transposition list),
 * rdiag_out[i] = |R_ii|, both of length min(m, nv).  Default: the host loop's steps with norms and dot products reduced in
 * parallel (fast mode; |R_ii| to rounding, reproducible from run to run) — and, if at any step the two best pivot candidates
 * were different but closer than those reductions can tell apart (1e-12 relative), the factorisation is done again in
 * the exact mode, in which every floating-point result is bitwise what the host loop of ellp_amd/csrc/host/dense.h
 * (ColPivQR) produces (4x slower): the pivot order is the host loop's in every case.  ELLP_QR_EXACT=1 in the environment:
 * exact from the start; ELLP_QR_EXACT=0: fast without the fall-back (measurements).  device < 0: current device.
 */
ellp_status ellp_hip_qr_transposed(int64_t m, int64_t nv, const double *A, int64_t *pivot_out, double *rdiag_out,
                                   int device, char *errbuf, size_t errbuf_len);

/* LU with partial pivoting of A^T on the device: `std_form.A.transpose().lu()` of DualPhase1::new
 * (src/solvers/dual/dual_problem.rs:139-160) reduced to what :141-160 consume.  A: m x nv column-major (the box
 * problem's standard-form matrix), nv >= m.  pivot_out[i] = the row of A^T (= column of A) exchanged with row i at
 * step i (i itself: no exchange; also for a skipped zero column), udiag_out[i] = U_ii, both of length m.  Every
 * floating-point result is bitwise what the host loop of ellp_amd/csrc/host/dense.h (LU) produces.
 * device < 0: current device. */
ellp_status ellp_hip_lu_transposed(int64_t m, int64_t nv, const double *A, int64_t *pivot_out, double *udiag_out,
                                   int device, char *errbuf, size_t errbuf_len);

#ifdef __cplusplus
}
#endif
#endif
