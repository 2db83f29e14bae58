// ellp_engine.hip — MI355X (gfx950) revised-simplex pivot engine behind include/ellp_hip.h.
//
// Replaces the loop bodies of kehlert/ellp 0.2.0
//   src/solvers/primal/primal_simplex_solver.rs:160-235 (+ pivot() :238-435)
//   src/solvers/dual/dual_simplex_solver.rs:188-334
// with hand-written HIP kernels.  Where the reference re-factorises A_B by LU every iteration
// (primal…:173, dual…:241) this engine keeps an explicit B^-1 resident in HBM, updates it by
// a rank-1 (eta) update per pivot and re-derives it from A_B every `refactor_period`
// iterations.  All decisions (entering, leaving, status) are taken on the device; the host
// only enqueues launches and polls one status word every `poll_interval` iterations.
//
// Data layout in HBM (all f64, ld = round_up(m, 16) so every column/row starts 128-B aligned):
//   A_N  : ld x n_N column-major  — nonbasic columns, physically swapped on a pivot exactly as
//          the reference swaps them (primal…:211-217); pricing streams it once per iteration.
//   A_B  : ld x m  column-major  — basic columns (only read by the refactorisation).
//   W[2] : m x ld  ROW-major     — B^-1, two buffers.  Row-major so that FTRAN (d_i = W[i,:].a_q)
//          is one coalesced dot product per row, the dual's rho = row r of B^-1 is one
//          contiguous row, and the eta update streams whole rows.  The eta update reads one
//          buffer and writes the other (ping-pong): no block ever reads a row another block is
//          overwriting, and the pivot row needs no staging copy.
//   u, d, x, c_B, c_N, keys, lambda_i : vectors.
//
// Launch structure — three launches per simplex iteration:
//   primal:  k_price<.,0>   r = c_N - A_N^T u, Dantzig keys, per-block maxima      (HBM-bound;
//                           k_price_wave<0> when A_N is cache-resident)
//            k_ftran2<0>    [entering fold] + d = +-B^-1 a_q + lambda_i per row    (HBM-bound)
//            k_update2<0>   [ratio-test fold] + eta update of B^-1 + bookkeeping   (HBM-bound)
//   dual:    k_price<.,1>   alpha = A_N^T rho, ratios d_j/alpha_j, per-block argmin
//            k_ftran2<1>    [argmin over blocks] + alpha_q = B^-1 a_q
//            k_update2<1>   eta update + d/y/x updates + next leaving row
// The bracketed decision steps are O(m) / O(|N|/cpb) folds.  A separate single-block kernel for
// each costs ~10 us of launch + cold-miss latency (measured), so instead EVERY block of the
// following bandwidth kernel recomputes the same fold in its prologue from inputs that no block
// of that kernel writes (deterministic => all blocks agree), and block 0 alone commits the
// decision to the state for later kernels.  Rule kept throughout: a kernel never reads a state
// field that one of its own blocks writes (except `status`, where a late reader that sees a
// FINAL status simply exits — the result it would have produced is unused; the one non-final
// status, ST_NEED_MAINT, is for that reason never written by the kernel whose blocks must all
// finish: see the comment at its definition).
//
// There is no CPU path in this file: without a HIP device every entry point returns
// ELLP_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is bound with dlopen (see RcclApi)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ellp_hip.h"
#include "ellp_lu_dev.h"

namespace {

constexpr int ST_RUNNING = 100;
// internal: a tiny pivot was taken; refresh B^-1 before going on.  k_update2 only raises
// DevState::tiny — if it wrote the status itself, blocks of the SAME launch that start late (the GPU
// is shared with other solves) would see it at their entry check and skip their rows of the eta
// update.  The leader of the next pricing launch turns the flag into this status, so every later
// kernel of the batch sees it from its first instruction and the rest of the batch is void.
constexpr int ST_NEED_MAINT = 101;
// internal, certified-hybrid engines (ellp_engine::hybrid): the pivot the ratio test chose is below the
// guard (an explicit inverse is good to cond * 2^-53: a structural zero of B^-1 a_q comes out as ~1e-9 on ill-conditioned
// bases, passes the reference's EPS tests, wins a degenerate ratio test and makes the basis singular).  NOTHING of the
// iteration has been committed: every block of k_update2 takes the same decision from the same inputs and returns
// before its first store, so — unlike ST_NEED_MAINT — the leader may write the status itself.  The host hands the
// iteration to the LU-per-iteration kernel (exact_takeover).
constexpr int ST_NEED_EXACT = 102;
constexpr int WAVE = 64;

struct DevState {
    // ---- head: six 16-byte groups that k_price2 fetches with one batch of vector loads (load_phead);
    // keep the groups and their order in step with PHead
    int32_t status;      // [0] ST_RUNNING or an ellp_status
    int32_t nan_flag;
    int32_t panic_code;  // which assert of the reference fired
    int32_t cur;         // index of the current B^-1 buffer (written by k_update2 block 0 / k_ftran_eta block 0 / refactor)
    // snapshot written by k_ftran2 / k_ftran_eta block 0, read by k_update2 / k_price2 (which never write it)
    int32_t s_cur, s_at_lower, s_side;  // [1]
    int32_t tiny;        // maintenance request (tiny pivot, drift monitor, refused refresh), see ST_NEED_MAINT
    int64_t s_q, s_jq;   // [2]
    int64_t s_r;         // [3]
    int64_t s_lv;        // dual: the leaving variable B_index[s_r] as k_ftran2 saw it
    double s_rq, s_lambda0;  // [4]
    int32_t fin;         // [5] k_ftran_eta: final status + 1 (0: none); the next kernel's leader makes it the status
    int32_t usel;        // which of the two u buffers is current
    int32_t s_nbq;       // Nb[s_q] as k_ftran_eta saw it
    int32_t tiny_p;      // k_price2's bookkeeping: tiny pivot; k_ftran_eta relays it into `tiny`
    double s_delta, s_theta_d;
    // ---- dual: leaving row for the coming iteration (written by k_dleave / k_update2<1> block 0)
    int64_t lr;
    double ldelta;
    int32_t lside, l_pad;
    // ---- refactorisation (single-block k_ref_pick -> k_ref_update)
    int32_t do_update;
    int32_t pad_du;
    int64_t r, refk;
    double d_r, alpha_r;
    // ---- results
    double lambda, obj;
    double drift;  // last k_drift_reduce: max|A_B (B^-1 a_q) - a_q| / max|a_q|
    // ---- two-launch pipeline (ellp_lagged.inc)
    int32_t pe_valid;   // k_price2 leader: k_ftran_eta has the eta update of the pivot just committed to apply
    int32_t f_src;      // k_price2 leader: the B^-1 buffer k_ftran_eta reads
    int64_t pe_r;
    double pe_d_r, pe_alpha_r;
    int32_t usel_next;  // the u buffer that becomes current after k_ftran_eta
    int32_t open;       // k_ftran_eta has entered (and counted) an iteration whose ratio test nobody has folded yet
    int32_t pad_open;
    int32_t mv_pending;
    int64_t mv_r;       // deferred half of the column move: A_B[:, mv_r] <- aq_save, c_B[mv_r] <- cq_save
    double cq_save;
    // ---- column-sharded engines (ellp_shard.inc)
    int64_t sh_q;        // k_sh_select: the entering position (-1: none)
    int64_t sh_want_q;   // ST_NEED_COLUMN: the position whose column must be shipped
    double sh_rq;        // reduced cost of sh_q (from the winner's pack)
    double resid;  // last Newton-Schulz refresh: max|I - A_B W| before the step (k_resid_reduce)
    int32_t need_rebuild;  // that residual was too large for a Newton-Schulz step: the host must rebuild
    int32_t pad_nr;
    unsigned long long iters, pivots, flips;
    // ---- partial pricing (opts.partial_segments > 1): the nonbasic positions [pp_lo, pp_hi) are
    // priced, a pass that finds no candidate there moves on to the next segment (pp_skip: the rest of that
    // iteration's kernels do nothing) and pp_P such passes in a row are the optimality test
    int64_t pp_lo, pp_hi, pp_S;
    int32_t pp_P, pp_seg, pp_empty, pp_skip;
    // ---- fused dual iterations (ellp_dualfu.inc): k_dual_fu number dp_seq has left the swap / scalars / next leaving
    // row of its pivot to be done (~0: nothing is left); dp_applied = the last one the following pricing launch did
    unsigned long long dp_seq, dp_applied;
    // ---- steepest-edge pricing (ELLP_FLAG_PRIMAL_STEEPEST_EDGE; ellp_se.inc): what k_update2<0> leaves of the pivot it made
    // for the weight update in the next pricing launch: the entering position, the unsigned pivot element alpha_q[r], the
    // entering variable's weight, and whether d = -alpha_q (entering at its lower bound)
    int32_t se_valid, se_neg;
    int64_t se_q;
    double se_arq, se_gq;
    unsigned long long mbox_gen;  // mailbox transport (ellp_shard.inc): the last exchange that ran
#ifdef ELLP_DBG_STAMPS
    long long dbg[3][4][8];  // [kernel][block selector][stamp] wall_clock64 (100 MHz) — dev builds only
#endif
};

#ifdef ELLP_DBG_STAMPS
#define STAMP(kern, slot)                                                                          \
    do {                                                                                           \
        if (threadIdx.x == 0) {                                                                    \
            const int bsel_ = blockIdx.x == 0 ? 0 : blockIdx.x == 1 ? 1 : blockIdx.x == gridDim.x / 2 ? 2 \
                              : blockIdx.x == gridDim.x - 1 ? 3 : -1;                              \
            if (bsel_ >= 0) a.st->dbg[kern][bsel_][slot] = wall_clock64();                         \
        }                                                                                          \
    } while (0)
#else
#define STAMP(kern, slot) do { } while (0)
#endif

// optional objective trace (ellp_opts.trace_len): ring of (iteration, objective) pairs, written by whichever
// leader completes an iteration
struct Trace {
    double *obj;
    unsigned long long *it;
    int len;
};
__device__ __forceinline__ void trace_put(const Trace &t, unsigned long long iters, double obj) {
    if (t.len <= 0) return;
    const int k = (int)(iters % (unsigned long long)t.len);
    t.obj[k] = obj;
    t.it[k] = iters;
}

// Debug build with bounds assertions (`python -m ellp_amd.build --debug-bounds` -> libellp_hip_dbg.so,
// selected with ELLP_HIP_LIB): every index a decision commits (entering position, leaving row, variable indices) is
// checked against its range by the thread that commits it; a violation stops the loop with ELLP_ERR_PANIC and the
// code of the check.  Compiled out of the product build.
#ifdef ELLP_DEBUG_BOUNDS
#define ELLP_CHECK(st, cond, code)               \
    do {                                         \
        if (!(cond)) {                           \
            (st)->panic_code = (code);           \
            (st)->status = ELLP_ERR_PANIC;       \
        }                                        \
    } while (0)
#else
#define ELLP_CHECK(st, cond, code) \
    do {                           \
    } while (0)
#endif

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt) but NOT
// for outstanding global loads (vmcnt), so register prefetches issued earlier stay in flight
// across it.  __syncthreads() carries a fence that drains vmcnt as well.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}

// ---- wave-wide max / min without the LDS crossbar (`__shfl_xor` is ds_bpermute on this target: ~100 cycles
// per step): four DPP butterfly steps inside each row of 16 lanes, then the four rows through readlane.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// v must not be NaN
__device__ __forceinline__ double wave_allmax_dpp(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = fmax(v, dpp_f64<0x140>(v));  // row_mirror
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ int wave_allmin_dpp(int v) {
    v = min(v, dpp_i32<0xB1>(v));
    v = min(v, dpp_i32<0x4E>(v));
    v = min(v, dpp_i32<0x141>(v));
    v = min(v, dpp_i32<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// one wave per column: a column with exactly one nonzero entry is recorded under its variable's index
__global__ __launch_bounds__(256) void k_scan_singletons(const double *A, int64_t ld, int64_t m, int64_t ncols, const int64_t *index,
                                                         int32_t *vs_row, double *vs_val) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const double *c = A + j * ld;
    int cnt = 0, row = -1;
    double val = 0.0;
    for (int64_t i = lane; i < m; i += 64) {
        const double v = c[i];
        if (v != 0.0) {  // a NaN counts: such a column is never a unit column
            cnt += (v == v) ? 1 : 2;
            row = (int)i;
            val = v;
        }
    }
    int tot = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    const unsigned long long has = __ballot(cnt > 0);
    if (tot == 1) {
        const int l = __ffsll((long long)has) - 1;
        if (lane == l) {
            const int64_t v = index[j];
            vs_row[v] = row;
            vs_val[v] = val;
        }
    }
}

// ------------------------------------------------------------------ pricing output / exchange buffer
// The pricing kernels write their per-column and per-block results into ONE buffer laid out in
// `world` equal segments, one per rank (world = 1 on a single GPU):
//   segment s = [ blockkey (nbs) | blockpos (nbs, stored as f64) | key (nbs*cpb) | r (nbs*cpb) ]
// where rank s prices the blocks [s*nbs, (s+1)*nbs), i.e. nonbasic positions
// [s*nbs*cpb, (s+1)*nbs*cpb).  With column-block sharding each rank fills its own
// segment and ONE all-gather of the buffer per iteration gives every rank the complete
// pricing result; everything downstream is replicated and deterministic.
struct Xchg {
    double *X;
    int64_t seg;  // doubles per segment = 2*nbs + 2*nbs*cpb
    int nbs, cpb;
    __device__ __forceinline__ double &bk(int b) const { return X[(int64_t)(b / nbs) * seg + (b % nbs)]; }
    __device__ __forceinline__ double &bp(int b) const { return X[(int64_t)(b / nbs) * seg + nbs + (b % nbs)]; }
    __device__ __forceinline__ double &key(int64_t j) const {
        const int64_t per = (int64_t)nbs * cpb, s = j / per;
        return X[s * seg + 2 * nbs + (j - s * per)];
    }
    __device__ __forceinline__ double &r(int64_t j) const {
        const int64_t per = (int64_t)nbs * cpb, s = j / per;
        return X[s * seg + 2 * nbs + per + (j - s * per)];
    }
};

// ------------------------------------------------------------------ pricing
// MODE 0 (primal): r_j = c_N[j] - A_N[:,j].u for every nonbasic column (primal…:189), fused with
// the eligibility filter / Dantzig key of pivot() (primal…:253-270) and a per-block key maximum.
// MODE 1 (dual): alpha_j = A_N[:,j].rho (dual…:255) with rho = row lr of B^-1, fused with the dual
// ratio d[N_j]/alpha_j over eligible columns (dual…:263-278) and a per-block FIRST argmin.
//
// One block = 256 threads = 4 waves streams `cpb` consecutive columns.  The 256 threads stride
// down a column 16 B per lane (perfectly coalesced 4 KiB per block-instruction); the slice of
// u each thread needs is loaded once into registers (T double2 per thread) and re-used for all
// the block's columns; four columns are in flight at a time so every lane keeps 4*T 16-byte
// loads outstanding.  HBM-bound: 8*ld bytes per column, 2 flops per 8 bytes.
struct PriceArgs {
    const double *A_N;
    const double *W0, *W1;  // dual: rho = row st->lr of the current B^-1 buffer (dual…:248-253)
    const double *u;        // primal: u
    const double *c_N;      // primal only
    const uint8_t *Nb;
    const int64_t *N_index;
    const double *dd;       // dual: reduced costs d (indexed by variable)
    Xchg xc;                // out: r / alpha, key, per-block max (primal) or first argmin (dual)
    DevState *st;
    int64_t ld, nN;
    int cpb;
    int block0;             // first pricing block of this rank
    double eps;
    int pp_on;              // partial pricing: only positions [st->pp_lo, st->pp_hi) may enter
    // unit columns ("sparse A", first step).  Every LP in standard form carries them: a slack per
    // inequality row (standard_form.rs:115-136) and, in phase 1, an artificial per row (primal_problem.rs:236-246) —
    // 29-57 % of the nonbasic columns of config 3.  vs_row[v] = the row of variable v's ONLY nonzero (-1: a general
    // column), vs_val[v] its value; made once per engine from the matrix (k_scan_singletons; the matrix never
    // changes, positions do: the table is indexed by VARIABLE, N_index leads to it).  The primal pricing kernels
    // form such a column's dot product as vs_val * u[vs_row] instead of streaming 8 ld bytes of zeros — the same
    // number the stream gives (every other term is an exact zero), so the pivots are the same.  null: off.
    const int32_t *vs_row;
    const double *vs_val;
    // dual engines whose fused iterations (ellp_dualfu.inc) are closed by the NEXT pricing launch instead of a block
    // of their own: what that takes (dp_seq == 0: not this engine)
    unsigned long long dp_seq;
    const int32_t *dp_lrow;
    const double *dp_ldelta, *dp_lside, *dp_d;
    int dp_nrb;
    int dp_maxviol;         // ELLP_FLAG_DUAL_MAX_VIOLATION: the leaving row of largest violation instead of the first violated one
    const double *rho_ovr;  // dual, certificate above 1,024 rows (ellp_exact.inc): rho from a fresh LU instead of row lr of B^-1
    double *dp_A_N, *dp_A_B, *dp_c_B, *dp_c_N, *dp_x, *dp_dd;
    int64_t *dp_B_index, *dp_N_index;
    uint8_t *dp_Nb;
    Trace dp_trace;
};

// ---- closing a fused dual iteration (dual…:313-333 and the choice of the next leaving row, dual…:200-236)
// the first row block of k_dual_fu that recorded a violation: its row (-1: none), delta and side; all 256 threads.
// Row, delta and side of every record are fetched in ONE round trip, so that the choice does not cost a second
// dependent one (this sits on the critical path of every pricing launch that closes a fused iteration).
struct DualLeave {
    long long lr;
    double delta;
    int side;
};
__device__ __forceinline__ DualLeave dual_first_violation(const int32_t *lrow, const double *ldelta, const double *lside, int nrb,
                                                          int tid, long long *s_tmp, int maxviol = 0) {
    __shared__ double s_ld[4], s_ad[4];
    __shared__ int s_lr[4], s_ls[4];
    const int lane = tid & 63, wave = tid >> 6;
    int bb = 0x7fffffff, brow = -1, bside = 0;
    double bdelta = 0.0, bad = -1.0;  // maxviol (ELLP_FLAG_DUAL_MAX_VIOLATION): the record of largest |delta|, first of equals
    for (int b0 = 0; b0 < nrb; b0 += 1024) {  // four records per thread in flight
        int v[4];
        double dl[4], sd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + tid + 256 * u;
            const int bc = b < nrb ? b : 0;
            v[u] = lrow[bc];
            dl[u] = ldelta[bc];
            sd[u] = lside[bc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + tid + 256 * u;
            if (!(b < nrb && v[u] >= 0)) continue;
            double ad = fabs(dl[u]);
            if (ad != ad) ad = INFINITY;
            const bool take = maxviol ? (ad > bad || (ad == bad && b < bb)) : (b < bb);
            if (take) {
                bb = b;
                brow = v[u];
                bdelta = dl[u];
                bside = (int)sd[u];
                bad = ad;
            }
        }
    }
    const double wad = maxviol ? wave_allmax_dpp(bad) : 0.0;
    const int wb = wave_allmin_dpp((!maxviol || bad == wad) ? bb : 0x7fffffff);
    if (bb == wb && wb != 0x7fffffff && (!maxviol || bad == wad)) {  // one lane per wave (the blocks a wave's lanes hold are distinct)
        s_lr[wave] = brow;
        s_ld[wave] = bdelta;
        s_ls[wave] = bside;
    }
    if (lane == 0) {
        s_tmp[wave] = wb;
        s_ad[wave] = wad;
    }
    lds_barrier();
    int best = (int)s_tmp[0], bw = 0;
    double bestad = s_ad[0];
    for (int w = 1; w < 4; ++w) {
        const int cb = (int)s_tmp[w];
        const bool take = maxviol ? (cb != 0x7fffffff && (best == 0x7fffffff || s_ad[w] > bestad || (s_ad[w] == bestad && cb < best)))
                                  : (cb < best);
        if (take) {
            best = cb;
            bestad = s_ad[w];
            bw = w;
        }
    }
    DualLeave o;
    if (best == 0x7fffffff) {
        o.lr = -1;
        o.delta = 0.0;
        o.side = 0;
    } else {
        o.lr = s_lr[bw];
        o.delta = s_ld[bw];
        o.side = s_ls[bw];
    }
    return o;
}
// the column / index / cost swap of the pivot (q, r); by all 256 threads of ONE block
__device__ __forceinline__ void dual_close_swap(double *A_N, double *A_B, double *c_B, double *c_N, double *dd,
                                                int64_t *N_index, uint8_t *Nb, const DevState *st, int64_t ld, int tid) {
    const int64_t q = st->s_q, r = st->s_r, jq = st->s_jq, lv = st->s_lv;
    const double theta_d = st->s_theta_d;
    double2 *cn = reinterpret_cast<double2 *>(A_N + q * ld);
    double2 *cb = reinterpret_cast<double2 *>(A_B + r * ld);
    for (int64_t t = tid; t < (ld >> 1); t += 256) {
        const double2 x = cn[t];
        cn[t] = cb[t];
        cb[t] = x;
    }
    if (tid == 0) {
        dd[lv] = -theta_d;
        dd[jq] = 0.0;
        N_index[q] = lv;
        Nb[q] = (uint8_t)st->s_side;
    } else if (tid == 64) {
        const double tc = c_N[q];
        c_N[q] = c_B[r];
        c_B[r] = tc;
    }
}
// the scalars of the pivot and the next leaving row; by ONE thread
__device__ __forceinline__ void dual_close_scalars(DevState *st, const double *d, double *x, int64_t *B_index, const Trace &trace,
                                                   long long lr, double ldelta, int lside) {
    const int64_t r = st->s_r, jq = st->s_jq;
    const double theta_d = st->s_theta_d, delta = st->s_delta;
    const double d_r = d[r];
    const double theta_p = delta / d_r;
    x[jq] = x[jq] + theta_p;
    B_index[r] = jq;
    const double obj = st->obj + theta_d * delta;
    const unsigned long long it = st->iters + 1;
    st->obj = obj;
    st->cur = st->s_cur ^ 1;
    st->pivots += 1;
    st->iters = it;
    trace_put(trace, it, obj);
    if (d_r != d_r || theta_p != theta_p) st->status = ELLP_ERR_NAN;
    st->lr = lr;
    if (lr >= 0) {
        st->ldelta = ldelta;
        st->lside = lside;
    }
}
// MODE 1 prologue of the pricing kernels: the leaving row, its delta and the current B^-1 buffer — from the state, or,
// when the fused iteration in front of this launch is still open, from its records, finishing that iteration on the way
// (the block that owns position q swaps the columns before it prices them; block 0 books the scalars)
__device__ __forceinline__ void dual_price_prologue(const PriceArgs &a, DevState *st, int gb, int tid, long long *s_tmp,
                                                    int64_t *lr_out, double *ldelta_out, int *cur_out) {
    // everything the prologue reads from the state in one batch of loads (each separate test in front of a branch
    // is a round trip of its own)
    const unsigned long long open_seq = st->dp_seq;
    const int64_t st_lr = st->lr, st_q = st->s_q;
    const double st_ldelta = st->ldelta;
    const int st_cur = st->cur, st_scur = st->s_cur;
    __builtin_amdgcn_sched_barrier(0);
    const bool pend = a.dp_seq != 0 && open_seq + 1 == a.dp_seq;
    if (!pend) {
        *lr_out = st_lr;
        *ldelta_out = st_ldelta;
        *cur_out = st_cur;
        return;
    }
    const DualLeave lv = dual_first_violation(a.dp_lrow, a.dp_ldelta, a.dp_lside, a.dp_nrb, tid, s_tmp, a.dp_maxviol);
    const long long lr = lv.lr;
    const double ldelta = lv.delta;
    const int lside = lv.side;
    if ((int64_t)gb == st_q / a.cpb) {
        dual_close_swap(a.dp_A_N, a.dp_A_B, a.dp_c_B, a.dp_c_N, a.dp_dd, a.dp_N_index, a.dp_Nb, st, a.ld, tid);
        __syncthreads();  // the block prices the swapped column next
    }
    if (blockIdx.x == 0 && tid == 0) {
        dual_close_scalars(st, a.dp_d, a.dp_x, a.dp_B_index, a.dp_trace, lr, ldelta, lside);
        st->dp_applied = st->dp_seq;
    }
    *lr_out = lr;
    *ldelta_out = ldelta;
    *cur_out = st_scur ^ 1;
}

// 16-byte column load; NT = non-temporal.  When A_N is much larger than the 256 MiB Infinity Cache
// (config 5: 1.4 GB) it is streamed once per iteration and should not displace B^-1, which the
// other two kernels re-read: nt loads measured +10 % on the pricing pass itself at that size and
// -4 % when A_N is cache-resident (tools/price_bench.hip), so the host picks per problem size.
template <bool NT>
__device__ __forceinline__ double2 load_col2(const double2 *p) {
    if (NT) {
        double2 r;
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
        return r;
    }
    return *p;
}

template <int T, int MODE, bool NT>
__global__ __launch_bounds__(256) void k_price(PriceArgs a) {
    __shared__ double s_part[2][4][4];
    __shared__ double s_k[4];
    __shared__ long long s_p[4];
    DevState *st = a.st;
    if (st->status != ST_RUNNING) return;
    if (st->tiny) {  // raised by the previous k_update2: stop the batch here (see ST_NEED_MAINT)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->status = ST_NEED_MAINT;
            __threadfence();
            st->tiny = 0;
        }
        return;
    }
    STAMP(0, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t half = a.ld >> 1;
    double sgn = 1.0;
    const double2 *u2;
    if (MODE == 0) {
        u2 = reinterpret_cast<const double2 *>(a.u);
    } else {
        int64_t lr;
        double ldelta;
        int cur;
        dual_price_prologue(a, st, a.block0 + (int)blockIdx.x, tid, s_p, &lr, &ldelta, &cur);
        if (lr < 0) {  // no primal-infeasible basic: optimal (dual…:243-246)
            if (blockIdx.x == 0 && tid == 0) {
                st->iters += 1;
                st->status = ELLP_OPTIMAL;
            }
            return;
        }
        sgn = (ldelta < 0.0) ? -1.0 : 1.0;
        u2 = reinterpret_cast<const double2 *>(a.rho_ovr ? a.rho_ovr : (cur ? a.W1 : a.W0) + lr * a.ld);
    }
    double2 ur[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t idx = tid + 256 * t;
        ur[t] = idx < half ? u2[idx] : make_double2(0.0, 0.0);
    }
    const int gb = a.block0 + (int)blockIdx.x;  // global pricing block
    const int64_t j0 = (int64_t)gb * a.cpb;
    const int64_t j1 = (j0 + a.cpb < a.nN) ? j0 + a.cpb : a.nN;
    if (j0 >= a.nN) return;  // a rank without pricing blocks still launches one block for the flag above
    int64_t plo = 0, phi = a.nN;
    if (MODE == 0 && a.pp_on) {  // partial pricing: a block outside the segment has nothing to offer
        plo = st->pp_lo;
        phi = st->pp_hi;
        if (j1 <= plo || j0 >= phi) {
            if (tid == 0) a.xc.bk(gb) = -INFINITY;
            return;
        }
    }
    double best = (MODE == 0) ? -INFINITY : INFINITY;
    long long bestpos = -1;
    int buf = 0;
    // unit columns (PriceArgs::vs_row): every wave fetches the flags of the block's columns itself (lane l: column j0 + l)
    const bool use_vs = a.vs_row != nullptr;  // primal: u[row]; dual: rho[row] — one product instead of a column of zeros
    int sgl = -1;
    double svl = 0.0;
    if (use_vs && j0 + lane < j1) {
        const int64_t v = a.N_index[j0 + lane];
        sgl = a.vs_row[v];
        svl = a.vs_val[v];
    }
    for (int64_t j = j0; j < j1; j += 4, buf ^= 1) {
        const int ncol = (int)((j1 - j) < 4 ? (j1 - j) : 4);
        // epilogue inputs of the (up to 4) column-owner threads: issued before the column stream
        // so that their dependent gathers (N_index -> d) are back when the dot products are
        int nb_pre = 0;
        double cd_pre = 0.0;  // primal: c_N[j] ; dual: d[N_index[j]]
        {
            const int64_t jo = j + (tid < ncol ? tid : 0);
            nb_pre = a.Nb[jo];
            cd_pre = (MODE == 0) ? a.c_N[jo] : a.dd[a.N_index[jo]];
        }
        int sg[4] = {-1, -1, -1, -1};
        bool all_unit = use_vs;
        int dense0 = -1;  // a general column of the group: unit columns are pointed at it (no traffic of their own)
        if (use_vs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < ncol) {
                    sg[k] = __builtin_amdgcn_readlane(sgl, (int)(j - j0) + k);
                    if (sg[k] < 0) {
                        all_unit = false;
                        if (dense0 < 0) dense0 = k;
                    }
                }
            }
        }
        // the column-owner threads (tid < ncol, all in wave 0): their own column's flag, value and entry of u
        // (the shuffles run on all lanes: a lane that sits out cannot be read from)
        const int sg_sh = __shfl(sgl, ((int)(j - j0) + (tid & 3)) & 63);
        const double my_sv = __shfl(svl, ((int)(j - j0) + (tid & 3)) & 63);
        const int my_sg = (use_vs && tid < ncol) ? sg_sh : -1;
        const double u_unit = my_sg >= 0 ? reinterpret_cast<const double *>(u2)[my_sg] : 0.0;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (!all_unit) {
            const double2 *col[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int64_t jj = (j + k < j1) ? j + k : j1 - 1;  // clamp: stay in bounds, no branch
                if (use_vs && k < ncol && sg[k] >= 0) jj = j + dense0;
                col[k] = reinterpret_cast<const double2 *>(a.A_N + jj * a.ld);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t idx = tid + 256 * t;
                if (idx < half) {
                    double2 v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = load_col2<NT>(col[k] + idx);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[k] = fma(v[k].x, ur[t].x, acc[k]);
                        acc[k] = fma(v[k].y, ur[t].y, acc[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = wave_sum(acc[k]);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_part[buf][wave][k] = acc[k];
        }
        __syncthreads();
        if (tid < ncol) {
            const int k = tid;
            double dot =
                ((s_part[buf][0][k] + s_part[buf][1][k]) + s_part[buf][2][k]) + s_part[buf][3][k];
            if (my_sg >= 0) dot = my_sv * u_unit;
            const int64_t jj = j + k;
            const int nb = nb_pre;
            if (MODE == 0) {
                const double rj = cd_pre - dot;
                double key = -INFINITY;
                if (rj != rj) {
                    st->nan_flag = 1;
                } else if (!(fabs(rj) < a.eps)) {
                    const bool pos = rj > 0.0;
                    if (pos && nb == ELLP_NB_UPPER) key = rj;
                    else if (!pos && nb == ELLP_NB_LOWER) key = -rj;
                    else if (nb == ELLP_NB_FREE) key = fabs(rj);
                }
                if (jj < plo || jj >= phi) key = -INFINITY;  // a column of a boundary block outside the segment
                a.xc.r(jj) = rj;
                a.xc.key(jj) = key;
                best = fmax(best, key);
            } else {
                a.xc.r(jj) = dot;  // alpha (un-negated, dual…:286-288 restores the sign anyway)
                const double al = sgn * dot;
                bool keep;
                if (nb == ELLP_NB_LOWER) keep = al > a.eps;
                else if (nb == ELLP_NB_UPPER) keep = al < -a.eps;
                else keep = true;
                if (keep) {
                    const double ratio = cd_pre / al;
                    if (ratio != ratio) st->nan_flag = 1;
                    if (bestpos < 0 || ratio < best) {  // strict '<' keeps the FIRST minimum
                        best = ratio;
                        bestpos = jj;
                    }
                }
            }
        }
    }
    // combine the (up to) four column-owner threads; they all live in wave 0
    if (MODE == 0) {
        if (wave == 0) {
            double v = (lane < 4) ? best : -INFINITY;
            v = fmax(v, __shfl_xor(v, 1));
            v = fmax(v, __shfl_xor(v, 2));
            if (lane == 0) a.xc.bk(gb) = v;
        }
    } else {
        if (tid < 4) {
            s_k[tid] = best;
            s_p[tid] = bestpos;
        }
        __syncthreads();
        if (tid == 0) {
            double bk = INFINITY;
            long long bp = -1;
            for (int k = 0; k < 4; ++k) {
                if (s_p[k] < 0) continue;
                if (bp < 0 || s_k[k] < bk || (s_k[k] == bk && s_p[k] < bp)) {
                    bk = s_k[k];
                    bp = s_p[k];
                }
            }
            a.xc.bk(gb) = bk;
            a.xc.bp(gb) = (double)bp;
        }
    }
    STAMP(0, 1);
}

// Wave-per-column variant for the cache-resident regime (8*ld*|N| within reach of the 256 MiB
// Infinity Cache): the block's four waves each own column pairs j0+2w, j0+2w+1 (+8, ...), every
// lane keeps 2 x 4 16-byte column loads in flight and re-reads its slice of u from L1/L2, and there
// is no block barrier inside the stream — only one at the end to combine the four waves' results.
// tools/price_bench.hip: 17.3 us against 18.3 us for the block-per-column-group shape at C3 (and no
// gain once A_N streams from HBM, where k_price<.,.,true> is used).  Same outputs as k_price.
template <int MODE>
__global__ __launch_bounds__(256) void k_price_wave(PriceArgs a) {
    __shared__ double s_k[4];
    __shared__ long long s_p[4];
    DevState *st = a.st;
    if (st->status != ST_RUNNING) return;
    if (st->tiny) {  // raised by the previous k_update2: stop the batch here (see ST_NEED_MAINT)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->status = ST_NEED_MAINT;
            __threadfence();
            st->tiny = 0;
        }
        return;
    }
    STAMP(0, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t half = a.ld >> 1;
    double sgn = 1.0;
    const double2 *u2;
    if (MODE == 0) {
        u2 = reinterpret_cast<const double2 *>(a.u);
    } else {
        int64_t lr;
        double ldelta;
        int cur;
        dual_price_prologue(a, st, a.block0 + (int)blockIdx.x, tid, s_p, &lr, &ldelta, &cur);
        if (lr < 0) {  // no primal-infeasible basic: optimal (dual…:243-246)
            if (blockIdx.x == 0 && tid == 0) {
                st->iters += 1;
                st->status = ELLP_OPTIMAL;
            }
            return;
        }
        sgn = (ldelta < 0.0) ? -1.0 : 1.0;
        u2 = reinterpret_cast<const double2 *>(a.rho_ovr ? a.rho_ovr : (cur ? a.W1 : a.W0) + lr * a.ld);
    }
    const int gb = a.block0 + (int)blockIdx.x;  // global pricing block
    const int64_t j0 = (int64_t)gb * a.cpb;
    const int64_t j1 = (j0 + a.cpb < a.nN) ? j0 + a.cpb : a.nN;
    if (j0 >= a.nN) return;  // a rank without pricing blocks still launches one block for the flag above
    int64_t plo = 0, phi = a.nN;
    if (MODE == 0 && a.pp_on) {  // partial pricing: a block outside the segment has nothing to offer
        plo = st->pp_lo;
        phi = st->pp_hi;
        if (j1 <= plo || j0 >= phi) {
            if (tid == 0) a.xc.bk(gb) = -INFINITY;
            return;
        }
    }
    double best = (MODE == 0) ? -INFINITY : INFINITY;  // lanes 0 and 1 own the pair's two columns
    long long bestpos = -1;
    // unit columns (PriceArgs::vs_row): lane l holds the flag of column j0 + l
    const bool use_vs = a.vs_row != nullptr;  // primal: u[row]; dual: rho[row] — one product instead of a column of zeros
    int sgl = -1;
    double svl = 0.0;
    if (use_vs && j0 + lane < j1) {
        const int64_t v = a.N_index[j0 + lane];
        sgl = a.vs_row[v];
        svl = a.vs_val[v];
    }
    for (int64_t j = j0 + 2 * wave; j < j1; j += 8) {
        const int ncol = (j1 - j) < 2 ? 1 : 2;
        int nb_pre = 0;
        double cd_pre = 0.0;  // primal: c_N[j] ; dual: d[N_index[j]]
        {
            const int64_t jo = j + (lane < ncol ? lane : 0);
            nb_pre = a.Nb[jo];
            cd_pre = (MODE == 0) ? a.c_N[jo] : a.dd[a.N_index[jo]];
        }
        const int64_t jb_ = (j + 1 < j1 ? j + 1 : j);
        const int s0 = use_vs ? __builtin_amdgcn_readlane(sgl, (int)(j - j0)) : -1;
        const int s1 = use_vs ? __builtin_amdgcn_readlane(sgl, (int)(jb_ - j0)) : -1;
        const bool stream = !(s0 >= 0 && s1 >= 0);
        // a unit column next to a general one is pointed at its neighbour: no traffic of its own
        const double2 *c0 = reinterpret_cast<const double2 *>(a.A_N + ((s0 >= 0) ? jb_ : j) * a.ld);
        const double2 *c1 = reinterpret_cast<const double2 *>(a.A_N + ((s1 >= 0) ? j : jb_) * a.ld);
        double acc0 = 0.0, acc1 = 0.0;
        for (int64_t t0 = lane; stream && t0 < half; t0 += 4 * WAVE) {
            double2 uu[4], v0[4], v1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t t = t0 + WAVE * k;
                const int64_t tc = t < half ? t : 0;  // clamped, unconditional
                const double2 uv = u2[tc];
                uu[k] = t < half ? uv : make_double2(0.0, 0.0);
                v0[k] = c0[tc];
                v1[k] = c1[tc];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = fma(v0[k].x, uu[k].x, acc0);
                acc0 = fma(v0[k].y, uu[k].y, acc0);
                acc1 = fma(v1[k].x, uu[k].x, acc1);
                acc1 = fma(v1[k].y, uu[k].y, acc1);
            }
        }
        if (stream) {
            acc0 = wave_sum(acc0);
            acc1 = wave_sum(acc1);
        }
        if (s0 >= 0) acc0 = readlane_f64(svl, (int)(j - j0)) * reinterpret_cast<const double *>(u2)[s0];
        if (s1 >= 0) acc1 = readlane_f64(svl, (int)(jb_ - j0)) * reinterpret_cast<const double *>(u2)[s1];
        if (lane < ncol) {
            const double dot = lane == 0 ? acc0 : acc1;
            const int64_t jj = j + lane;
            const int nb = nb_pre;
            if (MODE == 0) {
                const double rj = cd_pre - dot;
                double key = -INFINITY;
                if (rj != rj) {
                    st->nan_flag = 1;
                } else if (!(fabs(rj) < a.eps)) {
                    const bool pos = rj > 0.0;
                    if (pos && nb == ELLP_NB_UPPER) key = rj;
                    else if (!pos && nb == ELLP_NB_LOWER) key = -rj;
                    else if (nb == ELLP_NB_FREE) key = fabs(rj);
                }
                if (jj < plo || jj >= phi) key = -INFINITY;  // a column of a boundary block outside the segment
                a.xc.r(jj) = rj;
                a.xc.key(jj) = key;
                best = fmax(best, key);
            } else {
                a.xc.r(jj) = dot;  // alpha (un-negated, dual…:286-288 restores the sign anyway)
                const double al = sgn * dot;
                bool keep;
                if (nb == ELLP_NB_LOWER) keep = al > a.eps;
                else if (nb == ELLP_NB_UPPER) keep = al < -a.eps;
                else keep = true;
                if (keep) {
                    const double ratio = cd_pre / al;
                    if (ratio != ratio) st->nan_flag = 1;
                    if (bestpos < 0 || ratio < best) {  // this lane's columns come in increasing position
                        best = ratio;
                        bestpos = jj;
                    }
                }
            }
        }
    }
    // combine: lanes 0/1 of each wave, then the four waves (lexicographic (ratio, position) for the dual)
    if (MODE == 0) {
        double v = (lane < 2) ? best : -INFINITY;
        v = fmax(v, __shfl_xor(v, 1));
        if (lane == 0) s_k[wave] = v;
        __syncthreads();
        if (tid == 0) a.xc.bk(gb) = fmax(fmax(s_k[0], s_k[1]), fmax(s_k[2], s_k[3]));
    } else {
        double k0 = (lane < 2) ? best : INFINITY;
        long long p0 = (lane < 2) ? bestpos : -1;
        const double k1 = __shfl_xor(k0, 1);
        const long long p1 = __shfl_xor(p0, 1);
        if (p1 >= 0 && (p0 < 0 || k1 < k0 || (k1 == k0 && p1 < p0))) {
            k0 = k1;
            p0 = p1;
        }
        if (lane == 0) {
            s_k[wave] = k0;
            s_p[wave] = p0;
        }
        __syncthreads();
        if (tid == 0) {
            double bk = INFINITY;
            long long bp = -1;
            for (int k = 0; k < 4; ++k) {
                if (s_p[k] < 0) continue;
                if (bp < 0 || s_k[k] < bk || (s_k[k] == bk && s_p[k] < bp)) {
                    bk = s_k[k];
                    bp = s_p[k];
                }
            }
            a.xc.bk(gb) = bk;
            a.xc.bp(gb) = (double)bp;
        }
    }
    STAMP(0, 1);
}

// ------------------------------------------------------------------ primal entering fold
// Exact emulation of the reference's sequential `max_by` fold (primal…:271-287): candidate
// (key, N.index) replaces the accumulator iff NOT(|acc-key| >= EPS ? acc > key : acc.index >
// index).  The comparator is not transitive, so the fold cannot be a tree reduction.
//
// fold_elements(): one wave feeds the 64-column chunk of one pricing block through the fold.
// entering_fold_full(): walks ALL pricing blocks in position order, using ballots to skip every
//   block whose maximum key is <= acc - EPS (it cannot change the accumulator).
// Fast path (k_ftran2).  Let M be the maximum key, T = M - 4 EPS, and suppose NO key lies in the
//   gap (T - 2 EPS, T].  Then every element above T beats every element below the gap by more
//   than EPS and never ties with it.  So in the reference's fold (i) the first above-T element
//   replaces whatever the accumulator was (or finds it empty) — the state after it is that
//   element exactly; (ii) from then on the accumulator stays above T, and a below-gap element
//   can neither beat it nor tie with it, i.e. it is never an event.  Hence folding only the
//   blocks that contain an above-T element, in order, from an empty accumulator, ends in the
//   same state (below-gap elements inside those blocks are harmless: one can only be the
//   accumulator before the first above-T element, which replaces it).  The gap is CHECKED —
//   per block on the block maxima, per element on the blocks that are opened — and the full
//   walk runs if it is not clean.  Typically one block is opened, fetched in one round trip.
struct FoldState {
    bool have;
    double racc;
    long long iacc, qacc;
};
__device__ __forceinline__ void fold_elements(FoldState &f, double k, long long idx, int64_t jb, double eps,
                                              int lane) {
    int efrom = 0;
    for (;;) {
        bool ev = lane >= efrom && k > -INFINITY;
        if (ev && f.have) {
            const bool acc_greater = (fabs(f.racc - k) >= eps) ? (f.racc > k) : (f.iacc > idx);
            ev = !acc_greater;
        }
        const unsigned long long em = __ballot(ev);
        if (!em) break;
        const int l = __ffsll((long long)em) - 1;
        f.racc = __shfl(k, l);
        f.iacc = __shfl(idx, l);
        f.qacc = jb + l;
        f.have = true;
        efrom = l + 1;
    }
}

__device__ __forceinline__ long long entering_fold_full(const double *s_bk, int nblocks, int cpb, int64_t nN,
                                                        const Xchg &xc, const int64_t *N_index, double eps,
                                                        int lane) {
    FoldState f{false, 0.0, 0, -1};
    for (int g0 = 0; g0 < nblocks; g0 += WAVE) {
        const double bm = (g0 + lane < nblocks) ? s_bk[g0 + lane] : -INFINITY;
        int from = 0;
        for (;;) {
            // a block matters iff it may hold a key with racc - key < EPS.  Written as a DIFFERENCE: racc - EPS
            // is absorbed once racc exceeds ~4e6 (ulp > EPS) and `bm > racc - eps` would then skip bm == racc
            const bool pred = lane >= from && bm > -INFINITY && (!f.have || f.racc - bm < eps);
            const unsigned long long mask = __ballot(pred);
            if (!mask) break;
            const int bl = __ffsll((long long)mask) - 1;
            const int64_t jb = (int64_t)(g0 + bl) * cpb;
            const int64_t j = jb + lane;
            const bool valid = lane < cpb && j < nN;
            const double k = valid ? xc.key(j) : -INFINITY;
            const long long idx = valid ? N_index[j] : 0;
            fold_elements(f, k, idx, jb, eps, lane);
            from = bl + 1;
        }
    }
    return f.qacc;
}

constexpr int FC_SLOTS = 16;

// ------------------------------------------------------------------ FTRAN (+ decision prologue)
// d = +-B^-1 a_q (primal…:295-300, dual…:294): one wave per row of the row-major W, 16 B per
// lane.  The row loads do not depend on the entering column, so each wave issues its (first)
// row into registers BEFORE the decision prologue and the HBM stream overlaps the fold; the
// column a_q (8*ld bytes, L2-resident) is fetched once q is known.
// Prologue, every block: MODE 0 the entering fold over the pricing output, MODE 1 the dual
// ratio argmin over the per-block minima (dual…:279, min_by keeps the FIRST minimum).
// Epilogue MODE 0: the wave that produced d_i also evaluates the ratio lambda_i of basic row i
// by bound kind (primal…:320-367) so that k_update2's fold only reads three flat arrays.
#include "ellp_shard_select.inc"

struct Ftran2Args {
    const double *W0, *W1, *A_N;
    Xchg xc;
    const int64_t *N_index, *B_index;
    const uint8_t *Nb, *kind;
    const double *x, *lb, *ub;
    double *d, *lam;
    int32_t *bidx;
    uint8_t *dpos;
    DevState *st;
    int64_t m, ld, nN;
    int nblocks, cpb;
    double eps;
    const double *aq_cur;  // column-sharded engines: the entering position is st->sh_q (k_sh_select), its column is here
    int pp_on;             // partial pricing (DevState::pp_*)
    // column-sharded engines, compact exchange: the selection runs HERE, in every block's prologue, on the gathered packs
    // (null: k_sh_select has run); block 0 leaves the column in aq_out for k_update2 and commits the mailbox generation
    const double *sel_packs;
    double *aq_out;
    int sel_world, mbox_commit;
};

// NT = double2 per lane that hold one row (ceil(ld/128)); NT == 0: rows are streamed instead
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k_ftran2(Ftran2Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ long long s_q;
    __shared__ double s_wk[4];
    __shared__ long long s_wp[4];
    __shared__ int s_nh, s_band;
    __shared__ int s_hblk[FC_SLOTS];
    __shared__ double s_hkey[FC_SLOTS * 64];
    __shared__ int32_t s_hidx[FC_SLOTS * 64];
    DevState *st = a.st;
    if (st->status != ST_RUNNING) return;
    STAMP(1, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nan_flag = st->nan_flag;
    const int cur = st->cur;
    const double *W = cur ? a.W1 : a.W0;
    const int64_t half = a.ld >> 1;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nwaves = (int64_t)gridDim.x * 4;

    constexpr int NTR = NT > 0 ? NT : 1;
    double2 wrow[NTR];
    int64_t bi0 = 0;
    double xi0 = 0.0, lbi0 = 0.0, ubi0 = 0.0;
    int k0 = 0;
    double theta_d = 0.0;
    if (MODE == 0 && a.aq_cur) {
        // column-sharded engine: the selection was made by k_sh_select from the gathered packs
        const int64_t irow = wave_global < a.m ? wave_global : 0;
        if (NT > 0) {
            const double2 *row = reinterpret_cast<const double2 *>(W + irow * a.ld);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int64_t t = lane + u * WAVE;
                const double2 wv = row[t < half ? t : 0];
                wrow[u] = t < half ? wv : make_double2(0.0, 0.0);
            }
        }
        bi0 = a.B_index[irow];
        xi0 = a.x[bi0];
        k0 = a.kind[bi0];
        lbi0 = a.lb[bi0];
        ubi0 = a.ub[bi0];
        if (a.sel_packs) {
            if (wave == 0) {
                const WaveSelect ws = shard_select_wave(a.sel_packs, a.sel_world, a.ld, a.eps, lane);
                if (lane == 0) {
                    s_q = ws.verdict ? -2 : ws.q;
                    s_nh = ws.src_rank;
                    s_band = ws.src_c;
                }
            }
            lds_barrier();
            if (blockIdx.x == 0 && tid == 0) {
                if (a.mbox_commit) st->mbox_gen += 1;  // the packs have been consumed (also when they were not conclusive)
                if (s_q == -2) st->status = ST_NEED_FULL;
            }
            if (s_q == -2) return;
        } else {
            if (tid == 0) s_q = st->sh_q;
            lds_barrier();
        }
    } else if (MODE == 0) {
        double *s_bk = reinterpret_cast<double *>(smem);
        double tmax = -INFINITY;
        double v0[4];  // first (normally only) batch of block maxima: issued before the row prefetch
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = tid + 256 * u;
            const double t = a.xc.bk(b < a.nblocks ? b : 0);  // clamped, unconditional
            v0[u] = b < a.nblocks ? t : -INFINITY;
        }
        // the ratio-test gathers hang off B_index: fetch it WITH the staging loads, so that the
        // dependent x/kind/lb/ub loads below wait for it alone and not for the whole row batch
        bi0 = a.B_index[wave_global < a.m ? wave_global : 0];
        // ---- this wave's first row (and its ratio-test inputs): independent of the decision, so
        // issue them now (after the small staging loads, which must retire first) and let the
        // HBM stream overlap the fold.  sched_barrier: vmcnt retires in issue order, so the
        // scheduler must not hoist this big batch above the loads the prologue waits for.
        __builtin_amdgcn_sched_barrier(0);
        {
            // unconditional (clamped) so that every path has the same number of loads in flight —
            // otherwise the waits below degrade to vmcnt(0)
            const int64_t irow = wave_global < a.m ? wave_global : 0;
            if (NT > 0) {
                const double2 *row = reinterpret_cast<const double2 *>(W + irow * a.ld);
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int64_t t = lane + u * WAVE;
                    const double2 wv = row[t < half ? t : 0];
                    wrow[u] = t < half ? wv : make_double2(0.0, 0.0);
                }
            }
            if (MODE == 0) {
                xi0 = a.x[bi0];
                k0 = a.kind[bi0];
                lbi0 = a.lb[bi0];
                ubi0 = a.ub[bi0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (tid + 256 * u < a.nblocks) s_bk[tid + 256 * u] = v0[u];
            tmax = fmax(tmax, v0[u]);
        }
        for (int b0 = tid + 4 * 256; b0 < a.nblocks; b0 += 4 * 256) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + 256 * u;
                const double t = a.xc.bk(b < a.nblocks ? b : 0);
                v[u] = b < a.nblocks ? t : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b0 + 256 * u < a.nblocks) s_bk[b0 + 256 * u] = v[u];
                tmax = fmax(tmax, v[u]);
            }
        }
        if (tid == 0) {
            s_nh = 0;
            s_band = 0;
        }
        tmax = wave_allmax_dpp(tmax);  // DPP butterfly + readlane (no NaN here: keys are never NaN)
        if (lane == 0) s_wk[wave] = tmax;
        lds_barrier();
        STAMP(1, 1);
        const double M = fmax(fmax(s_wk[0], s_wk[1]), fmax(s_wk[2], s_wk[3]));
        // H = blocks within 4 EPS of the maximum; band = blocks in the next 2 EPS
        for (int b = tid; b < a.nblocks; b += 256) {
            const double v = s_bk[b];
            // distances from the maximum, never M - 4 EPS itself: for keys above ~4e6 that expression is
            // absorbed (M - 4e-10 == M), no block would qualify and the fold would report "no candidate"
            if (M - v < 4.0 * a.eps) {
                const int slot = atomicAdd(&s_nh, 1);
                if (slot < FC_SLOTS) s_hblk[slot] = b;
            } else if (M - v < 6.0 * a.eps) {
                s_band = 1;
            }
        }
        lds_barrier();
        const int nh = s_nh;
        const bool fast = M > -INFINITY && nh <= FC_SLOTS && !s_band;
        if (fast) {
            if (tid == 0) {  // order the (few) H blocks by position
                for (int i = 1; i < nh; ++i) {
                    const int v = s_hblk[i];
                    int j = i - 1;
                    while (j >= 0 && s_hblk[j] > v) {
                        s_hblk[j + 1] = s_hblk[j];
                        --j;
                    }
                    s_hblk[j + 1] = v;
                }
            }
            lds_barrier();
            for (int sl = wave; sl < nh; sl += 4) {  // all waves fetch the H blocks: one round trip
                const int64_t j = (int64_t)s_hblk[sl] * a.cpb + lane;
                const bool ok = lane < a.cpb && j < a.nN;
                const int64_t jc = ok ? j : 0;
                const double kl = a.xc.key(jc);
                const int64_t il = a.N_index[jc];
                const double kv = ok ? kl : -INFINITY;
                s_hkey[sl * 64 + lane] = kv;
                s_hidx[sl * 64 + lane] = ok ? (int32_t)il : 0;
                if (M - kv < 6.0 * a.eps && !(M - kv < 4.0 * a.eps)) s_band = 1;  // an element in the gap
            }
            lds_barrier();
        }
        if (wave == 0) {
            long long q = -1;
            bool done = false;
            if (fast && !s_band) {
                FoldState f{false, 0.0, 0, -1};
                for (int sl = 0; sl < nh; ++sl)
                    fold_elements(f, s_hkey[sl * 64 + lane], s_hidx[sl * 64 + lane], (int64_t)s_hblk[sl] * a.cpb,
                                  a.eps, lane);
                q = f.qacc;
                done = true;
            }
            if (!done && M > -INFINITY)
                q = entering_fold_full(s_bk, a.nblocks, a.cpb, a.nN, a.xc, a.N_index, a.eps, lane);
            if (lane == 0) s_q = q;
        }
        lds_barrier();
    } else {
        if (st->lr < 0) return;  // k_price<.,1> block 0 has already reported Optimal
        // ---- this wave's first row (and its ratio-test inputs): independent of the decision, so
        // issue them now (after the small staging loads, which must retire first) and let the
        // HBM stream overlap the fold.  sched_barrier: vmcnt retires in issue order, so the
        // scheduler must not hoist this big batch above the loads the prologue waits for.
        __builtin_amdgcn_sched_barrier(0);
        {
            // unconditional (clamped) so that every path has the same number of loads in flight —
            // otherwise the waits below degrade to vmcnt(0)
            const int64_t irow = wave_global < a.m ? wave_global : 0;
            if (NT > 0) {
                const double2 *row = reinterpret_cast<const double2 *>(W + irow * a.ld);
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int64_t t = lane + u * WAVE;
                    const double2 wv = row[t < half ? t : 0];
                    wrow[u] = t < half ? wv : make_double2(0.0, 0.0);
                }
            }
            if (MODE == 0) {
                bi0 = a.B_index[irow];
                xi0 = a.x[bi0];
                k0 = a.kind[bi0];
                lbi0 = a.lb[bi0];
                ubi0 = a.ub[bi0];
            }
        }
        double bk = INFINITY;
        long long bp = -1;
        for (int b = tid; b < a.nblocks; b += 256) {
            const long long p = (long long)a.xc.bp(b);
            if (p < 0) continue;
            const double k = a.xc.bk(b);
            if (bp < 0 || k < bk || (k == bk && p < bp)) {
                bk = k;
                bp = p;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ok = __shfl_xor(bk, o);
            const long long op = __shfl_xor(bp, o);
            if (op >= 0 && (bp < 0 || ok < bk || (ok == bk && op < bp))) {
                bk = ok;
                bp = op;
            }
        }
        if (lane == 0) {
            s_wk[wave] = bk;
            s_wp[wave] = bp;
        }
        lds_barrier();
        bk = s_wk[0];
        bp = s_wp[0];
        for (int w = 1; w < 4; ++w)
            if (s_wp[w] >= 0 && (bp < 0 || s_wk[w] < bk || (s_wk[w] == bk && s_wp[w] < bp))) {
                bk = s_wk[w];
                bp = s_wp[w];
            }
        if (tid == 0) s_q = bp;
        theta_d = (st->ldelta < 0.0) ? -bk : bk;  // dual…:286-289
        lds_barrier();
    }
    const long long q = s_q;
    STAMP(1, 2);
    if (nan_flag) {
        if (blockIdx.x == 0 && tid == 0) st->status = ELLP_ERR_NAN;
        return;
    }
    if (q < 0) {
        if (blockIdx.x == 0 && tid == 0) {
            st->iters += 1;
            if (MODE == 0 && a.pp_on && st->pp_empty + 1 < st->pp_P) {
                // partial pricing: nothing in this segment — the next pass prices the next one; only pp_P empty
                // passes in a row (every segment, no pivot in between) are the reference's `None => Optimal`
                const int seg = (st->pp_seg + 1) % st->pp_P;
                const int64_t lo = (int64_t)seg * st->pp_S;
                st->pp_seg = seg;
                st->pp_lo = lo;
                st->pp_hi = lo + st->pp_S < a.nN ? lo + st->pp_S : a.nN;
                st->pp_empty = st->pp_empty + 1;
                st->pp_skip = 1;
            } else {
                st->status = (MODE == 0) ? ELLP_OPTIMAL      // primal…:289-292
                                         : ELLP_INFEASIBLE;  // dual unbounded, dual…:281-284
            }
        }
        return;
    }
    if (MODE == 0 && a.pp_on && blockIdx.x == 0 && tid == 0) {
        st->pp_empty = 0;
        st->pp_skip = 0;
    }
    const int at_lower = (MODE == 0) ? (a.Nb[q] == ELLP_NB_LOWER ? 1 : 0) : 0;
    const double sgn = at_lower ? -1.0 : 1.0;
    const double *sel_rec = (MODE == 0 && a.aq_cur && a.sel_packs)
                                ? a.sel_packs + (int64_t)s_nh * pack_doubles(a.ld) + SH_HDR + (int64_t)s_band * (SH_REC + a.ld)
                                : nullptr;  // the winner's record: key, N.index, position, r_q, then its column
    const double2 *col = reinterpret_cast<const double2 *>(sel_rec ? sel_rec + SH_REC : ((MODE == 0 && a.aq_cur) ? a.aq_cur : a.A_N + q * a.ld));
    bool first = true;
    for (int64_t i = wave_global; i < a.m; i += nwaves, first = false) {
        const double2 *row = reinterpret_cast<const double2 *>(W + i * a.ld);
        int64_t bi = bi0;
        double xi = xi0, lbi = lbi0, ubi = ubi0;
        int k = k0;
        if (MODE == 0 && !first) {
            bi = a.B_index[i];
            xi = a.x[bi];
            k = a.kind[bi];
            lbi = a.lb[bi];
            ubi = a.ub[bi];
        }
        double acc0 = 0.0, acc1 = 0.0;
        if (NT > 0 && first) {
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int64_t t = lane + u * WAVE;
                const double2 c = t < half ? col[t] : make_double2(0.0, 0.0);
                acc0 = fma(wrow[u].x, c.x, acc0);
                acc1 = fma(wrow[u].y, c.y, acc1);
            }
        } else {
            for (int64_t t0 = lane; t0 < half; t0 += 8 * WAVE) {
                double2 w[8], c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t t = t0 + u * WAVE;
                    w[u] = t < half ? row[t] : make_double2(0.0, 0.0);
                    c[u] = t < half ? col[t] : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc0 = fma(w[u].x, c[u].x, acc0);
                    acc1 = fma(w[u].y, c[u].y, acc1);
                }
            }
        }
        const double di = sgn * wave_sum(acc0 + acc1);
        if (lane == 0) {
            a.d[i] = di;
            if (MODE == 0) {
                double li = INFINITY;
                if (!(fabs(di) < a.eps)) {
                    if (k == ELLP_BOUND_FREE) {
                        li = INFINITY;
                    } else if (k == ELLP_BOUND_LOWER) {
                        if (di > 0.0) li = INFINITY;
                        else if (xi > lbi) li = (lbi - xi) / di;
                        else li = 0.0;
                    } else if (k == ELLP_BOUND_UPPER) {
                        if (di > 0.0) li = (xi < ubi) ? (ubi - xi) / di : 0.0;
                        else li = INFINITY;
                    } else if (k == ELLP_BOUND_TWOSIDED) {
                        if (di > 0.0) li = (xi < ubi) ? (ubi - xi) / di : 0.0;
                        else if (xi < lbi) li = (lbi - xi) / di;  // quirk Q1 (primal…:359)
                        else li = 0.0;
                    } else {
                        li = 0.0;  // Fixed
                    }
                    if (li != li) st->nan_flag = 1;
                }
                a.lam[i] = li;
                a.bidx[i] = (int32_t)bi;
                a.dpos[i] = di > 0.0 ? 1 : 0;
            }
        }
    }
    STAMP(1, 3);
    if (sel_rec && blockIdx.x == 0) {  // k_update2 and the drift monitor take the entering column from aq_cur
        double2 *dst = reinterpret_cast<double2 *>(a.aq_out);
        for (int64_t t = tid; t < half; t += 256) dst[t] = col[t];
    }
    if (blockIdx.x == 0 && tid == 0) {  // commit the decision for k_update2
        ELLP_CHECK(st, q >= 0 && q < a.nN, 9101);
        ELLP_CHECK(st, MODE == 0 || (st->lr >= 0 && st->lr < a.m), 9102);
        st->s_cur = cur;
        st->s_q = q;
        st->s_at_lower = at_lower;
        if (MODE == 0) {
            const int64_t jq = a.N_index[q];
            st->s_jq = jq;
            st->s_rq = sel_rec ? sel_rec[3] : (a.aq_cur ? st->sh_rq : a.xc.r(q));
            const int kk = a.kind[jq];  // primal…:305-311
            st->s_lambda0 = (kk == ELLP_BOUND_TWOSIDED) ? a.ub[jq] - a.lb[jq] : (kk == ELLP_BOUND_FIXED ? 0.0 : INFINITY);
        } else {
            st->s_jq = a.N_index[q];
            st->s_r = st->lr;
            st->s_lv = a.B_index[st->lr];
            st->s_delta = st->ldelta;
            st->s_side = st->lside;
            st->s_theta_d = theta_d;
        }
    }
}

// ------------------------------------------------------------------ primal ratio-test fold (one wave)
// Exact emulation of the sequential fold of primal…:379-399, including the quirk that
// `new_basic_index` is only written in the tie branch.  chunkmin[c] = min lambda over rows
// 64c..64c+63; a chunk whose minimum is >= lambda + EPS cannot change the state.
struct RatioResult {
    double lambda;
    long long nb;
    int side;
};
__device__ __forceinline__ RatioResult ratio_fold(const double *chunkmin, int nchunks, int64_t m, const double *lam,
                                                  const int32_t *bidx, const uint8_t *dpos, double lambda0,
                                                  double eps, int lane) {
    double lambda = lambda0;
    long long nb = -1;
    int side = ELLP_NB_LOWER;
    bool have_nbi = false;
    int32_t nbi = 0;
    for (int g0 = 0; g0 < nchunks; g0 += WAVE) {
        const double cm = (g0 + lane < nchunks) ? chunkmin[g0 + lane] : INFINITY;
        int from = 0;
        for (;;) {
            // skip a chunk only if every lambda_i in it is at least EPS above lambda (difference form:
            // lambda + eps is absorbed for lambda above ~4e6 and a tie at lambda would be skipped)
            const bool pred = lane >= from && !(cm - lambda >= eps);
            const unsigned long long mask = __ballot(pred);
            if (!mask) break;
            const int cl = __ffsll((long long)mask) - 1;
            const int64_t i = (int64_t)(g0 + cl) * 64 + lane;
            const double li = (i < m) ? lam[i] : INFINITY;
            const int32_t b_i = (i < m) ? bidx[i] : 0;
            const int dp = (i < m) ? dpos[i] : 0;
            int efrom = 0;
            for (;;) {
                const bool strict = li < lambda - eps;
                const bool tie = !strict && fabs(li - lambda) < eps && (!have_nbi || b_i < nbi);
                const bool ev = lane >= efrom && (strict || tie);
                const unsigned long long em = __ballot(ev);
                if (!em) break;
                const int l = __ffsll((long long)em) - 1;
                const int was_tie = __shfl((int)tie, l);
                lambda = __shfl(li, l);
                nb = (long long)(g0 + cl) * 64 + l;
                side = __shfl(dp, l) ? ELLP_NB_UPPER : ELLP_NB_LOWER;
                if (was_tie) {
                    have_nbi = true;
                    nbi = __shfl(b_i, l);
                }
                efrom = l + 1;
            }
            from = cl + 1;
        }
    }
    return RatioResult{lambda, nb, side};
}

// 16-byte store of a rewritten row of B^-1.  NT (k_update2): non-temporal — the 8*m*ld bytes a pivot
// rewrites are not read again before the next k_ftran2, and as ordinary stores they sit dirty in the
// L2s until the end-of-kernel write-back (measured at C3: k_update2 19.3 -> 17.9 us, k_ftran2 +0.6).
// The refactorisation kernels re-read their rows at once and keep ordinary stores.
template <bool NT>
__device__ __forceinline__ void store_row2(double2 *p, double2 v) {
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
    } else {
        *p = v;
    }
}

// rows [row0, row0+nrows) (nrows <= 4) of the eta update:
// dst[i,:] = src[i,:] - (d_i/d_r) * src[r,:]   (i != r),   dst[r,:] = src[r,:] / alpha_r.
// The (up to) four rows go through together: every thread keeps four independent 16-byte loads
// in flight and re-uses its chunk of the pivot row for all of them.  dv[k] = d[row0+k] is loaded
// by the caller ahead of time (it does not depend on the pivot row).
constexpr int UPD_ROWS = 4;
template <bool NT>
__device__ __forceinline__ void eta_update_rows(const double *src, double *dst, int64_t m, int64_t ld, int64_t r,
                                                const double dv[UPD_ROWS], double d_r, double alpha_r,
                                                int64_t row0, int nrows, int tid) {
    const int64_t half = ld >> 1;
    const double2 *rho2 = reinterpret_cast<const double2 *>(src + r * ld);
    double f[UPD_ROWS];
    int64_t ii[UPD_ROWS];
#pragma unroll
    for (int k = 0; k < UPD_ROWS; ++k) {
        const int64_t i = row0 + k;
        const bool ok = i < m && k < nrows;
        ii[k] = ok ? i : -1;
        f[k] = (ok && i != r) ? -(dv[k] / d_r) : 0.0;
    }
    for (int64_t t = tid; t < half; t += 256) {
        const double2 p = rho2[t];
        double2 w[UPD_ROWS];
#pragma unroll
        for (int k = 0; k < UPD_ROWS; ++k)
            if (ii[k] >= 0) w[k] = reinterpret_cast<const double2 *>(src + ii[k] * ld)[t];
#pragma unroll
        for (int k = 0; k < UPD_ROWS; ++k) {
            if (ii[k] < 0) continue;
            double2 o;
            if (ii[k] == r) {
                o = make_double2(p.x / alpha_r, p.y / alpha_r);
            } else {
                o.x = fma(f[k], p.x, w[k].x);
                o.y = fma(f[k], p.y, w[k].y);
            }
            store_row2<NT>(reinterpret_cast<double2 *>(dst + ii[k] * ld) + t, o);
        }
    }
}

__device__ __forceinline__ void swap_columns(double *A_N, double *A_B, int64_t q, int64_t r, int64_t ld, int tid) {
    double2 *cn = reinterpret_cast<double2 *>(A_N + q * ld);
    double2 *cb = reinterpret_cast<double2 *>(A_B + r * ld);
    for (int64_t t = tid; t < (ld >> 1); t += 256) {
        const double2 x = cn[t];
        cn[t] = cb[t];
        cb[t] = x;
    }
}

// first basic position (in B order) whose variable violates a bound by more than EPS
// (dual…:200-236); Free and Fixed basics never leave (quirk Q3)
__device__ __forceinline__ bool dual_violation(const int64_t *B_index, const double *x, const uint8_t *kind,
                                               const double *lb, const double *ub, double eps, int64_t i,
                                               double *delta, int *side) {
    const int64_t bi = B_index[i];
    const double xi = x[bi];
    const int k = kind[bi];
    if (k == ELLP_BOUND_LOWER) {
        if (xi < lb[bi] - eps) { *delta = xi - lb[bi]; *side = ELLP_NB_LOWER; return true; }
    } else if (k == ELLP_BOUND_UPPER) {
        if (xi > ub[bi] + eps) { *delta = xi - ub[bi]; *side = ELLP_NB_UPPER; return true; }
    } else if (k == ELLP_BOUND_TWOSIDED) {
        if (xi > ub[bi] + eps) { *delta = xi - ub[bi]; *side = ELLP_NB_UPPER; return true; }
        if (xi < lb[bi] - eps) { *delta = xi - lb[bi]; *side = ELLP_NB_LOWER; return true; }
    }
    return false;
}
// block-wide (256 threads) search for the leaving row; commits lr/ldelta/lside (lr = -1: none).  The reference takes the
// FIRST violated basic position (dual…:200-236); maxviol (ELLP_FLAG_DUAL_MAX_VIOLATION, an extension): the one with the
// largest violation, the first of equals.
__device__ __forceinline__ void find_leaving(const int64_t *B_index, const double *x, const uint8_t *kind,
                                             const double *lb, const double *ub, double eps, int64_t m, int tid,
                                             long long *s_tmp /*[4]*/, DevState *st, int maxviol = 0) {
    __shared__ double s_fad[4];
    const int lane = tid & 63, wave = tid >> 6;
    long long best = INT64_MAX;
    double bad = -1.0;
    double dl;
    int sd;
    for (int64_t i = tid; i < m; i += 256) {
        if (dual_violation(B_index, x, kind, lb, ub, eps, i, &dl, &sd)) {
            if (!maxviol) {
                best = i;
                break;  // increasing i per thread: the first hit is this thread's minimum
            }
            double ad = fabs(dl);
            if (ad != ad) ad = INFINITY;
            if (ad > bad) {
                bad = ad;
                best = i;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long ob = __shfl_xor(best, o);
        const double oa = __shfl_xor(bad, o);
        const bool take = maxviol ? (oa > bad || (oa == bad && ob < best)) : (ob < best);
        if (take) {
            best = ob;
            bad = oa;
        }
    }
    if (lane == 0) {
        s_tmp[wave] = best;
        s_fad[wave] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            const bool take = maxviol ? (s_fad[w] > bad || (s_fad[w] == bad && s_tmp[w] < best)) : (s_tmp[w] < best);
            if (take) {
                best = s_tmp[w];
                bad = s_fad[w];
            }
        }
        if (best == INT64_MAX) {
            st->lr = -1;
        } else {
            double delta = 0.0;
            int side = 0;
            dual_violation(B_index, x, kind, lb, ub, eps, best, &delta, &side);
            st->lr = best;
            st->ldelta = delta;
            st->lside = side;
        }
    }
}

// ------------------------------------------------------------------ eta update (+ decision prologue, + bookkeeping)
// MODE 0 prologue, every block: stage lambda_i / variable index / sign(d_i) (written by
// k_ftran2) into LDS, per-64 minima, then wave 0 runs the ratio-test fold.  Then all blocks
// stream their rows of the eta update (16*m*ld bytes read+write in total).  Block 0 finally
// applies what the reference does after pivot(): x update (primal…:408-417), index / cost /
// column swap (primal…:205-221) or bound flip (:223-231), and the O(m) BTRAN update
// u += (r_q/alpha_r) * rho (u = B^-T c_B, primal…:184-187, maintained incrementally).
// MODE 1: no fold (the leaving row was fixed before pricing); block 0 applies dual…:296-333 and
// searches the next leaving row.
struct Update2Args {
    double *W0, *W1;
    const double *d;     // primal: d ; dual: alpha_q
    const double *lam;
    const int32_t *bidx;
    const uint8_t *dpos;
    Xchg xc;              // dual: alpha_j = xc.r(j)
    double *u, *u_alt, *A_N, *A_B, *c_B, *c_N, *x, *y, *dd;  // u_alt: second u buffer (DevState::usel, two-launch pipeline)
    const double *lb, *ub;
    const uint8_t *kind;
    int64_t *B_index, *N_index;
    uint8_t *Nb;
    DevState *st;
    int64_t m, ld, nN;
    int rows_per_block;
    int update_u;
    int stage_lds;
    double ill_tol;   // > 0: after a pivot with |alpha_r| < ill_tol * max|alpha| stop for maintenance
    double guard_abs; // > 0 (certified hybrid): a pivot with |alpha_r| < guard_abs is NOT taken, the loop stops with ST_NEED_EXACT
    double eps;
    const double *aq_cur;   // column-sharded engines: the entering column (A_N holds only positions [own0, own1))
    int64_t own0, own1;
    int count_iter;         // 0: closing kernel of the two-launch pipeline (k_ftran_eta has counted the iteration)
    int maxviol;            // dual: ELLP_FLAG_DUAL_MAX_VIOLATION (find_leaving)
    const double *se_gamma; // primal, steepest edge: weights by nonbasic position (null: off)
    double *se_rho;         // ... the pivot row of the inverse this pivot was decided with, kept for the weight update
    double *se_vpart;       // ... per row block: sum_i d_i * (row i of the inverse FTRAN used) over the block's rows — the partial
                            // sums of v = B^-T d (k_se_vreduce adds them in block order); null: v comes from k_btran_part / _reduce
    Trace trace;
};

// NR = double2 per thread per row (ceil(ld/512)); NR == 0: rows are streamed after the fold
template <int MODE, int NR>
__global__ __launch_bounds__(256) void k_update2(Update2Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_lambda;
    __shared__ long long s_nb;
    __shared__ int s_side;
    __shared__ long long s_tmp[4];
    DevState *st = a.st;
    if (st->status != ST_RUNNING) return;
    if (MODE == 0 && st->fin) {  // closing kernel of the two-launch pipeline: k_ftran_eta found the end of the solve
        if (blockIdx.x == 0 && threadIdx.x == 0) st->status = st->fin - 1;
        return;
    }
    if (MODE == 0 && st->pp_skip) return;  // partial pricing: this pass found no candidate in its segment (k_ftran2)
    STAMP(2, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m = a.m;
    const int cur = st->s_cur;
    const int64_t q = st->s_q;
    const double *src = cur ? a.W1 : a.W0;
    double *dst = cur ? a.W0 : a.W1;
    int64_t r;
    double lambda = 0.0;
    int side = 0;
    // the first NBK blocks do bookkeeping only (primal 2: block 0 x / the basic side / the counters,
    // block 1 u / the column swap / the nonbasic side; dual 3: block 0 the O(|N|) reduced-cost
    // update, block 1 y / the column swap / the nonbasic side, block 2 x / the basic side / the next
    // leaving row); blocks NBK.. stream rows_per_block rows each
    constexpr int NBK = MODE == 0 ? 2 : 3;
    const bool row_block = blockIdx.x >= NBK;
    const int64_t row0 = ((int64_t)blockIdx.x - NBK) * a.rows_per_block;
    double dv[UPD_ROWS];
#pragma unroll
    for (int k = 0; k < UPD_ROWS; ++k)
        dv[k] = (row_block && k < a.rows_per_block && row0 + k < m) ? a.d[row0 + k] : 0.0;
    // rows_per_block may be up to 2*UPD_ROWS (large m: half as many blocks re-stage the fold's inputs
    // and re-read the pivot row); rows UPD_ROWS.. of the block are streamed after the first group
    double dv2[UPD_ROWS];
#pragma unroll
    for (int k = 0; k < UPD_ROWS; ++k)
        dv2[k] = (row_block && UPD_ROWS + k < a.rows_per_block && row0 + UPD_ROWS + k < m) ? a.d[row0 + UPD_ROWS + k] : 0.0;
    constexpr int NRR = NR > 0 ? NR : 1;
    double2 wreg[UPD_ROWS][NRR];
    const int64_t halfw = a.ld >> 1;
    if (MODE == 0) {
        const int nchunks = (int)((m + 63) >> 6);
        double *chunkmin = reinterpret_cast<double *>(smem);
        const double *lam = a.lam;
        const int32_t *bidx = a.bidx;
        const uint8_t *dpos = a.dpos;
        double *l_lam = chunkmin + nchunks;
        int32_t *l_bidx = reinterpret_cast<int32_t *>(l_lam + (a.stage_lds ? m : 0));
        uint8_t *l_dpos = reinterpret_cast<uint8_t *>(l_bidx + (a.stage_lds ? m : 0));
        // each wave takes chunks wave, wave+4, ... ; 8 chunks (24 independent loads) per pass
        bool rows_issued = false;
        for (int c0 = wave; c0 < nchunks || !rows_issued; c0 += 32) {
            double v[8];
            int32_t bi[8];
            uint8_t dp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // unconditional loads from a clamped index (a predicated load becomes a branch
                // with its own wait: eight serialized round trips instead of one)
                const int64_t i = (int64_t)(c0 + 4 * k) * 64 + lane;
                const bool ok = (c0 + 4 * k) < nchunks && i < m;
                const int64_t ic = ok ? i : 0;
                const double lv = lam[ic];
                bi[k] = bidx[ic];
                dp[k] = dpos[ic];
                v[k] = ok ? lv : INFINITY;
            }
            if (!rows_issued) {
                rows_issued = true;
                // The rows this block rewrites do not depend on which row pivots: issue them into
                // registers now (behind the small staging loads) so the HBM stream overlaps the fold.
                // (Tried and not better: all rows only after the staging data has arrived, half of them
                // before and half after, non-temporal row loads.)
                __builtin_amdgcn_sched_barrier(0);
                if (NR > 0) {  // every block, bookkeeping blocks included (clamped): same loads in flight on all paths
#pragma unroll
                    for (int k = 0; k < UPD_ROWS; ++k) {
                        const int64_t i = row0 + k;
                        const bool ok = row_block && k < a.rows_per_block && i < m;
                        const double2 *srow = reinterpret_cast<const double2 *>(src + (ok ? i : 0) * a.ld);
#pragma unroll
                        for (int u = 0; u < NR; ++u) {
                            const int64_t t = tid + 256 * u;
                            const double2 wv = srow[t < halfw ? t : 0];
                            wreg[k][u] = (ok && t < halfw) ? wv : make_double2(0.0, 0.0);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + 4 * k;
                if (c >= nchunks) continue;
                const int64_t i = (int64_t)c * 64 + lane;
                if (a.stage_lds && i < m) {
                    l_lam[i] = v[k];
                    l_bidx[i] = bi[k];
                    l_dpos[i] = dp[k];
                }
                const double cm = wave_min(v[k]);
                if (lane == 0) chunkmin[c] = cm;
            }
        }
        lds_barrier();
        STAMP(2, 1);
        if (wave == 0) {
            const RatioResult rr = a.stage_lds
                                       ? ratio_fold(chunkmin, nchunks, m, l_lam, l_bidx, l_dpos, st->s_lambda0, a.eps, lane)
                                       : ratio_fold(chunkmin, nchunks, m, lam, bidx, dpos, st->s_lambda0, a.eps, lane);
            if (lane == 0) {
                s_lambda = rr.lambda;
                s_nb = rr.nb;
                s_side = rr.side;
            }
        }
        lds_barrier();
        lambda = s_lambda;
        r = s_nb;
        side = s_side;
        STAMP(2, 2);
        const bool leader = blockIdx.x == 0 && tid == 0;
        if (st->nan_flag) {
            if (leader) st->status = ELLP_ERR_NAN;
            return;
        }
        if (!(lambda >= 0.0)) {  // primal…:402
            if (leader) {
                st->panic_code = 402;
                st->status = ELLP_ERR_PANIC;
            }
            return;
        }
        if (isinf(lambda)) {  // primal…:404-406
            if (leader) {
                if (a.count_iter) st->iters += 1;
                st->status = ELLP_UNBOUNDED;
            }
            return;
        }
    } else {
        r = st->s_r;
        // The rows this block rewrites do not depend on which row pivots: issue them into
        // registers now (behind the small staging loads) so the HBM stream overlaps the fold.
        __builtin_amdgcn_sched_barrier(0);
        if (NR > 0) {  // every block, block 0 included (clamped): same loads in flight on all paths
#pragma unroll
            for (int k = 0; k < UPD_ROWS; ++k) {
                const int64_t i = row0 + k;
                const bool ok = row_block && k < a.rows_per_block && i < m;
                const double2 *srow = reinterpret_cast<const double2 *>(src + (ok ? i : 0) * a.ld);
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int64_t t = tid + 256 * u;
                    const double2 wv = srow[t < halfw ? t : 0];
                    wreg[k][u] = (ok && t < halfw) ? wv : make_double2(0.0, 0.0);
                }
            }
        }
    }

    const int at_lower = st->s_at_lower;
    double d_r = 0.0, alpha_r = 0.0;
    if (r >= 0) {
        d_r = a.d[r];
        alpha_r = (MODE == 0 && at_lower) ? -d_r : d_r;
        if (a.guard_abs > 0.0 && fabs(d_r) < a.guard_abs) {  // see ST_NEED_EXACT: every block returns, nothing is stored
            if (blockIdx.x == 0 && tid == 0) st->status = ST_NEED_EXACT;
            return;
        }
        if (row_block) {
            if (NR > 0) {
                const double2 *rho2 = reinterpret_cast<const double2 *>(src + r * a.ld);
                double2 pr[NRR];
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int64_t t = tid + 256 * u;
                    const double2 pv = rho2[t < halfw ? t : 0];
                    pr[u] = t < halfw ? pv : make_double2(0.0, 0.0);
                }
                if (MODE == 0 && a.se_vpart) {
                    // steepest edge: this block's share of v = B^-T d, from the rows of the OLD inverse it holds anyway (the
                    // separate transposed GEMV re-read all of B^-1: 8 m ld bytes and two launches per iteration)
                    double2 *vp = reinterpret_cast<double2 *>(a.se_vpart + ((int64_t)blockIdx.x - NBK) * a.ld);
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        const int64_t t = tid + 256 * u;
                        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
                        for (int k = 0; k < UPD_ROWS; ++k) {
                            if (!(k < a.rows_per_block && row0 + k < m)) continue;
                            acc.x = fma(dv[k], wreg[k][u].x, acc.x);
                            acc.y = fma(dv[k], wreg[k][u].y, acc.y);
                        }
                        if (t < halfw) vp[t] = acc;
                    }
                }
#pragma unroll
                for (int k = 0; k < UPD_ROWS; ++k) {
                    const int64_t i = row0 + k;
                    if (!(k < a.rows_per_block && i < m)) continue;
                    const double f = (i != r) ? -(dv[k] / d_r) : 0.0;
                    double2 *drow = reinterpret_cast<double2 *>(dst + i * a.ld);
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        const int64_t t = tid + 256 * u;
                        if (t >= halfw) continue;
                        double2 o;
                        if (i == r) {
                            o = make_double2(pr[u].x / alpha_r, pr[u].y / alpha_r);
                        } else {
                            o.x = fma(f, pr[u].x, wreg[k][u].x);
                            o.y = fma(f, pr[u].y, wreg[k][u].y);
                        }
                        store_row2<true>(drow + t, o);
                    }
                }
            } else {
                eta_update_rows<true>(src, dst, m, a.ld, r, dv, d_r, alpha_r, row0,
                                a.rows_per_block < UPD_ROWS ? a.rows_per_block : UPD_ROWS, tid);
            }
            if (a.rows_per_block > UPD_ROWS)
                eta_update_rows<true>(src, dst, m, a.ld, r, dv2, d_r, alpha_r, row0 + UPD_ROWS, a.rows_per_block - UPD_ROWS, tid);
        }
    }
    STAMP(2, 3);
    if (row_block) return;

    // ---------------- bookkeeping block(s)
    const int64_t jq = st->s_jq;
    if (blockIdx.x == 0 && tid == 0) {
        ELLP_CHECK(st, r >= -1 && r < m, 9103);
        ELLP_CHECK(st, q >= 0 && q < a.nN, 9104);
        ELLP_CHECK(st, jq >= 0, 9105);
    }
    // A pivot that is tiny next to the rest of its column makes the basis ill-conditioned for a
    // while and leaves O(cond * eps) error in B^-1 that the eta updates then carry along, where
    // the reference's per-iteration LU would forget it at once.  On small LPs (a refresh is a few
    // tens of us) stop the batch after such a pivot; the host refreshes B^-1 now and once more
    // after the following iteration.
    bool tiny_pivot = false;
    if (a.ill_tol > 0.0 && r >= 0) {
        double amax = 0.0;
        for (int64_t i = tid; i < m; i += 256) amax = fmax(amax, fabs(a.d[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o));
        __shared__ double s_amax[4];
        if (lane == 0) s_amax[wave] = amax;
        __syncthreads();
        amax = fmax(fmax(s_amax[0], s_amax[1]), fmax(s_amax[2], s_amax[3]));
        tiny_pivot = fabs(d_r) < a.ill_tol * amax;
    }
    if (MODE == 0) {
        if (blockIdx.x == 1) {
            // block 1: u += (r_q / alpha_r) * rho, the column swap and everything indexed by the nonbasic
            // position q.  The leaving variable comes from bidx[] (B_index as k_ftran2 saw it): block 0
            // overwrites B_index[r] meanwhile.
            if (r < 0) return;  // bound flip: block 0 does it
            if (a.update_u) {
                const double cf = st->s_rq / alpha_r;
                const double2 *rho2 = reinterpret_cast<const double2 *>(src + r * a.ld);
                double2 *u2 = reinterpret_cast<double2 *>(st->usel ? a.u_alt : a.u);
                double2 *sr = reinterpret_cast<double2 *>(a.se_rho);
                for (int64_t t = tid; t < (a.ld >> 1); t += 256) {
                    const double2 p = rho2[t];
                    double2 w = u2[t];
                    w.x = fma(cf, p.x, w.x);
                    w.y = fma(cf, p.y, w.y);
                    u2[t] = w;
                    if (sr) sr[t] = p;
                }
            } else if (a.se_rho) {
                const double2 *rho2 = reinterpret_cast<const double2 *>(src + r * a.ld);
                double2 *sr = reinterpret_cast<double2 *>(a.se_rho);
                for (int64_t t = tid; t < (a.ld >> 1); t += 256) sr[t] = rho2[t];
            }
            if (a.aq_cur) {
                // column-sharded: the owner of position q takes the leaving column (A_B is replicated), every
                // rank puts the entering column into A_B[:, r]
                const bool mine = q >= a.own0 && q < a.own1;
                double2 *cn = reinterpret_cast<double2 *>(a.A_N + (mine ? q : a.own0) * a.ld);
                double2 *cb = reinterpret_cast<double2 *>(a.A_B + r * a.ld);
                const double2 *aq = reinterpret_cast<const double2 *>(a.aq_cur);
                for (int64_t t = tid; t < (a.ld >> 1); t += 256) {
                    const double2 lv = cb[t];
                    if (mine) cn[t] = lv;
                    cb[t] = aq[t];
                }
            } else {
                swap_columns(a.A_N, a.A_B, q, r, a.ld, tid);
            }
            if (tid == 0) {
                const double tc = a.c_N[q];
                a.c_N[q] = a.c_B[r];
                a.c_B[r] = tc;
                a.N_index[q] = a.bidx[r];
                a.Nb[q] = (uint8_t)side;
            }
            return;
        }
        // block 0: the point, the basic side of the swap, the counters
        double se_norm2 = 0.0;
        if (a.se_gamma && r >= 0) {  // steepest edge: the entering variable's weight taken EXACTLY, 1 + |alpha_q|^2
            __shared__ double s_n2[4];
            double acc = 0.0;
            for (int64_t i = tid; i < m; i += 256) acc = fma(a.d[i], a.d[i], acc);
            acc = wave_sum(acc);
            if (lane == 0) s_n2[wave] = acc;
            __syncthreads();
            se_norm2 = ((s_n2[0] + s_n2[1]) + s_n2[2]) + s_n2[3];
        }
        if (lambda > 0.0) {  // primal…:408-417
            for (int64_t i0 = tid; i0 < m; i0 += 4 * 256) {
                int64_t bi[4];
                double xv[4], dv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t i = i0 + 256 * k;
                    bi[k] = i < m ? a.B_index[i] : -1;
                    dv[k] = i < m ? a.d[i] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[k] = bi[k] >= 0 ? a.x[bi[k]] : 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (bi[k] >= 0) a.x[bi[k]] = xv[k] + lambda * dv[k];
            }
            if (tid == 0) {
                if (at_lower) a.x[jq] = a.x[jq] + lambda;
                else a.x[jq] = a.x[jq] - lambda;
            }
        }
        __syncthreads();  // B_index reads above vs the write below
        if (r >= 0) {
            if (tid == 0) {
                a.B_index[r] = jq;
                st->lambda = lambda;
                st->cur = cur ^ 1;
                st->pivots += 1;
                st->open = 0;
                if (a.count_iter) st->iters += 1;
                if (lambda > 0.0) st->obj = st->obj + (at_lower ? lambda * st->s_rq : -(lambda * st->s_rq));
                trace_put(a.trace, st->iters, st->obj);
                if (tiny_pivot) st->tiny = 1;
                if (a.se_gamma) {
                    st->se_valid = 1;
                    st->se_neg = at_lower;
                    st->se_q = q;
                    st->se_arq = alpha_r;
                    st->se_gq = 1.0 + se_norm2;
                }
            }
        } else if (tid == 0) {  // primal…:223-231
            const int nbq = a.Nb[q];
            st->lambda = lambda;
            if (a.se_gamma) st->se_valid = 0;  // a bound flip leaves the weights alone
            if (nbq == ELLP_NB_LOWER) a.Nb[q] = ELLP_NB_UPPER;
            else if (nbq == ELLP_NB_UPPER) a.Nb[q] = ELLP_NB_LOWER;
            else {
                st->panic_code = 229;  // "pivot should have been unbounded"
                st->status = ELLP_ERR_PANIC;
            }
            st->flips += 1;
            st->open = 0;
            if (a.count_iter) st->iters += 1;
            if (lambda > 0.0) st->obj = st->obj + (at_lower ? lambda * st->s_rq : -(lambda * st->s_rq));
            trace_put(a.trace, st->iters, st->obj);
        }
    } else {
        // dual…:296-316
        const double theta_d = st->s_theta_d, delta = st->s_delta;
        const double theta_p = delta / d_r;
        const double *rho = src + r * a.ld;
        if (blockIdx.x == 0) {  // block 0: d[N_j] -= theta_d * alpha_j for every nonbasic j != q
        for (int64_t j0 = tid; j0 < a.nN; j0 += 4 * 256) {
            int64_t v[4];
            double al[4], dv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t j = j0 + 256 * k;
                const bool ok = j < a.nN && j != q;
                v[k] = ok ? a.N_index[j] : -1;
                al[k] = ok ? a.xc.r(j) : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) dv[k] = v[k] >= 0 ? a.dd[v[k]] : 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] >= 0) a.dd[v[k]] = dv[k] - theta_d * al[k];
        }
        return;
        }
        if (blockIdx.x == 1) {
            // block 1: y, the column swap and everything indexed by the nonbasic position q.  The leaving
            // variable comes from the snapshot: block 2 overwrites B_index[r] meanwhile.
            const int64_t lv = st->s_lv;
            for (int64_t i = tid; i < m; i += 256) a.y[i] = a.y[i] + theta_d * rho[i];
            swap_columns(a.A_N, a.A_B, q, r, a.ld, tid);
            if (tid == 0) {
                a.dd[lv] = -theta_d;
                a.dd[jq] = 0.0;
                a.N_index[q] = lv;  // dual…:322-333
                a.Nb[q] = (uint8_t)st->s_side;
                const double tc = a.c_N[q];
                a.c_N[q] = a.c_B[r];
                a.c_B[r] = tc;
            }
            return;
        }
        // block 2: x, the basic side of the swap, the scalars and the next leaving row
        for (int64_t i = tid; i < m; i += 256) {
            const int64_t bi = a.B_index[i];
            a.x[bi] = a.x[bi] - theta_p * a.d[i];
        }
        __syncthreads();
        if (tid == 0) {
            a.x[jq] = a.x[jq] + theta_p;
            st->obj = st->obj + theta_d * delta;
            a.B_index[r] = jq;
            st->cur = cur ^ 1;
            st->pivots += 1;
            st->iters += 1;
            trace_put(a.trace, st->iters, st->obj);
            if (d_r != d_r || theta_p != theta_p) st->status = ELLP_ERR_NAN;
        }
        __syncthreads();
        find_leaving(a.B_index, a.x, a.kind, a.lb, a.ub, a.eps, m, tid, s_tmp, st, a.maxviol);
        if (tid == 0 && tiny_pivot && st->status == ST_RUNNING) st->tiny = 1;
    }
    STAMP(2, 4);
}

// dual: leaving row before the first iteration of a run() slice
struct DLeaveArgs {
    const double *x, *lb, *ub;
    const uint8_t *kind;
    const int64_t *B_index;
    DevState *st;
    int64_t m;
    double eps;
    int maxviol;
};
__global__ __launch_bounds__(256) void k_dleave(DLeaveArgs a) {
    __shared__ long long s_tmp[4];
    if (a.st->status != ST_RUNNING) return;
    find_leaving(a.B_index, a.x, a.kind, a.lb, a.ub, a.eps, a.m, threadIdx.x, s_tmp, a.st, a.maxviol);
}

// ------------------------------------------------------------------ BTRAN  u = B^-T c_B  (primal…:184-187)
// u_j = sum_i c_B[i] * W[i][j].  grid (col tiles of 512, row tiles): partial sums per row tile
// then a fixed-order reduction — deterministic, no atomics.  Rows with c_B[i] == 0 are skipped
// (phase 1: only artificial basics carry cost).  Runs every `btran_refresh` iterations; in
// between u is carried by the O(m) update in k_update2.
struct BtranArgs {
    const double *W0, *W1, *c_B;
    double *upart, *u, *u_alt;
    DevState *st;
    int64_t m, ld;
    int rows_per_tile, ntiles;
};

__global__ __launch_bounds__(256) void k_btran_part(BtranArgs a) {
    if (a.st->status != ST_RUNNING) return;
    const int64_t half = a.ld >> 1;
    const int64_t j2 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * a.rows_per_tile;
    const int64_t i1 = (i0 + a.rows_per_tile < a.m) ? i0 + a.rows_per_tile : a.m;
    double2 acc = make_double2(0.0, 0.0);
    if (j2 < half) {
        const double2 *W2 = reinterpret_cast<const double2 *>(a.st->cur ? a.W1 : a.W0);
        for (int64_t i = i0; i < i1; ++i) {
            const double ci = a.c_B[i];
            if (ci != 0.0) {
                const double2 w = W2[i * half + j2];
                acc.x = fma(ci, w.x, acc.x);
                acc.y = fma(ci, w.y, acc.y);
            }
        }
        reinterpret_cast<double2 *>(a.upart)[(int64_t)blockIdx.y * half + j2] = acc;
    }
}
__global__ __launch_bounds__(256) void k_btran_reduce(BtranArgs a) {
    if (a.st->status != ST_RUNNING) return;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= a.ld) return;
    double s = 0.0;
    int t = 0;
    for (; t + 8 <= a.ntiles; t += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a.upart[(int64_t)(t + k) * a.ld + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; t < a.ntiles; ++t) s += a.upart[(int64_t)t * a.ld + j];
    (a.st->usel ? a.u_alt : a.u)[j] = s;
}

// ------------------------------------------------------------------ drift monitor of B^-1
// The reference factorises A_B afresh every iteration; here B^-1 carries the rounding of every eta
// update since its last refresh, and how fast that grows depends on the LP (a tall 500 x 100 LP of
// the benign synthetic family reached 1e-9 in x after 160 updates, config 3 stays at 1e-12 after
// 1000).  So every few iterations the FTRAN result is checked against the basis itself:
// t = A_B * alpha - a_q must vanish (alpha = B^-1 a_q = +-d).  One GEMV over A_B (8*m*ld bytes),
// same shape as k_btran_part: tiles of basis columns, coalesced along the rows, partials in
// `upart`; k_drift_reduce folds them in a fixed order (deterministic: replicated engines of a
// sharded run must take the same decision) and raises DevState::tiny — the maintenance request of
// a tiny pivot — when max|t| exceeds drift_tol * max|a_q|.
struct DriftArgs {
    const double *A_B, *A_N, *d;
    const double *aq_cur;  // column-sharded engines: the entering column
    double *upart;
    DevState *st;
    int64_t m, ld;
    int cols_per_tile, ntiles;
    double tol;
};
__global__ __launch_bounds__(256) void k_drift_part(DriftArgs a) {
    if (a.st->status != ST_RUNNING || a.st->pp_skip) return;
    const int64_t half = a.ld >> 1;
    const int64_t i2 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.y * a.cols_per_tile;
    const int64_t k1 = (k0 + a.cols_per_tile < a.m) ? k0 + a.cols_per_tile : a.m;
    if (i2 >= half) return;
    const double2 *AB2 = reinterpret_cast<const double2 *>(a.A_B);
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t k = k0; k < k1; ++k) {
        const double dk = a.d[k];
        const double2 c = AB2[k * half + i2];
        acc.x = fma(dk, c.x, acc.x);
        acc.y = fma(dk, c.y, acc.y);
    }
    reinterpret_cast<double2 *>(a.upart)[(int64_t)blockIdx.y * half + i2] = acc;
}
__global__ __launch_bounds__(1024) void k_drift_reduce(DriftArgs a) {
    __shared__ double s_r[16], s_s[16];
    DevState *st = a.st;
    if (st->status != ST_RUNNING || st->pp_skip) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double sgn = st->s_at_lower ? -1.0 : 1.0;  // alpha = sgn * d (k_ftran2 stores d = +-alpha)
    const double *aq = a.aq_cur ? a.aq_cur : a.A_N + st->s_q * a.ld;
    double res = 0.0, scale = 0.0;
    for (int64_t i = tid; i < a.m; i += 1024) {
        double t = 0.0;
        for (int k = 0; k < a.ntiles; ++k) t += a.upart[(int64_t)k * a.ld + i];
        const double q = aq[i];
        res = fmax(res, fabs(sgn * t - q));
        scale = fmax(scale, fabs(q));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        res = fmax(res, __shfl_xor(res, o));
        scale = fmax(scale, __shfl_xor(scale, o));
    }
    if (lane == 0) {
        s_r[wave] = res;
        s_s[wave] = scale;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            res = fmax(res, s_r[w]);
            scale = fmax(scale, s_s[w]);
        }
        const double rel = res / (scale > 0.0 ? scale : 1.0);
        st->drift = rel;
        if (rel > a.tol || rel != rel) st->tiny = 1;
    }
}

// ------------------------------------------------------------------ resynchronising x_B with the basis
// The reference updates x incrementally (x_B += lambda d, primal…:408-417; x_B -= theta_p alpha_q,
// dual…:308-313) from a d / alpha_q of LU quality at every iteration.  Here d comes from a B^-1
// that is only as good as its last refresh, and what a few ill-conditioned bases add to x stays
// there: a netlib ADLITTLE in another variable order ended its dual phase 1 "Infeasible" because a
// basic variable sat at -8e-10 — 2.6e-9 away from B^-1 (b - N x_N) on a basis of condition 1e3 —
// and was picked as leaving row with no eligible column.  So whenever B^-1 has just been refreshed,
// x_B is recomputed from it:  t = b - A_N x_N  (k_resync_part over A_N + k_resync_rhs),
// cand_i = B^-1[i,:] . t  (k_resync_xb) — and adopted only if it differs from the carried x_B by more
// than 1e-11 (1 + max|x_B|) somewhere (k_resync_apply): on benign LPs the carried x is as good as
// the reference's and keeps the reference's rounding path (a tall synthetic LP whose phase 1 ends on
// EPS-ties stays pivot-for-pivot), on LPs that went through ill-conditioned bases the error is
// removed before it can flip an EPS decision.  The nonbasic values are left exactly as they are.
struct ResyncArgs {
    const double *A_N, *W0, *W1, *b;
    double *x, *xg, *tvec, *upart, *cand;
    unsigned long long *maxbits;  // [0] max|cand - x_B|, [1] max|x_B| as bit patterns (non-negative doubles)
    const int64_t *B_index, *N_index;
    DevState *st;
    int64_t m, ld, nN;
    int cols_per_tile, ntiles;
    int force;  // 1: adopt the recomputed x_B whatever the difference (phase hand-off of the dual)
};
__global__ __launch_bounds__(256) void k_resync_gather(ResyncArgs a) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < a.nN) a.xg[j] = a.x[a.N_index[j]];
}
// partial[tile][i] = sum over the tile's columns j of A_N[i,j] * xg[j]   (coalesced along i)
__global__ __launch_bounds__(256) void k_resync_part(ResyncArgs a) {
    if (a.st->status != ST_RUNNING && a.st->status != ST_NEED_MAINT) return;
    const int64_t half = a.ld >> 1;
    const int64_t i2 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.y * a.cols_per_tile;
    const int64_t j1 = (j0 + a.cols_per_tile < a.nN) ? j0 + a.cols_per_tile : a.nN;
    if (i2 >= half) return;
    const double2 *AN2 = reinterpret_cast<const double2 *>(a.A_N);
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t j = j0; j < j1; ++j) {
        const double xj = a.xg[j];
        if (xj == 0.0) continue;
        const double2 c = AN2[j * half + i2];
        acc.x = fma(xj, c.x, acc.x);
        acc.y = fma(xj, c.y, acc.y);
    }
    reinterpret_cast<double2 *>(a.upart)[(int64_t)blockIdx.y * half + i2] = acc;
}
__global__ __launch_bounds__(256) void k_resync_rhs(ResyncArgs a) {
    if (a.st->status != ST_RUNNING && a.st->status != ST_NEED_MAINT) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.ld) return;
    double s = 0.0;
    for (int k = 0; k < a.ntiles; ++k) s += a.upart[(int64_t)k * a.ld + i];
    a.tvec[i] = i < a.m ? a.b[i] - s : 0.0;
}
// one wave per basic row: cand_i = B^-1[i,:] . t, and the two maxima the decision needs (a maximum is
// exact in any order: atomicMax on the bit pattern of a non-negative double)
__global__ __launch_bounds__(256) void k_resync_xb(ResyncArgs a) {
    if (a.st->status != ST_RUNNING && a.st->status != ST_NEED_MAINT) return;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.m) return;
    const int64_t half = a.ld >> 1;
    const double2 *row = reinterpret_cast<const double2 *>((a.st->cur ? a.W1 : a.W0) + i * a.ld);
    const double2 *t2 = reinterpret_cast<const double2 *>(a.tvec);
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t t = lane; t < half; t += WAVE) {
        const double2 w = row[t], v = t2[t];
        acc0 = fma(w.x, v.x, acc0);
        acc1 = fma(w.y, v.y, acc1);
    }
    const double r = wave_sum(acc0 + acc1);
    if (lane == 0) {
        const double xo = a.x[a.B_index[i]];
        a.cand[i] = r;
        const double df = fabs(r - xo), ax = fabs(xo);
        if (df == df) atomicMax(&a.maxbits[0], (unsigned long long)__double_as_longlong(df));
        else atomicMax(&a.maxbits[0], 0x7ff0000000000000ull);  // NaN: adopt nothing sensible anyway
        if (ax == ax) atomicMax(&a.maxbits[1], (unsigned long long)__double_as_longlong(ax));
    }
}
__global__ __launch_bounds__(256) void k_resync_apply(ResyncArgs a) {
    if (a.st->status != ST_RUNNING && a.st->status != ST_NEED_MAINT) return;
    if (a.st->need_rebuild) return;  // B^-1 failed its refresh: it is rebuilt first, then x_B is checked again
    const double maxdiff = __longlong_as_double((long long)a.maxbits[0]);
    const double maxx = __longlong_as_double((long long)a.maxbits[1]);
    if (!a.force && (!(maxdiff > 1e-11 * (1.0 + maxx)) || isinf(maxdiff))) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.m) a.x[a.B_index[i]] = a.cand[i];
}

// ------------------------------------------------------------------ refactorisation of B^-1 from A_B
// Product-form rebuild with partial pivoting: start from W = I and bring the m basic columns
// in one at a time: alpha = W a_k, pivot row p = first max |alpha_i| over the rows not used yet
// (the same pivot partial-pivot LU takes, primal…:173), eta update with row p.  The pivot
// magnitudes are LU's U_kk, so the reference's singularity guard (any |U_ii| < EPS,
// primal…:175-179) is checked on them.  Afterwards rows are permuted so that row k belongs to
// basic position k.
struct RefArgs {
    double *W0, *W1, *d;
    const double *A_B;
    int32_t *used;
    int64_t *perm;
    DevState *st;
    int64_t m, ld;
    int rows_per_block;
    double eps;
};

__global__ __launch_bounds__(256) void k_ref_init(RefArgs a) {
    if (a.st->status != ST_RUNNING) return;
    double *W = a.st->cur ? a.W1 : a.W0;
    const int64_t total = a.m * a.ld;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / a.ld, j = t - i * a.ld;
        W[t] = (i == j) ? 1.0 : 0.0;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.m; i += (int64_t)gridDim.x * 256) a.used[i] = 0;
}
__global__ void k_ref_begin(RefArgs a) {
    if (a.st->status != ST_RUNNING) return;
    a.st->refk = 0;
}

__global__ __launch_bounds__(256) void k_ref_ftran(RefArgs a) {
    if (a.st->status != ST_RUNNING) return;
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t half = a.ld >> 1;
    const double *W = a.st->cur ? a.W1 : a.W0;
    const double2 *col = reinterpret_cast<const double2 *>(a.A_B + a.st->refk * a.ld);
    for (int64_t i = wave_global; i < a.m; i += nwaves) {
        const double2 *row = reinterpret_cast<const double2 *>(W + i * a.ld);
        double acc0 = 0.0, acc1 = 0.0;
        for (int64_t t0 = lane; t0 < half; t0 += 8 * WAVE) {
            double2 w[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t t = t0 + u * WAVE;
                w[u] = t < half ? row[t] : make_double2(0.0, 0.0);
                c[u] = t < half ? col[t] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = fma(w[u].x, c[u].x, acc0);
                acc1 = fma(w[u].y, c[u].y, acc1);
            }
        }
        const double s = wave_sum(acc0 + acc1);
        if (lane == 0) a.d[i] = s;
    }
}

__global__ __launch_bounds__(1024) void k_ref_pick(RefArgs a) {
    __shared__ double s_v[16];
    __shared__ long long s_i[16];
    DevState *st = a.st;
    if (st->status != ST_RUNNING) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bv = -1.0;
    long long bi = -1;
    for (int64_t i = tid; i < a.m; i += 1024) {
        if (a.used[i]) continue;
        const double v = fabs(a.d[i]);
        if (v > bv) {  // strict: first maximum within this thread's increasing i
            bv = v;
            bi = i;
        }
    }
    // lexicographic (max value, min index) — associative, so a tree is exact
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o);
        const long long oi = __shfl_xor(bi, o);
        if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            if (s_i[w] >= 0 && (bi < 0 || s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi))) {
                bv = s_v[w];
                bi = s_i[w];
            }
        }
        const long long p = bi;
        // primal: any |U_ii| < EPS is Err("A_B is not invertible") (primal…:175-179); the dual loop
        // has no such guard — only an exactly zero pivot makes its lu.solve() fail (dual…:294)
        if (p < 0 || bv != bv || bv == 0.0 || bv < a.eps) {
            st->status = ELLP_ERR_SINGULAR;
        } else {
            a.used[p] = 1;
            a.perm[st->refk] = p;
            st->r = p;
            st->d_r = a.d[p];
            st->alpha_r = a.d[p];
            st->cur ^= 1;  // k_ref_update reads buffer cur^1, writes buffer cur
            st->do_update = 1;
            st->refk += 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_ref_update(RefArgs a) {
    const DevState *st = a.st;
    if (st->status != ST_RUNNING || !st->do_update) return;
    const double *src = st->cur ? a.W0 : a.W1;
    double *dst = st->cur ? a.W1 : a.W0;
    const int64_t row0 = (int64_t)blockIdx.x * a.rows_per_block;
    double dv[UPD_ROWS];
#pragma unroll
    for (int k = 0; k < UPD_ROWS; ++k) dv[k] = (k < a.rows_per_block && row0 + k < a.m) ? a.d[row0 + k] : 0.0;
    eta_update_rows<false>(src, dst, a.m, a.ld, st->r, dv, st->d_r, st->alpha_r, row0, a.rows_per_block, threadIdx.x);
}

__global__ __launch_bounds__(256) void k_ref_permute(RefArgs a) {
    // reads buffer cur, writes buffer cur^1; k_ref_finish then flips cur
    if (a.st->status != ST_RUNNING) return;
    const double *src = a.st->cur ? a.W1 : a.W0;
    double *dstb = a.st->cur ? a.W0 : a.W1;
    const int64_t k = blockIdx.x;
    const int64_t p = a.perm[k];
    const double2 *s = reinterpret_cast<const double2 *>(src + p * a.ld);
    double2 *dst = reinterpret_cast<double2 *>(dstb + k * a.ld);
    for (int64_t t = threadIdx.x; t < (a.ld >> 1); t += 256) dst[t] = s[t];
}
__global__ void k_ref_finish(RefArgs a) {
    if (a.st->status != ST_RUNNING) return;
    a.st->cur ^= 1;
    a.st->do_update = 0;
}

// ------------------------------------------------------------------ Newton-Schulz refresh of B^-1
// Between rebuilds the explicit inverse picks up rounding drift from the eta updates.  Instead of
// re-deriving it column by column (m dependent steps), one Newton-Schulz step
//     E = I - A_B W ;  W <- W + W E
// squares the residual (|E| ~ 1e-12 -> rounding level) with two m x m x m f64 GEMMs and no
// sequential dependency.  k_gemm128: 128 x 128 output tile per 256-thread block, 8 x 8
// accumulators per thread, K-step 16 staged through LDS.
//   AMODE 0: A(i,k) = A[k*ld + i] (column-major A_B)      OMODE 0: out = delta_ij - acc  (E)
//   AMODE 1: A(i,k) = A[i*ld + k] (row-major W)            OMODE 1: out = Cin + acc       (W + W E)
//   B(k,j) = B[k*ld + j] (row-major) in both uses.
// WSEL picks operands from the two B^-1 buffers by st->cur on the device.
struct GemmArgs {
    double *W0, *W1;
    const double *A_B;
    double *T;         // third m x ld buffer
    double *tilemax;   // per-tile max |out| (OMODE 0: the residual)
    const DevState *st;
    int64_t m, ld;
};

template <int STEP>  // STEP 0: E = I - A_B W[cur] -> W[cur^1] ; STEP 1: T = W[cur] + W[cur] E
__global__ __launch_bounds__(256) void k_gemm128(GemmArgs a) {
    constexpr int BM = 128, BN = 128, BK = 16;
    __shared__ double sA[BK][BM];
    __shared__ double sB[BK][BN];
    __shared__ double s_red[4];
    if (a.st->status != ST_RUNNING) return;
    if (STEP == 1 && a.st->need_rebuild) return;  // k_resid_reduce: not a small perturbation any more
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t m = a.m, ld = a.ld;
    const double *Wc = a.st->cur ? a.W1 : a.W0;
    double *Wo = a.st->cur ? a.W0 : a.W1;
    const double *A = STEP == 0 ? a.A_B : Wc;
    const double *B = STEP == 0 ? Wc : Wo;
    double *C = STEP == 0 ? Wo : a.T;
    const int64_t i0 = (int64_t)blockIdx.y * BM, j0 = (int64_t)blockIdx.x * BN;
    double acc[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = 0.0;
    for (int64_t k0 = 0; k0 < m; k0 += BK) {
        // ---- stage A tile (BM x BK) and B tile (BK x BN)
        if (STEP == 0) {  // column-major A: 128 contiguous i per k
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int kk = (tid >> 5) + 8 * p, i4 = (tid & 31) * 4;
                const int64_t k = k0 + kk;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t i = i0 + i4 + c;
                    sA[kk][i4 + c] = (k < m && i < m) ? A[k * ld + i] : 0.0;
                }
            }
        } else {  // row-major A: 16 contiguous k per i
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int il = (tid >> 2) + 64 * p, k4 = (tid & 3) * 4;
                const int64_t i = i0 + il;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t k = k0 + k4 + c;
                    sA[k4 + c][il] = (k < m && i < m) ? A[i * ld + k] : 0.0;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int kk = (tid >> 5) + 8 * p, j4 = (tid & 31) * 4;
            const int64_t k = k0 + kk;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t j = j0 + j4 + c;
                sB[kk][j4 + c] = (k < m && j < m) ? B[k * ld + j] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            double av[8], bv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) av[r] = sA[kk][ty * 8 + r];
#pragma unroll
            for (int c = 0; c < 8; ++c) bv[c] = sB[kk][tx * 8 + c];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
        }
        __syncthreads();
    }
    double worst = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int64_t i = i0 + ty * 8 + r;
        if (i >= m) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int64_t j = j0 + tx * 8 + c;
            if (j >= m) continue;
            double o;
            if (STEP == 0) o = (i == j ? 1.0 : 0.0) - acc[r][c];
            else o = Wc[i * ld + j] + acc[r][c];
            C[i * ld + j] = o;
            worst = fmax(worst, fabs(o));
        }
    }
    if (STEP == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o));
        if ((tid & 63) == 0) s_red[tid >> 6] = worst;
        __syncthreads();
        if (tid == 0)
            a.tilemax[blockIdx.y * gridDim.x + blockIdx.x] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    }
}

// T -> W[cur^1], then cur ^= 1 (k_ref_finish)
__global__ __launch_bounds__(256) void k_copy_to_other(GemmArgs a) {
    if (a.st->status != ST_RUNNING || a.st->need_rebuild) return;
    double2 *dst = reinterpret_cast<double2 *>(a.st->cur ? a.W0 : a.W1);
    const double2 *src = reinterpret_cast<const double2 *>(a.T);
    const int64_t total = a.m * (a.ld >> 1);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) dst[t] = src[t];
}

// residual of the refresh's first GEMM, folded on the device: max over the tile maxima of |I - A_B W|.
// A Newton-Schulz step only converges from a small residual; at 1e-4 and beyond the step is skipped
// (k_gemm128<1>, k_copy_to_other, k_refresh_finish test need_rebuild) and the loop is stopped with the
// same request a tiny pivot raises, so that the host rebuilds B^-1 from A_B when it services it.
// Deciding here instead of on the host keeps the stream from draining once per refresh.
__global__ __launch_bounds__(256) void k_resid_reduce(GemmArgs a, int ntiles) {
    __shared__ double s_r[4];
    DevState *st = const_cast<DevState *>(a.st);
    if (st->status != ST_RUNNING) return;
    double w = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const double v = a.tilemax[t];
        w = (v > w || v != v) ? v : w;
    }
    // NaN-propagating maximum: once w is NaN every comparison below is false and it stays
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(w, o);
        w = (w != w) ? w : ((v > w || v != v) ? v : w);
    }
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            const double v = s_r[k];
            w = (w != w) ? w : ((v > w || v != v) ? v : w);
        }
        st->resid = w;
        if (!(w < 1e-4)) {
            st->need_rebuild = 1;
            st->tiny = 1;
        }
    }
}
__global__ void k_refresh_finish(GemmArgs a) {
    DevState *st = const_cast<DevState *>(a.st);
    if (st->status != ST_RUNNING || st->need_rebuild) return;
    st->cur ^= 1;
}

__global__ __launch_bounds__(256) void k_scale_inverse(double *W0, double *W1, const DevState *st, int64_t total,
                                                       double factor) {
    double *W = st->cur ? W1 : W0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) W[t] *= factor;
}

// max |W A_B - I| (drift monitor, test/diagnostic only — m^3 work)
__global__ __launch_bounds__(256) void k_inv_residual(const double *W0, const double *W1, const DevState *st,
                                                      const double *A_B, int64_t m, int64_t ld, double *out) {
    __shared__ double s_m[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x;
    const double *W = st->cur ? W1 : W0;
    const double2 *row = reinterpret_cast<const double2 *>(W + i * ld);
    double worst = 0.0;
    for (int64_t k = wave; k < m; k += 4) {
        const double2 *col = reinterpret_cast<const double2 *>(A_B + k * ld);
        double acc = 0.0;
        for (int64_t t = lane; t < (ld >> 1); t += WAVE) {
            const double2 w = row[t], c = col[t];
            acc = fma(w.x, c.x, acc);
            acc = fma(w.y, c.y, acc);
        }
        acc = wave_sum(acc);
        const double e = fabs(acc - (i == k ? 1.0 : 0.0));
        worst = fmax(worst, e);
    }
    if (lane == 0) s_m[wave] = worst;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]));
}

// gather columns of A (m x n, ld m) into a padded destination (ld) by an index list
__global__ __launch_bounds__(256) void k_gather_cols(const double *A, int64_t m, const int64_t *index, double *dst,
                                                     int64_t ld) {
    const int64_t k = blockIdx.x;
    const double *src = A + index[k] * m;
    double *o = dst + k * ld;
    for (int64_t i = threadIdx.x; i < ld; i += 256) o[i] = i < m ? src[i] : 0.0;
}
__global__ __launch_bounds__(256) void k_gather_vec(const double *v, const int64_t *index, double *dst, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) dst[k] = v[index[k]];
}

// c . x for a primal engine (standard_form.rs:48), from the gathered costs: sum_i c_B[i] x[B_i] +
// sum_j c_N[j] x[N_j]; one block, fixed reduction order.  Only for ellp_stats.obj (the loop never
// needs it: the reference computes it outside the loop too, primal…:42-45).
__global__ __launch_bounds__(1024) void k_primal_obj(const double *c_B, const double *c_N, const double *x,
                                                    const int64_t *B_index, const int64_t *N_index, int64_t m,
                                                    int64_t nN, DevState *st) {
    __shared__ double s_p[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) acc = fma(c_B[i], x[B_index[i]], acc);
    for (int64_t j = threadIdx.x; j < nN; j += 1024) {
        const double cj = c_N[j];
        if (cj != 0.0) acc = fma(cj, x[N_index[j]], acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_p[w];
        st->obj = t;
    }
}

// ---- DualPhase2::from(phase_1) on the device (dual_problem.rs:258-404), see ellp_engine_dual_rephase
// nonbasic columns into variable-index order: dst[:, p] = src[:, perm[p]]
__global__ __launch_bounds__(256) void k_permute_cols(const double *src, double *dst, const int64_t *perm, int64_t ld) {
    const double2 *s = reinterpret_cast<const double2 *>(src + perm[blockIdx.x] * ld);
    double2 *d = reinterpret_cast<double2 *>(dst + (int64_t)blockIdx.x * ld);
    for (int64_t t = threadIdx.x; t < (ld >> 1); t += 256) d[t] = s[t];
}
// ---- certified hybrid, "certify or redo" (DESIGN.md §3.1c): the invariants the reference's loops maintain, measured on the
// point a solve ended on.  out[0] = SUM of the bound violations of x over all variables (every iterate of the primal loop is
// feasible; the caller's phase-1 test, primal…:42-50, is on a sum: the objective over the artificials); out[1] = SUM of the
// violations of the dual loop's entry assertion on d (dual_simplex_solver.rs:139-151, which holds at every iteration of the
// reference's loop: Lower: d >= 0, Upper: d <= 0, Free: d = 0; the caller's phase-1 test, dual…:45-50, is again a sum);
// out[2] = the dual objective recomputed from (y, d) (standard_form.rs:52-68), for diagnostics.
struct InvArgs {
    const double *x, *lb, *ub, *b, *y, *dd;
    const uint8_t *kind, *Nb;
    const int64_t *N_index;
    int64_t m, n_c, nN;
    int dual;
    double *out;
};
__global__ __launch_bounds__(1024) void k_invariants(InvArgs a) {
    __shared__ double s_a[16], s_b[16], s_c[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double pv = 0.0, dv = 0.0, ob = 0.0;
    for (int64_t i = tid; i < a.n_c; i += 1024) {
        const int k = a.kind[i];
        const double xi = a.x[i], l = a.lb[i], u = a.ub[i];
        double v = 0.0;
        if (k == ELLP_BOUND_LOWER) v = l - xi;
        else if (k == ELLP_BOUND_UPPER) v = xi - u;
        else if (k == ELLP_BOUND_TWOSIDED) v = fmax(l - xi, xi - u);
        else if (k == ELLP_BOUND_FIXED) v = fabs(xi - l);
        if (v != v) v = INFINITY;
        if (v > 0.0) pv += v;
        if (a.dual) {
            const double di = a.dd[i];
            if (k == ELLP_BOUND_LOWER) ob += l * di;
            else if (k == ELLP_BOUND_UPPER) ob += u * di;
            else if (k == ELLP_BOUND_TWOSIDED) ob += (di > 0.0) ? l * di : u * di;
            else if (k == ELLP_BOUND_FIXED) ob += l * di;
        }
    }
    if (a.dual) {
        for (int64_t i = tid; i < a.m; i += 1024) ob += a.b[i] * a.y[i];
        for (int64_t j = tid; j < a.nN; j += 1024) {
            const double di = a.dd[a.N_index[j]];
            const int nb = a.Nb[j];
            double v = nb == ELLP_NB_LOWER ? -di : (nb == ELLP_NB_UPPER ? di : fabs(di));
            if (v != v) v = INFINITY;
            if (v > 0.0) dv += v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pv += __shfl_xor(pv, o);
        dv += __shfl_xor(dv, o);
        ob += __shfl_xor(ob, o);
    }
    if (lane == 0) {
        s_a[wave] = pv;
        s_b[wave] = dv;
        s_c[wave] = ob;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            pv += s_a[w];
            dv += s_b[w];
            ob += s_c[w];
        }
        a.out[0] = pv;
        a.out[1] = dv;
        a.out[2] = ob;
    }
}
// column j of the restored arrangement comes from A_B (src[j] >= 0: position) or A_N (src[j] < 0: position -1 - src[j])
__global__ __launch_bounds__(256) void k_restore_cols(const double *A_B, const double *A_N, const int64_t *src, double *dst, int64_t ld) {
    const int64_t j = blockIdx.x;
    const int64_t sj = src[j];
    const double2 *s = reinterpret_cast<const double2 *>(sj >= 0 ? A_B + sj * ld : A_N + (-1 - sj) * ld);
    double2 *d = reinterpret_cast<double2 *>(dst + j * ld);
    for (int64_t t = threadIdx.x; t < (ld >> 1); t += 256) d[t] = s[t];
}

struct DualRephaseArgs {
    const double *A_N, *A_B, *y, *c;
    const uint8_t *kind;
    const double *lb, *ub;
    const int64_t *N_index, *B_index;
    double *dd, *x;
    uint8_t *Nb;
    DevState *st;
    int64_t m, ld, nN;
    double eps;
    int phase1;  // DualPhase1::new's labelling (dual_problem.rs:177-203) instead of DualPhase2::from's (:286-323)
};
// one wave per variable: d_i = c_i - a_i . y (dual_problem.rs:284 / :173); nonbasic i: value and label by bound
// kind and the sign of d_i, with the reference's assertions (:293-321 / :201)
__global__ __launch_bounds__(256) void k_dual_rephase(DualRephaseArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.nN + a.m) return;
    const bool nonbasic = w < a.nN;
    const int64_t var = nonbasic ? a.N_index[w] : a.B_index[w - a.nN];
    const double2 *col = reinterpret_cast<const double2 *>(nonbasic ? a.A_N + w * a.ld : a.A_B + (w - a.nN) * a.ld);
    const double2 *y2 = reinterpret_cast<const double2 *>(a.y);
    double acc = 0.0;
    for (int64_t t = lane; t < (a.ld >> 1); t += WAVE) {
        const double2 cv = col[t], yv = y2[t];
        acc = fma(cv.x, yv.x, acc);
        acc = fma(cv.y, yv.y, acc);
    }
    acc = wave_sum(acc);
    if (lane != 0) return;
    const double di = a.c[var] - acc;
    a.dd[var] = di;
    if (!nonbasic) return;
    double xi = 0.0;
    int label = ELLP_NB_LOWER, bad = 0;
    if (a.phase1) {
        const int k = a.kind[var];
        label = di >= 0.0 ? ELLP_NB_LOWER : ELLP_NB_UPPER;
        if (k == ELLP_BOUND_TWOSIDED) xi = di >= 0.0 ? a.lb[var] : a.ub[var];
        else if (k == ELLP_BOUND_FIXED) xi = a.lb[var];
        else {
            a.st->panic_code = 201;  // "bounds should always be fixed or two-sided" (dual_problem.rs:201)
            a.st->status = ELLP_ERR_PANIC;
        }
        a.x[var] = xi;
        a.Nb[w] = (uint8_t)label;
        return;
    }
    switch (a.kind[var]) {
    case ELLP_BOUND_FREE: bad = !(fabs(di) < a.eps); xi = 0.0; label = ELLP_NB_FREE; break;
    case ELLP_BOUND_LOWER: bad = !(di > -a.eps); xi = a.lb[var]; label = ELLP_NB_LOWER; break;
    case ELLP_BOUND_UPPER: bad = !(di < a.eps); xi = a.ub[var]; label = ELLP_NB_UPPER; break;
    case ELLP_BOUND_TWOSIDED:
        if (di >= 0.0) { xi = a.lb[var]; label = ELLP_NB_LOWER; }
        else { xi = a.ub[var]; label = ELLP_NB_UPPER; }
        break;
    default: xi = a.lb[var]; label = ELLP_NB_LOWER; break;  // Fixed
    }
    a.x[var] = xi;
    a.Nb[w] = (uint8_t)label;
    if (bad) {
        a.st->panic_code = 293;  // the assert!s of dual_problem.rs:293-305
        a.st->status = ELLP_ERR_PANIC;
    }
}

// primal_problem.rs:236-246: the artificial column of row i is signum(b~_i) e_i (f64::signum: +1 for +0.0,
// -1 for -0.0) and its variable starts at |b~_i|; b~ = b - A v is in `bt`.  Basic position i holds artificial i.
__global__ __launch_bounds__(256) void k_phase1_art(const double *bt, double *A_B, double *x, const int64_t *B_index,
                                                    int64_t m, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const double v = bt[i];
    const double sgn = (v != v) ? v : (signbit(v) ? -1.0 : 1.0);
    A_B[i * ld + i] = sgn;
    x[B_index[i]] = fabs(v);
}

// phase hand-off (ellp_engine_rephase): re-gather the costs by the current index sets and relabel the
// nonbasic variables that became Free
__global__ __launch_bounds__(256) void k_rephase(const double *c, const uint8_t *kind, const int64_t *B_index,
                                                 const int64_t *N_index, double *c_B, double *c_N, uint8_t *Nb,
                                                 int64_t m, int64_t nN) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < m) c_B[k] = c[B_index[k]];
    if (k < nN) {
        const int64_t j = N_index[k];
        c_N[k] = c[j];
        if (kind[j] == ELLP_BOUND_FREE) Nb[k] = ELLP_NB_FREE;
    }
}

#include "ellp_gemm.inc"
#include "ellp_lagged.inc"
#include "ellp_dualfu.inc"
#include "ellp_shard.inc"
#include "ellp_rebuild.inc"
#include "ellp_small.inc"
#include "ellp_mid.inc"
#include "ellp_se.inc"
#include "ellp_exact.inc"

}  // namespace

// ====================================================================== host side

// RCCL entry points, bound at run time so that the library has no link-time dependency on RCCL
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// a stream and the pinned status buffers of an engine; pooled per process (host_pool)
struct HostSet {
    int device = -1;
    hipStream_t stream = nullptr;
    DevState *h_st = nullptr;
    DevState *h_look = nullptr;
};

struct ellp_engine {
    HostSet host_set;
    int kind = 0;
    int64_t m = 0, n = 0, n_c = 0, nN = 0, ld = 0;
    double eps = 1e-10;
    ellp_opts opts{};
    int device = 0;
    hipStream_t stream = nullptr;
    // device memory
    double *A_B = nullptr, *A_N = nullptr, *W = nullptr, *W2 = nullptr;
    double *b_dev = nullptr, *xg = nullptr, *tvec = nullptr, *cand = nullptr;  // resync of x_B (k_resync_*)
    unsigned long long *maxbits = nullptr;
    uint64_t resyncs = 0;
    double *c_B = nullptr, *c_N = nullptr, *u = nullptr, *X = nullptr;
    double *x = nullptr, *lb = nullptr, *ub = nullptr, *d = nullptr;
    double *trace_obj = nullptr;
    unsigned long long *trace_it = nullptr;
    int trace_len = 0;
    int32_t *binfo = nullptr;
    // blocked rebuild (ellp_rebuild.inc)
    double *bl_Cpart = nullptr, *bl_C = nullptr, *bl_V = nullptr, *bl_Vs = nullptr, *bl_Wp = nullptr, *bl_nzval = nullptr;
    int64_t *bl_Pm = nullptr, *bl_nzrow = nullptr;
    int32_t *bl_nzcnt = nullptr;
    int bl_splits = 1;
    uint64_t rebuild_shortcuts = 0;
    double *aq_save = nullptr, *bmin = nullptr;  // two-launch pipeline: parked entering column, row-block minima of lambda
    bool hst_fresh = false;  // h_st is the device state as the previous call (an ellp_engine_run) left it: no read-back needed
    bool obj_fresh = false;  // h_st->obj is c . x of the state in h_st (primal; see ellp_engine_run / fill_stats)
    bool lagged = false;    // two launches per primal iteration (ellp_lagged.inc)
    bool dual_fused = false;  // dual: FTRAN and eta update in one pass over B^-1 (ellp_dualfu.inc)
    bool dual_fold = false;   // ... and its closing work done by the next pricing launch (no tiny-pivot maintenance)
    bool dual_open = false;   // a fused iteration may still be open on the device (launch_dual_close)
    unsigned long long dual_seq = 0;  // number of the dual iteration being enqueued
    bool lag_open = false;  // a k_ftran_eta has been enqueued whose ratio test no kernel has folded yet
    size_t price2_lds = 0;
    double *upart = nullptr, *y = nullptr, *dd = nullptr, *lam = nullptr, *resid = nullptr, *T = nullptr;
    uint64_t refreshes = 0, maint_requests = 0;
    double last_residual = 0.0;
    int64_t *B_index = nullptr, *N_index = nullptr, *perm = nullptr;
    uint8_t *kindv = nullptr, *Nb = nullptr, *dpos = nullptr;
    int32_t *used = nullptr, *bidx = nullptr;
    DevState *st = nullptr;
    DevState *h_st = nullptr;  // pinned
    DevState *h_look = nullptr;  // pinned, 2 slots: status read-backs of the look-ahead loop
    hipEvent_t look_ev[2] = {nullptr, nullptr};
    // pricing shard (column-block sharding across ranks; world = 1 on a single GPU)
    int rank = 0, world = 1, nbs = 1;
    int64_t seg = 0;
    hipStream_t own_stream = nullptr;
    bool need_dleave = true;
    // column-sharded storage of A_N (ellp_shard.inc)
    bool colshard = false;
    int64_t own0 = 0, own1 = 0;          // nonbasic positions stored and priced here
    double *A_N_store = nullptr;          // the allocation behind the virtual base e->A_N
    double *packs = nullptr, *aq_cur = nullptr;  // gathered packs (world * pack_doubles), entering column
    int32_t *se_perm = nullptr;  // steepest edge: device flag "the starting basis is a signed permutation"
    bool sel_in_ftran = false;  // this k_ftran2 launch runs the compact selection itself (launch_sharded_iteration)
    bool sel_commit = false;    // ... and commits the mailbox generation of the exchange in front of it
    int64_t slot_doubles = 0;             // doubles per exchange slot = max(pack, full pricing segment)
    int transport = 0;                    // 0 none, 1 RCCL, 2 peer-to-peer mailbox, 3 host callback (tests, gloo)
    ellp_exchange_fn xfn = nullptr;
    void *xuser = nullptr;
    std::vector<char> xhost;              // staging for the callback transport
    // mailbox
    double *mbox = nullptr;               // own mailbox (uncached): [parity][sender] slots
    unsigned long long *mflags = nullptr; // own flags: [parity][sender]
    double **d_peer_slots = nullptr;      // device arrays of the peers' mapped bases
    unsigned long long **d_peer_flags = nullptr;
    std::vector<void *> ipc_opened;
    uint64_t full_exchanges = 0, column_requests = 0;
    // small LPs (m <= 128): the reference's LU-per-iteration loop in one persistent workgroup (ellp_small.inc)
    bool small = false;      // run() uses k_small
    bool w_valid = true;     // the explicit inverse W (not kept by k_small) matches A_B
    bool exact_large_only = false;  // after a redo above 1,024 rows: every loop body on a fresh LU (run_exact_large), until the next phase
    size_t small_lds = 0;
    int pp_P = 0;            // partial pricing: number of segments (<= 1: off)
    int64_t pp_S = 0;        // positions per segment
    char *slab = nullptr;  // see dmalloc
    size_t slab_size = 0, slab_used = 0;
    unsigned long long *small_stamps = nullptr;  // ELLP_SMALL_STAMPS: per-phase tick sums of k_small, printed at destroy
    int small_nt = SMALL_THREADS;  // workgroup size of k_small for this LP (small_threads)
    // unit columns (PriceArgs::vs_row): per VARIABLE, made once at creation; null = the pricing kernels stream everything
    int32_t *vs_row = nullptr;
    double *vs_val = nullptr;
    uint8_t *pos_hint = nullptr;
    int dual_maxviol = 0;  // ELLP_FLAG_DUAL_MAX_VIOLATION
    int dual_bflip = 0;    // ELLP_FLAG_DUAL_BOUND_FLIPPING: the long-step ratio test (LU-per-iteration kernels only)
    long long *bf_list = nullptr;  // ... the positions passed in an iteration (nN entries)
    bool se = false;       // ELLP_FLAG_PRIMAL_STEEPEST_EDGE (ellp_se.inc): three launches + one transposed GEMV per iteration
    double *se_gamma = nullptr, *se_rho = nullptr, *se_v = nullptr;
    double *se_vpart = nullptr;  // per row block of k_update2: its share of v = B^-T d (null: the separate transposed GEMV)
    int64_t unit_columns = 0;  // how many variables have one (diagnostics)
    // 128 < m <= 1024: the same loop with its factors in global memory (ellp_mid.inc); `small` is set as well, so
    // that everything that asks "is there an explicit inverse" keeps working unchanged
    bool mid = false;
    size_t mid_lds = 0;
    int mid_nt = 0;
    int64_t ldn = 0;
    double *LUa = nullptr, *Ut = nullptr, *A_Nt = nullptr;
    // certified hybrid: the explicit-inverse loop with a pivot guard; every terminal status and every
    // guarded iteration is handed to the LU-per-iteration kernel (k_mid) for up to exact_K iterations (exact_takeover)
    bool hybrid = false;
    double guard_abs = 0.0;
    int exact_K = 8;
    uint64_t hy_guards = 0, hy_certs = 0, hy_disagree = 0, hy_exact_iters = 0, hy_rebuilds = 0, hy_redos = 0;
    // "certify or redo": the start of the phase (taken at the first run() after creation / a hand-off), so that a solve whose
    // end point violates an invariant of the reference's loop can be repeated by the exact kernel from where it began
    struct Snapshot {
        bool valid = false;
        std::vector<double> x, y, d, c_B, c_N;
        std::vector<int64_t> B, N;
        std::vector<uint8_t> Nb;
        double obj = 0.0;
    } snap;
    double *inv_out = nullptr;
    // above 1,024 rows (ellp_exact.inc): terminal statuses are certified by one iteration whose u / rho and d come from a fresh
    // LU of the basis (no pivot guard, no redo at these sizes)
    bool cert_large = false;
    bool guard_off = false;   // an exact iteration is being enqueued: its pivots are the reference's, however small
    bool luw_ready = false;
    EllpLuWork luw{};
    double *ex_rhs = nullptr, *ex_sol = nullptr, *ex_rho = nullptr;
    int *ex_fail = nullptr;
    const double *price_rho_ovr = nullptr;  // launch_price<1>: PriceArgs::rho_ovr for the next launch
    uint64_t hy_uncertified = 0;
    // the LP is a "box problem" — every bound TwoSided or Fixed and b = 0: the shape of DualPhase1's LP (dual_problem.rs:89-131),
    // whose dual objective is minus the dual infeasibility of the original problem, i.e. <= 0 and = 0 at a feasible end
    bool box_problem = false;
    double ill_tol = 0.0;  // reactive maintenance threshold (small LPs only, see ellp_engine_create)
    int maint_chain = 0;   // > 0: refresh again after the next single iteration
    int drift_every = 0;   // iterations between two drift checks of B^-1 (0: off)
    double drift_tol = 0.0;
    uint64_t since_drift = 0, drift_checks = 0;
    // launch geometry
    int cpb = 1, nblocks = 1, priceT = 1;
    bool price_nt = false;
    bool price_wave = false;  // wave-per-column pricing (cache-resident A_N, rows long enough to fill a wave)
    int upd_rows = 4, upd_blocks = 1;    // refactorisation kernels (<= UPD_ROWS rows per block)
    int upd2_rows = 4, upd2_blocks = 1;  // k_update2 (<= 2*UPD_ROWS)
    int ftran_blocks = 1;
    int btran_tiles = 1, btran_rows = 1;
    int upd_stage = 1;
    size_t upd_lds = 0, ftran_lds = 0;
    // loop bookkeeping
    uint64_t since_refactor = 0, since_btran = 0;
    uint64_t enqueued = 0;  // iterations enqueued since the counters were last reconciled with DevState::iters
    uint64_t iters_seen = 0;  // DevState::iters at that moment
    int refactor_period = 0;
    int btran_refresh = 32;
    uint64_t refactors = 0;
    bool u_valid = false;
    double t_setup = 0.0;
    // profiling
    std::vector<hipEvent_t> ev_pool;
    struct Pending { int id; hipEvent_t a, b; };
    std::vector<Pending> pending;
    size_t ev_next = 0;
    double ev_overhead_ms = -1.0;
    double kernel_ms[ELLP_K_COUNT] = {0};
    uint64_t kernel_calls[ELLP_K_COUNT] = {0};
    std::vector<void *> allocs;
    // direct RCCL exchange (ellp_engine_comm_init)
    ncclComm_t comm = nullptr;
    const struct RcclApi *rccl = nullptr;
};

namespace {

void set_err(char *errbuf, size_t len, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
void set_err(char *errbuf, size_t len, const char *fmt, ...) {
    if (!errbuf || !len) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(errbuf, len, fmt, ap);
    va_end(ap);
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            set_err(errbuf, errlen, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, \
                    __LINE__, #expr);                                                             \
            return ELLP_ERR_DEVICE;                                                               \
        }                                                                                         \
    } while (0)

// Device memory of an engine.  Arrays up to 512 KB come out of one 6 MB slab (an engine has ~45 arrays; for a
// small LP their hipMalloc / hipFree calls were 3 of the 4.8 ms of an AFIRO solve), larger ones are allocations
// of their own.  Everything is released by ellp_engine_destroy.
template <typename T>
hipError_t dmalloc(ellp_engine *e, T **p, size_t count) {
    const size_t bytes = ((count ? count : 1) * sizeof(T) + 255) / 256 * 256;
    if (bytes <= (512u << 10)) {
        if (!e->slab) {
            void *sl = nullptr;
            if (hipMalloc(&sl, 6u << 20) == hipSuccess) {
                e->slab = static_cast<char *>(sl);
                e->slab_size = 6u << 20;
                e->slab_used = 0;
                e->allocs.push_back(sl);
            } else {
                (void)hipGetLastError();
            }
        }
        if (e->slab && e->slab_used + bytes <= e->slab_size) {
            *p = reinterpret_cast<T *>(e->slab + e->slab_used);
            e->slab_used += bytes;
            return hipSuccess;
        }
    }
    void *q = nullptr;
    hipError_t rc = hipMalloc(&q, bytes);
    if (rc == hipSuccess) {
        e->allocs.push_back(q);
        *p = static_cast<T *>(q);
    }
    return rc;
}

// `neu` (a hipMalloc of its own) takes the place of the engine array `old`
inline void replace_alloc(ellp_engine *e, void *old, void *neu) {
    for (auto &p : e->allocs)
        if (p == old) {
            (void)hipFree(p);
            p = neu;
            return;
        }
    e->allocs.push_back(neu);  // `old` lives in the slab
}

inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Prof {
    ellp_engine *e;
    int id;
    bool on;
    hipEvent_t a{}, b{};
    Prof(ellp_engine *e_, int id_) : e(e_), id(id_), on(e_->opts.profile != 0) {
        if (!on) return;
        if (e->ev_next + 2 > e->ev_pool.size()) {
            for (int k = 0; k < 64; ++k) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) { on = false; return; }
                e->ev_pool.push_back(ev);
            }
        }
        a = e->ev_pool[e->ev_next++];
        b = e->ev_pool[e->ev_next++];
        (void)hipEventRecord(a, e->stream);
    }
    ~Prof() {
        if (!on) return;
        (void)hipEventRecord(b, e->stream);
        e->pending.push_back({id, a, b});
    }
};

// What an event bracket adds to the one kernel inside it, relative to rocprofv3's duration of
// that kernel.  Measured on MI355X / ROCm 7.2 with tools/event_cal.hip and with bench.py run
// under rocprofv3: 2.4 us around a 23.6 us kernel, 2.45 us around a null kernel, 2.2-2.5 us
// around k_price / k_ftran2 / k_update2.  It is a property of the event markers, not of the
// kernel, and it cannot be measured live without the profiler: an EMPTY pair reads 4.4 us, and
// differences between one launch and N back-to-back launches include a dispatch gap that varies
// from 0.35 to 2 us with the kernel.  So the constant below is subtracted (2.3 us errs on the
// side of longer kernels); ELLP_EVENT_BRACKET_US overrides it, ELLP_PROF_RAW=1 reports raw
// brackets.  profiles/ holds the rocprofv3 summaries the bench's figures are checked against.
constexpr double EVENT_BRACKET_US = 2.3;

void prof_calibrate(ellp_engine *e) {
    if (e->ev_overhead_ms >= 0.0 || !e->opts.profile) return;
    e->ev_overhead_ms = EVENT_BRACKET_US * 1e-3;
    if (const char *v = getenv("ELLP_PROF_RAW"); v && v[0] == '1') e->ev_overhead_ms = 0.0;
    if (const char *v = getenv("ELLP_EVENT_BRACKET_US"); v && v[0]) {
        const double us = atof(v);
        if (us >= 0.0 && us < 20.0) e->ev_overhead_ms = us * 1e-3;
    }
}

void prof_collect(ellp_engine *e) {
    prof_calibrate(e);
    for (auto &p : e->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            double v = (double)ms - (p.id == ELLP_K_REFACTOR ? 0.0 : e->ev_overhead_ms);
            e->kernel_ms[p.id] += v > 0.0 ? v : 0.0;
            e->kernel_calls[p.id] += 1;
        }
    }
    e->pending.clear();
    e->ev_next = 0;
}

// ---- launches -----------------------------------------------------------------------------
template <int MODE>
void launch_price(ellp_engine *e) {
    PriceArgs a{};
    a.A_N = e->A_N;
    a.W0 = e->W;
    a.W1 = e->W2;
    a.u = e->u;
    a.c_N = e->c_N;
    a.Nb = e->Nb;
    a.N_index = e->N_index;
    a.dd = e->dd;
    a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb};
    a.st = e->st;
    a.ld = e->ld;
    a.nN = e->nN;
    a.cpb = e->cpb;
    a.block0 = e->rank * e->nbs;
    a.eps = e->eps;
    a.pp_on = e->pp_P > 1 ? 1 : 0;
    a.vs_row = !(e->opts.flags & ELLP_FLAG_DENSE_PRICING) ? e->vs_row : nullptr;
    a.vs_val = a.vs_row ? e->vs_val : nullptr;
    a.rho_ovr = MODE == 1 ? e->price_rho_ovr : nullptr;
    if (MODE == 1 && e->dual_fold) {
        a.dp_seq = e->dual_seq;
        a.dp_lrow = e->binfo; a.dp_ldelta = e->bmin; a.dp_lside = e->bmin + e->m; a.dp_d = e->d;
        a.dp_nrb = (int)((e->m + UPD_ROWS - 1) / UPD_ROWS);
        a.dp_maxviol = e->dual_maxviol;
        a.dp_A_N = e->A_N; a.dp_A_B = e->A_B; a.dp_c_B = e->c_B; a.dp_c_N = e->c_N; a.dp_x = e->x; a.dp_dd = e->dd;
        a.dp_B_index = e->B_index; a.dp_N_index = e->N_index; a.dp_Nb = e->Nb;
        a.dp_trace = Trace{e->trace_obj, e->trace_it, e->trace_len};
    }
    int mine = e->nblocks - a.block0;
    if (mine > e->nbs) mine = e->nbs;
    if (mine <= 0) mine = 1;  // empty shard (more ranks than pricing blocks): the block only serves DevState::tiny
    dim3 g(mine), b(256);
    if (e->price_wave) {
        hipLaunchKernelGGL((k_price_wave<MODE>), g, b, 0, e->stream, a);
        return;
    }
    if (e->price_nt) {
        switch (e->priceT) {
        case 1: hipLaunchKernelGGL((k_price<1, MODE, true>), g, b, 0, e->stream, a); break;
        case 2: hipLaunchKernelGGL((k_price<2, MODE, true>), g, b, 0, e->stream, a); break;
        case 4: hipLaunchKernelGGL((k_price<4, MODE, true>), g, b, 0, e->stream, a); break;
        case 8: hipLaunchKernelGGL((k_price<8, MODE, true>), g, b, 0, e->stream, a); break;
        default: hipLaunchKernelGGL((k_price<16, MODE, true>), g, b, 0, e->stream, a); break;
        }
    } else {
        switch (e->priceT) {
        case 1: hipLaunchKernelGGL((k_price<1, MODE, false>), g, b, 0, e->stream, a); break;
        case 2: hipLaunchKernelGGL((k_price<2, MODE, false>), g, b, 0, e->stream, a); break;
        case 4: hipLaunchKernelGGL((k_price<4, MODE, false>), g, b, 0, e->stream, a); break;
        case 8: hipLaunchKernelGGL((k_price<8, MODE, false>), g, b, 0, e->stream, a); break;
        default: hipLaunchKernelGGL((k_price<16, MODE, false>), g, b, 0, e->stream, a); break;
        }
    }
}

template <int MODE>
void launch_ftran2(ellp_engine *e) {
    Ftran2Args a{};
    a.W0 = e->W; a.W1 = e->W2; a.A_N = e->A_N;
    a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb};
    a.N_index = e->N_index; a.B_index = e->B_index; a.Nb = e->Nb; a.kind = e->kindv;
    a.x = e->x; a.lb = e->lb; a.ub = e->ub;
    a.d = e->d; a.lam = e->lam; a.bidx = e->bidx; a.dpos = e->dpos;
    a.st = e->st; a.m = e->m; a.ld = e->ld; a.nN = e->nN; a.nblocks = e->nblocks; a.cpb = e->cpb; a.eps = e->eps;
    a.aq_cur = (MODE == 0 && e->colshard) ? e->aq_cur : nullptr;
    if (MODE == 0 && e->colshard && e->sel_in_ftran) {
        a.sel_packs = e->packs; a.aq_out = e->aq_cur; a.sel_world = e->world; a.mbox_commit = e->sel_commit ? 1 : 0;
    }
    a.pp_on = (MODE == 0 && e->pp_P > 1) ? 1 : 0;
    const dim3 g(e->ftran_blocks), b(256);
    const int64_t nt = ((e->ld >> 1) + 63) / 64;  // double2 per lane for one row
    if (nt <= 4) hipLaunchKernelGGL((k_ftran2<MODE, 4>), g, b, e->ftran_lds, e->stream, a);
    else if (nt <= 8) hipLaunchKernelGGL((k_ftran2<MODE, 8>), g, b, e->ftran_lds, e->stream, a);
    else if (nt <= 16) hipLaunchKernelGGL((k_ftran2<MODE, 16>), g, b, e->ftran_lds, e->stream, a);
    else hipLaunchKernelGGL((k_ftran2<MODE, 0>), g, b, e->ftran_lds, e->stream, a);
}

template <int MODE>
void launch_update2(ellp_engine *e, int update_u) {
    Update2Args a{};
    a.W0 = e->W; a.W1 = e->W2; a.d = e->d; a.lam = e->lam; a.bidx = e->bidx; a.dpos = e->dpos; a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb};
    a.u = e->u; a.u_alt = e->u + e->ld; a.A_N = e->A_N; a.A_B = e->A_B; a.c_B = e->c_B; a.c_N = e->c_N; a.x = e->x; a.y = e->y; a.dd = e->dd;
    a.lb = e->lb; a.ub = e->ub; a.kind = e->kindv; a.B_index = e->B_index; a.N_index = e->N_index; a.Nb = e->Nb;
    a.st = e->st; a.m = e->m; a.ld = e->ld; a.nN = e->nN; a.rows_per_block = e->upd2_rows; a.update_u = update_u;
    a.stage_lds = e->upd_stage; a.eps = e->eps;
    a.ill_tol = e->ill_tol;
    a.guard_abs = ((e->hybrid || e->cert_large) && !e->guard_off) ? e->guard_abs : 0.0;
    a.aq_cur = (MODE == 0 && e->colshard) ? e->aq_cur : nullptr; a.own0 = e->own0; a.own1 = e->own1;
    a.count_iter = (MODE == 0 && e->lagged) ? 0 : 1;
    a.maxviol = e->dual_maxviol;
    a.se_gamma = (MODE == 0 && e->se) ? e->se_gamma : nullptr;
    a.se_rho = (MODE == 0 && e->se) ? e->se_rho : nullptr;
    a.se_vpart = (MODE == 0 && e->se) ? e->se_vpart : nullptr;
    a.trace = Trace{e->trace_obj, e->trace_it, e->trace_len};
    const dim3 g(e->upd2_blocks + (MODE == 0 ? 2 : 3)), b(256);
    const size_t lds = MODE == 0 ? e->upd_lds : 0;
    const int64_t nr = ((e->ld >> 1) + 255) / 256;  // double2 per thread per row
    if (nr <= 1) hipLaunchKernelGGL((k_update2<MODE, 1>), g, b, lds, e->stream, a);
    else if (nr <= 2) hipLaunchKernelGGL((k_update2<MODE, 2>), g, b, lds, e->stream, a);
    else if (nr <= 4) hipLaunchKernelGGL((k_update2<MODE, 4>), g, b, lds, e->stream, a);
    else if (nr <= 8) hipLaunchKernelGGL((k_update2<MODE, 8>), g, b, lds, e->stream, a);
    else hipLaunchKernelGGL((k_update2<MODE, 0>), g, b, lds, e->stream, a);
}

void launch_btran(ellp_engine *e) {
    BtranArgs a{e->W, e->W2, e->c_B, e->upart, e->u, e->u + e->ld, e->st, e->m, e->ld, e->btran_rows, e->btran_tiles};
    const int64_t half = e->ld >> 1;
    dim3 g((unsigned)((half + 255) / 256), (unsigned)e->btran_tiles);
    hipLaunchKernelGGL(k_btran_part, g, dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_btran_reduce, dim3((unsigned)((e->ld + 255) / 256)), dim3(256), 0, e->stream, a);
}

void launch_refactor_columnwise(ellp_engine *e) {
    Prof p(e, ELLP_K_REFACTOR);
    RefArgs a{e->W, e->W2, e->d, e->A_B, e->used, e->perm, e->st, e->m, e->ld, e->upd_rows,
              e->kind == ELLP_ENGINE_PRIMAL ? e->eps : 0.0};
    hipLaunchKernelGGL(k_ref_begin, dim3(1), dim3(1), 0, e->stream, a);
    hipLaunchKernelGGL(k_ref_init, dim3(1024), dim3(256), 0, e->stream, a);
    for (int64_t k = 0; k < e->m; ++k) {
        hipLaunchKernelGGL(k_ref_ftran, dim3(e->ftran_blocks), dim3(256), 0, e->stream, a);
        hipLaunchKernelGGL(k_ref_pick, dim3(1), dim3(1024), 0, e->stream, a);
        hipLaunchKernelGGL(k_ref_update, dim3(e->upd_blocks), dim3(256), 0, e->stream, a);
    }
    hipLaunchKernelGGL(k_ref_permute, dim3((unsigned)e->m), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_ref_finish, dim3(1), dim3(1), 0, e->stream, a);
    e->refactors += 1;
    e->since_refactor = 0;
    e->u_valid = false;
}

// Blocked rebuild (ellp_rebuild.inc).  One host synchronisation: after the probe for a generalised
// permutation matrix (the artificial / slack bases every phase 1 starts from), to know whether the
// general elimination has to be enqueued at all.  Rebuilds are rare (engine creation, a refused refresh).
void launch_refactor(ellp_engine *e) {
    const bool legacy = e->bl_Cpart == nullptr || (getenv("ELLP_REBUILD") && !strcmp(getenv("ELLP_REBUILD"), "columnwise"));
    if (legacy) {
        launch_refactor_columnwise(e);
        return;
    }
    Prof p(e, ELLP_K_REFACTOR);
    const int64_t m = e->m;
    BlArgs a{};
    a.W0 = e->W; a.W1 = e->W2; a.A_B = e->A_B; a.Cpart = e->bl_Cpart; a.C0 = e->bl_C; a.C1 = e->bl_C + m * BL_NB;
    a.V0 = e->bl_V; a.V1 = e->bl_V + m * BL_NB; a.Vs = e->bl_Vs; a.Wp = e->bl_Wp; a.used = e->used; a.perm = e->perm;
    a.Pm = e->bl_Pm; a.st = e->st; a.m = m; a.ld = e->ld; a.splits = e->bl_splits;
    a.ksplit = (int)round_up((m + e->bl_splits - 1) / e->bl_splits, 16);
    a.eps = e->kind == ELLP_ENGINE_PRIMAL ? e->eps : 0.0;
    e->refactors += 1;
    e->since_refactor = 0;
    e->u_valid = false;
    // ---- shortcut: one nonzero per column, distinct rows
    hipLaunchKernelGGL(k_bl_probe, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, e->stream, a, e->bl_nzval, e->bl_nzrow, e->bl_nzcnt);
    hipLaunchKernelGGL(k_bl_perm_check, dim3(1), dim3(1024), 0, e->stream, a, e->bl_nzval, e->bl_nzrow, e->bl_nzcnt);
    hipLaunchKernelGGL(k_bl_perm_fill, dim3((unsigned)m), dim3(256), 0, e->stream, a, e->bl_nzval, e->bl_nzrow);
    DevState probe;
    if (hipMemcpyAsync(&probe, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
        return;
    static const int32_t zero = 0;  // static: the copy may still be in flight when this function returns
    (void)hipMemcpyAsync(&e->st->do_update, &zero, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    if (probe.status != ST_RUNNING) return;  // singular (or the engine is not running: the kernels did nothing)
    if (probe.do_update) {
        e->rebuild_shortcuts += 1;
        return;
    }
    // ---- general case
    RefArgs ra{e->W, e->W2, e->d, e->A_B, e->used, e->perm, e->st, e->m, e->ld, e->upd_rows, a.eps};
    hipLaunchKernelGGL(k_ref_init, dim3(1024), dim3(256), 0, e->stream, ra);
    const int bl_nt = (getenv("ELLP_BL_NT") && atoi(getenv("ELLP_BL_NT")) == 1024) ? 1024 : 512;
    int nbw_max = m <= 2048 ? 16 : (m <= 4096 ? 8 : 4);
    if (const char *v = getenv("ELLP_BL_NBW"); v && v[0]) {  // tests: a narrower sub-panel than the size needs
        const int w = atoi(v);
        if ((w == 8 || w == 4) && w < nbw_max) nbw_max = w;
    }
    for (int64_t k0 = 0; k0 < m; k0 += BL_NB) {
        a.k0 = (int)k0;
        a.nbc = (int)((m - k0) < BL_NB ? (m - k0) : BL_NB);
        a.sel = 0;
        hipLaunchKernelGGL(k_bl_gemm1, dim3((unsigned)((m + 63) / 64), (unsigned)a.splits), dim3(256), 0, e->stream, a);
        hipLaunchKernelGGL(k_bl_sum, dim3(256), dim3(256), 0, e->stream, a);
        for (int j0 = 0; j0 < a.nbc; j0 += nbw_max) {
            a.j0 = j0;
            a.nbw = (a.nbc - j0) < nbw_max ? (a.nbc - j0) : nbw_max;
            a.nacc = j0;
            if (bl_nt == 512) {  // two waves per SIMD, 256 VGPRs each: 4 / 8 / 16 rows per thread
                if (nbw_max == 16) hipLaunchKernelGGL((k_bl_factor<16, 4, 512>), dim3(1), dim3(512), 0, e->stream, a);
                else if (nbw_max == 8) hipLaunchKernelGGL((k_bl_factor<8, 8, 512>), dim3(1), dim3(512), 0, e->stream, a);
                else hipLaunchKernelGGL((k_bl_factor<4, 16, 512>), dim3(1), dim3(512), 0, e->stream, a);
            } else {
                if (nbw_max == 16) hipLaunchKernelGGL((k_bl_factor<16, 2, 1024>), dim3(1), dim3(1024), 0, e->stream, a);
                else if (nbw_max == 8) hipLaunchKernelGGL((k_bl_factor<8, 4, 1024>), dim3(1), dim3(1024), 0, e->stream, a);
                else hipLaunchKernelGGL((k_bl_factor<4, 8, 1024>), dim3(1), dim3(1024), 0, e->stream, a);
            }
            hipLaunchKernelGGL(k_bl_apply, dim3((unsigned)((m + 7) / 8)), dim3(256), 0, e->stream, a);
            a.sel ^= 1;
        }
        hipLaunchKernelGGL(k_bl_gather, dim3((unsigned)a.nbc), dim3(256), 0, e->stream, a);
        hipLaunchKernelGGL(k_bl_gemm2, dim3((unsigned)((e->ld + 127) / 128), (unsigned)((m + 63) / 64)), dim3(256), 0, e->stream, a);
    }
    hipLaunchKernelGGL(k_ref_permute, dim3((unsigned)e->m), dim3(256), 0, e->stream, ra);
    hipLaunchKernelGGL(k_ref_finish, dim3(1), dim3(1), 0, e->stream, ra);
}

// One Newton-Schulz step on the current inverse, enqueued without any host synchronisation: the
// residual max|I - A_B W| of the first GEMM is folded on the device (k_resid_reduce), which also
// decides whether the step is safe; if it is not, the step is skipped, DevState::need_rebuild is set
// and the loop stops with a maintenance request that the host services by rebuilding from A_B.
void launch_refresh(ellp_engine *e) {
    Prof p(e, ELLP_K_REFACTOR);
    GemmArgs a{e->W, e->W2, e->A_B, e->T, e->resid, e->st, e->m, e->ld};
    const unsigned nt = (unsigned)((e->m + 127) / 128);
    static const bool valu = [] {  // measurement: ELLP_GEMM=valu runs round 1's vector-ALU GEMM
        const char *v = getenv("ELLP_GEMM");
        return v && strcmp(v, "valu") == 0;
    }();
    if (valu) hipLaunchKernelGGL(k_gemm128<0>, dim3(nt, nt), dim3(256), 0, e->stream, a);
    else hipLaunchKernelGGL(k_gemm_mfma<0>, dim3(nt, nt), dim3(512), 0, e->stream, a);
    hipLaunchKernelGGL(k_resid_reduce, dim3(1), dim3(256), 0, e->stream, a, (int)(nt * nt));
    if (valu) hipLaunchKernelGGL(k_gemm128<1>, dim3(nt, nt), dim3(256), 0, e->stream, a);
    else hipLaunchKernelGGL(k_gemm_mfma<1>, dim3(nt, nt), dim3(512), 0, e->stream, a);
    hipLaunchKernelGGL(k_copy_to_other, dim3(1024), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_refresh_finish, dim3(1), dim3(1), 0, e->stream, a);
    e->refreshes += 1;
    e->since_refactor = 0;
    e->u_valid = false;
}

// Default maintenance period, from a cost model: a refresh costs T_r ~ 2*(2 m^3)/25 TFLOP/s + 60 us
// (two GEMM launches, a residual read-back with a host sync), an iteration T_i ~ (8 ld |N| + 24 m ld)
// / 5 TB/s + 15 us, and the period is the smallest one whose refreshes stay within a BUDGET of the
// loop time: 30 % up to m = 1024, falling linearly to 3 % at m = 2000 and beyond (config 3: 1000
// iterations, as measured), never below 16.  Why so generous on small and mid-size LPs: the reference
// factorises afresh every iteration, and the error of B^-1 a_q is cond(A_B) times that of B^-1 — a
// tall 500 x 100 LP of the benign synthetic family left the oracle's path after 166 pivots with B^-1
// untouched and after 162 with a period of 62 (x off by 2e-9 against EPS = 1e-10; the error grew
// 25-fold within 9 pivots at cond 6e4), and stays on it for 1200 pivots with a period of 16.  Beyond
// the model, the drift monitor (k_drift_part) asks for a refresh when the LP needs one.
int64_t default_period(const ellp_engine *e) {
    const double m = (double)e->m, ld = (double)e->ld, nN = (double)(e->nN > 0 ? e->nN : 1);
    const double t_refresh = 4.0 * m * m * m / 25e12 + 60e-6;
    const double t_iter = (8.0 * ld * nN + 24.0 * m * ld) / 5e12 + 15e-6;
    double budget = 0.30;
    if (m > 1024.0) budget = m >= 2000.0 ? 0.03 : 0.30 - 0.27 * (m - 1024.0) / 976.0;
    int64_t p = (int64_t)std::ceil(t_refresh / (budget * t_iter));
    if (p < 16) p = 16;
    if (p > 1000) p = 1000;
    return p;
}

// periodic maintenance of B^-1: Newton-Schulz refresh, full rebuild only if that is not safe
void launch_dleave(ellp_engine *e);
void launch_resync(ellp_engine *e, int force);

// x_B from the freshly maintained B^-1 (see k_resync_part)
void launch_resync(ellp_engine *e, int force) {
    if (e->nN <= 0) return;
    if (e->colshard) return;  // b - A_N x_N would need every rank's columns (a reduction over the ranks): not done
    ResyncArgs a{e->A_N, e->W, e->W2, e->b_dev, e->x, e->xg, e->tvec, e->upart, e->cand, e->maxbits, e->B_index,
                 e->N_index, e->st, e->m, e->ld, e->nN, 0, e->btran_tiles, force};
    (void)hipMemsetAsync(e->maxbits, 0, 2 * sizeof(unsigned long long), e->stream);
    a.cols_per_tile = (int)((e->nN + e->btran_tiles - 1) / e->btran_tiles);
    const int64_t half = e->ld >> 1;
    hipLaunchKernelGGL(k_resync_gather, dim3((unsigned)((e->nN + 255) / 256)), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_resync_part, dim3((unsigned)((half + 255) / 256), (unsigned)e->btran_tiles), dim3(256), 0,
                       e->stream, a);
    hipLaunchKernelGGL(k_resync_rhs, dim3((unsigned)((e->ld + 255) / 256)), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_resync_xb, dim3((unsigned)((e->m + 3) / 4)), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_resync_apply, dim3((unsigned)((e->m + 255) / 256)), dim3(256), 0, e->stream, a);
    if (e->kind == ELLP_ENGINE_DUAL) launch_dleave(e);  // the leaving row is chosen from x (dual…:200-236)
    e->resyncs += 1;
    if (getenv("ELLP_RESYNC_DEBUG")) {  // diagnostics: what the resync found
        unsigned long long hb[2] = {0, 0};
        (void)hipMemcpyAsync(hb, e->maxbits, sizeof(hb), hipMemcpyDeviceToHost, e->stream);
        (void)hipStreamSynchronize(e->stream);
        double dv[2];
        memcpy(dv, hb, sizeof(dv));
        fprintf(stderr, "ellp resync %llu: max|cand - x_B| %.3e, max|x_B| %.3e\n", (unsigned long long)e->resyncs, dv[0], dv[1]);
    }
}

// Maintenance of B^-1: Newton-Schulz refresh, full rebuild if that is not safe, then x_B is checked
// against the fresh inverse (launch_resync).  `reactive` (asked for by the device, or the follow-up of
// such a request) is kept for diagnostics: resynchronising only then was tried and is worse (netlib
// ADLITTLE / BLEND in 60 variable orders, dual: 4 wrong outcomes instead of 1).
void launch_dual_close(ellp_engine *e);
void maintain_inverse(ellp_engine *e, bool reactive = false, bool rebuild = false) {
    // two-launch pipeline: an open iteration (priced, FTRAN done, ratio test not folded) was decided with
    // the inverse that is about to be replaced — drop it; the next k_price2 starts afresh (use_pend = 0) and
    // the iteration is priced again from the maintained inverse.  Its eta-update half is already in B^-1.
    launch_dual_close(e);  // a fused dual iteration is completed, not dropped: its pivot is in B^-1 already
    if (e->lagged) hipLaunchKernelGGL(k_drop_open, dim3(1), dim3(1), 0, e->stream, e->st);
    e->lag_open = false;
    if (rebuild) launch_refactor(e);
    else launch_refresh(e);
    const char *mode = getenv("ELLP_RESYNC");  // diagnostics: "0" never, "1" only on reactive maintenance
    const bool want = mode && mode[0] == '0' ? false : (mode && mode[0] == '1' ? reactive : true);
    if (want) launch_resync(e, 0);
}

// After a read-back with the stream drained: iterations that were enqueued behind a stop (a final
// status or a maintenance request) returned at entry and did not happen, but the host counted them
// when it enqueued them.  Take them back, so that the maintenance schedule follows the iterations
// that really ran (DevState::iters counts every loop body entered).
void reconcile_counters(ellp_engine *e) {
    const uint64_t ran = e->h_st->iters >= e->iters_seen ? e->h_st->iters - e->iters_seen : 0;
    const uint64_t lost = e->enqueued > ran ? e->enqueued - ran : 0;
    if (lost) {
        e->since_refactor = e->since_refactor > lost ? e->since_refactor - lost : 0;
        e->since_btran = e->since_btran > lost ? e->since_btran - lost : 0;
        e->since_drift = e->since_drift > lost ? e->since_drift - lost : 0;
    }
    e->enqueued = 0;
    e->iters_seen = e->h_st->iters;
}

// two-launch pipeline: k_ftran_eta reports the end of the solve in DevState::fin and leaves it to the next
// kernel's leader to make it the status; after a drained read-back the host can do that itself
void adopt_fin(ellp_engine *e) {
    if (e->h_st->status != ST_RUNNING || !e->h_st->fin) return;
    static int32_t fin_status;  // static: asynchronous copy
    fin_status = e->h_st->fin - 1;
    (void)hipMemcpyAsync(&e->st->status, &fin_status, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    (void)hipStreamSynchronize(e->stream);
    e->h_st->status = fin_status;
}

// After a status read-back: if a kernel asked for maintenance, do it, re-arm the loop and report
// true (the caller keeps going).  The pivot that raised the request is committed (k_update2 raises the
// flag after its bookkeeping); what is void is the rest of the batch behind it.  The device is
// re-armed (status = RUNNING, tiny = 0) BEFORE the maintenance kernels are enqueued: all of them
// (k_gemm128, k_copy_to_other, k_ref_*, k_dleave) return at entry on any other status, so servicing
// the request under ST_NEED_MAINT would refresh nothing and re-select the dual's leaving row from
// nothing.
bool service_maintenance_request(ellp_engine *e) {
    // the request is the status (a later pricing launch of the batch saw the flag) or still the flag
    // (the batch ended with the k_update2 that raised it)
    if (e->h_st->status != ST_NEED_MAINT && !(e->h_st->status == ST_RUNNING && e->h_st->tiny)) return false;
    const int32_t running = ST_RUNNING, zero = 0;
    const bool rebuild = e->h_st->need_rebuild != 0;  // a refresh found B^-1 too far off for Newton-Schulz
    (void)hipMemcpyAsync(&e->st->status, &running, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    (void)hipMemcpyAsync(&e->st->tiny, &zero, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    (void)hipMemcpyAsync(&e->st->need_rebuild, &zero, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    e->h_st->status = ST_RUNNING;
    e->h_st->tiny = 0;
    e->h_st->need_rebuild = 0;
    maintain_inverse(e, true, rebuild);
    e->maint_chain = 1;
    (void)hipStreamSynchronize(e->stream);
    e->maint_requests += 1;
    return true;
}

// between k_ftran2 and k_update2 of an iteration: see k_drift_part
void launch_drift_check(ellp_engine *e) {
    if (e->drift_every <= 0) return;
    if (++e->since_drift < (uint64_t)e->drift_every) return;
    e->since_drift = 0;
    e->drift_checks += 1;
    DriftArgs a{e->A_B, e->A_N, e->d, e->colshard ? e->aq_cur : nullptr, e->upart, e->st, e->m, e->ld, e->btran_rows, e->btran_tiles, e->drift_tol};
    const int64_t half = e->ld >> 1;
    hipLaunchKernelGGL(k_drift_part, dim3((unsigned)((half + 255) / 256), (unsigned)e->btran_tiles), dim3(256), 0,
                       e->stream, a);
    hipLaunchKernelGGL(k_drift_reduce, dim3(1), dim3(1024), 0, e->stream, a);
}

// ---- two-launch pipeline (ellp_lagged.inc) -------------------------------------------------------
void launch_price2(ellp_engine *e, int use_pend) {
    Price2Args a{};
    a.p.A_N = e->A_N; a.p.W0 = e->W; a.p.W1 = e->W2; a.p.u = nullptr; a.p.c_N = e->c_N; a.p.Nb = e->Nb;
    a.p.N_index = e->N_index; a.p.dd = nullptr; a.p.xc = Xchg{e->X, e->seg, e->nbs, e->cpb}; a.p.st = e->st;
    a.p.ld = e->ld; a.p.nN = e->nN; a.p.cpb = e->cpb; a.p.block0 = e->rank * e->nbs; a.p.eps = e->eps;
    a.p.vs_row = e->vs_row; a.p.vs_val = e->vs_val; a.pos_hint = e->pos_hint;
    a.u0 = e->u; a.u1 = e->u + e->ld; a.W0 = e->W; a.W1 = e->W2;
    a.d = e->d; a.lam = e->lam; a.bmin = e->bmin; a.bsec = e->bmin + e->m; a.bidx = e->bidx; a.binfo = e->binfo; a.dpos = e->dpos;
    a.A_N = e->A_N; a.A_B = e->A_B; a.aq_save = e->aq_save; a.c_B = e->c_B; a.c_N = e->c_N; a.x = e->x;
    a.lb = e->lb; a.ub = e->ub; a.kind = e->kindv; a.B_index = e->B_index; a.N_index = e->N_index; a.Nb = e->Nb;
    a.m = e->m; a.rpb = e->upd2_rows; a.use_pend = use_pend; a.ill_tol = e->ill_tol;
    a.guard_abs = (e->cert_large && !e->guard_off) ? e->guard_abs : 0.0;
    a.aq_cur = e->colshard ? e->aq_cur : nullptr; a.own0 = e->own0; a.own1 = e->own1;
    a.trace = Trace{e->trace_obj, e->trace_it, e->trace_len};
    int mine = e->nblocks - a.p.block0;
    if (mine > e->nbs) mine = e->nbs;
    if (mine < 0) mine = 0;
    const dim3 g(mine + P2_BOOK), b(256);
    const size_t lds = e->price2_lds;
    if (e->colshard) {  // the sharded instantiations (ellp_lagged.inc, SH)
        if (e->price_wave) {
            hipLaunchKernelGGL((k_price2_wave<true>), g, b, lds + sizeof(double) * (size_t)e->ld, e->stream, a);
            return;
        }
        if (e->price_nt) {
            switch (e->priceT) {
            case 1: hipLaunchKernelGGL((k_price2<1, true, true>), g, b, lds, e->stream, a); break;
            case 2: hipLaunchKernelGGL((k_price2<2, true, true>), g, b, lds, e->stream, a); break;
            case 4: hipLaunchKernelGGL((k_price2<4, true, true>), g, b, lds, e->stream, a); break;
            case 8: hipLaunchKernelGGL((k_price2<8, true, true>), g, b, lds, e->stream, a); break;
            default: hipLaunchKernelGGL((k_price2<16, true, true>), g, b, lds, e->stream, a); break;
            }
        } else {
            switch (e->priceT) {
            case 1: hipLaunchKernelGGL((k_price2<1, false, true>), g, b, lds, e->stream, a); break;
            case 2: hipLaunchKernelGGL((k_price2<2, false, true>), g, b, lds, e->stream, a); break;
            case 4: hipLaunchKernelGGL((k_price2<4, false, true>), g, b, lds, e->stream, a); break;
            case 8: hipLaunchKernelGGL((k_price2<8, false, true>), g, b, lds, e->stream, a); break;
            default: hipLaunchKernelGGL((k_price2<16, false, true>), g, b, lds, e->stream, a); break;
            }
        }
        return;
    }
    if (e->price_wave) {
        hipLaunchKernelGGL((k_price2_wave<false>), g, b, lds + sizeof(double) * (size_t)e->ld, e->stream, a);
        return;
    }
    if (e->price_nt) {
        switch (e->priceT) {
        case 1: hipLaunchKernelGGL((k_price2<1, true>), g, b, lds, e->stream, a); break;
        case 2: hipLaunchKernelGGL((k_price2<2, true>), g, b, lds, e->stream, a); break;
        case 4: hipLaunchKernelGGL((k_price2<4, true>), g, b, lds, e->stream, a); break;
        case 8: hipLaunchKernelGGL((k_price2<8, true>), g, b, lds, e->stream, a); break;
        default: hipLaunchKernelGGL((k_price2<16, true>), g, b, lds, e->stream, a); break;
        }
    } else {
        switch (e->priceT) {
        case 1: hipLaunchKernelGGL((k_price2<1, false>), g, b, lds, e->stream, a); break;
        case 2: hipLaunchKernelGGL((k_price2<2, false>), g, b, lds, e->stream, a); break;
        case 4: hipLaunchKernelGGL((k_price2<4, false>), g, b, lds, e->stream, a); break;
        case 8: hipLaunchKernelGGL((k_price2<8, false>), g, b, lds, e->stream, a); break;
        default: hipLaunchKernelGGL((k_price2<16, false>), g, b, lds, e->stream, a); break;
        }
    }
}

void launch_ftran_eta(ellp_engine *e) {
    FtranEtaArgs a{};
    a.W0 = e->W; a.W1 = e->W2; a.A_N = e->A_N; a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb};
    a.N_index = e->N_index; a.B_index = e->B_index; a.Nb = e->Nb; a.kind = e->kindv;
    a.x = e->x; a.lb = e->lb; a.ub = e->ub; a.d = e->d; a.lam = e->lam; a.bmin = e->bmin; a.bsec = e->bmin + e->m; a.bidx = e->bidx;
    a.binfo = e->binfo; a.dpos = e->dpos;
    a.A_B = e->A_B; a.c_B = e->c_B; a.aq_save = e->aq_save; a.st = e->st; a.m = e->m; a.ld = e->ld; a.nN = e->nN;
    a.nblocks = e->nblocks; a.cpb = e->cpb; a.rows_per_block = e->upd2_rows; a.eps = e->eps;
    if (e->colshard) {
        a.aq_cur = e->aq_cur; a.aq_out = e->aq_cur;
        if (e->sel_in_ftran) {
            a.sel_packs = e->packs; a.sel_world = e->world; a.mbox_commit = e->sel_commit ? 1 : 0;
        }
    }
    const dim3 g(e->upd2_blocks + 1), b(256);
    const int64_t nr = ((e->ld >> 1) + 255) / 256;  // double2 per thread per row
    if (e->colshard) {  // the sharded instantiations (selection in the prologue instead of the entering fold)
        if (nr <= 1) hipLaunchKernelGGL((k_ftran_eta<1, true>), g, b, e->ftran_lds, e->stream, a);
        else if (nr <= 2) hipLaunchKernelGGL((k_ftran_eta<2, true>), g, b, e->ftran_lds, e->stream, a);
        else if (nr <= 4) hipLaunchKernelGGL((k_ftran_eta<4, true>), g, b, e->ftran_lds, e->stream, a);
        else if (nr <= 8) hipLaunchKernelGGL((k_ftran_eta<8, true>), g, b, e->ftran_lds, e->stream, a);
        else hipLaunchKernelGGL((k_ftran_eta<0, true>), g, b, e->ftran_lds, e->stream, a);
        return;
    }
    if (nr <= 1) hipLaunchKernelGGL((k_ftran_eta<1>), g, b, e->ftran_lds, e->stream, a);
    else if (nr <= 2) hipLaunchKernelGGL((k_ftran_eta<2>), g, b, e->ftran_lds, e->stream, a);
    else if (nr <= 4) hipLaunchKernelGGL((k_ftran_eta<4>), g, b, e->ftran_lds, e->stream, a);
    else if (nr <= 8) hipLaunchKernelGGL((k_ftran_eta<8>), g, b, e->ftran_lds, e->stream, a);
    else hipLaunchKernelGGL((k_ftran_eta<0>), g, b, e->ftran_lds, e->stream, a);
}

// iteration k of the two-launch pipeline: P_k folds and books iteration k-1 (if one is open), prices k;
// F_k applies the eta of k-1 and forms d_k.  Iteration k itself stays open until the next P or the flush.
void launch_primal_iteration_lagged(ellp_engine *e) {
    const bool full_btran = !e->u_valid || e->since_btran >= (uint64_t)e->btran_refresh;
    if (full_btran) {
        // u = B^-T c_B from the materialised inverse: W and c_B both stand at "all pivots but the open one",
        // exactly the u_{k-1} that P_k advances by the open pivot's rank-1 term
        Prof p(e, ELLP_K_BTRAN);
        launch_btran(e);
        e->since_btran = 0;
        e->u_valid = true;
    }
    {
        Prof p(e, ELLP_K_PRICE);
        launch_price2(e, e->lag_open ? 1 : 0);
    }
    {
        Prof p(e, ELLP_K_FTRAN);
        launch_ftran_eta(e);
    }
    launch_drift_check(e);
    e->lag_open = true;
    e->since_btran += 1;
    e->since_refactor += 1;
    e->enqueued += 1;
}

// closing kernel of a slice: k_update2 folds the open iteration's ratio test, applies its eta update and
// does its bookkeeping (the three-launch path's third kernel, unchanged)
void launch_dual_close(ellp_engine *e);
void launch_flush(ellp_engine *e) {
    launch_dual_close(e);
    if (!e->lag_open) return;
    {
        Prof p(e, ELLP_K_UPDATE);
        launch_update2<0>(e, 1);
    }
    e->lag_open = false;
}

void launch_price_se(ellp_engine *e) {
    SeArgs a{};
    a.A_N = e->A_N; a.u = e->u; a.rho = e->se_rho; a.v = e->se_v; a.c_N = e->c_N; a.Nb = e->Nb; a.N_index = e->N_index;
    a.gamma = e->se_gamma;
    const bool unit_ok = !(e->opts.flags & ELLP_FLAG_DENSE_PRICING);
    a.vs_row = unit_ok ? e->vs_row : nullptr; a.vs_val = unit_ok ? e->vs_val : nullptr;
    a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb}; a.st = e->st; a.ld = e->ld; a.nN = e->nN; a.cpb = e->cpb; a.eps = e->eps;
    hipLaunchKernelGGL(k_price_se, dim3((unsigned)e->nblocks), dim3(256), 0, e->stream, a);
}

void launch_primal_iteration(ellp_engine *e) {
    if (e->lagged) {
        launch_primal_iteration_lagged(e);
        return;
    }
    const bool full_btran = (e->opts.btran_mode == 1) || !e->u_valid || e->since_btran >= (uint64_t)e->btran_refresh;
    if (full_btran) {
        Prof p(e, ELLP_K_BTRAN);
        launch_btran(e);
        e->since_btran = 0;
        e->u_valid = true;
    }
    {
        Prof p(e, ELLP_K_PRICE);
        if (e->se) launch_price_se(e);
        else launch_price<0>(e);
    }
    {
        Prof p(e, ELLP_K_FTRAN);
        launch_ftran2<0>(e);
    }
    if (e->se && !e->se_vpart) {  // v = B^-T d from the inverse this iteration's FTRAN used: the weight update of the NEXT pricing launch
        Prof p(e, ELLP_K_BTRAN);
        BtranArgs a{e->W, e->W2, e->d, e->upart, e->se_v, e->se_v, e->st, e->m, e->ld, e->btran_rows, e->btran_tiles};
        const int64_t half = e->ld >> 1;
        hipLaunchKernelGGL(k_btran_part, dim3((unsigned)((half + 255) / 256), (unsigned)e->btran_tiles), dim3(256), 0, e->stream, a);
        hipLaunchKernelGGL(k_btran_reduce, dim3((unsigned)((e->ld + 255) / 256)), dim3(256), 0, e->stream, a);
    }
    launch_drift_check(e);
    {
        Prof p(e, ELLP_K_UPDATE);
        launch_update2<0>(e, e->opts.btran_mode == 1 ? 0 : 1);
    }
    if (e->se && e->se_vpart) {  // v = the row blocks' partial sums of k_update2, added in block order
        Prof p(e, ELLP_K_BTRAN);
        hipLaunchKernelGGL(k_se_vreduce, dim3((unsigned)((e->ld + 15) / 16)), dim3(256), 0, e->stream, e->se_vpart, e->se_v, e->st, e->ld,
                           e->upd2_blocks);
    }
    e->since_btran += 1;
    e->since_refactor += 1;
    e->enqueued += 1;
}

// the closing block of the fused dual iteration in front (a no-op on the device if that has been closed already)
void launch_dual_close(ellp_engine *e) {
    if (!e->dual_open) return;
    Prof p(e, ELLP_K_DUPDATE);
    DualCloseArgs c{e->d, e->A_N, e->A_B, e->c_B, e->c_N, e->x, e->dd, e->B_index, e->N_index, e->Nb, e->binfo, e->bmin,
                    e->bmin + e->m, (int)((e->m + UPD_ROWS - 1) / UPD_ROWS), e->st, e->m, e->ld, e->ill_tol,
                    Trace{e->trace_obj, e->trace_it, e->trace_len}, e->dual_maxviol};
    hipLaunchKernelGGL(k_dual_close, dim3(1), dim3(256), 0, e->stream, c);
    e->dual_open = false;
}

// FTRAN + eta update + x_B / d_N / y updates in one pass (ellp_dualfu.inc); the iteration stays open
void launch_dual_fu(ellp_engine *e) {
    Prof p(e, ELLP_K_FTRAN);
    DualFuArgs a{};
    a.W0 = e->W; a.W1 = e->W2; a.A_N = e->A_N; a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb};
    a.N_index = e->N_index; a.B_index = e->B_index; a.x = e->x; a.d = e->d; a.y = e->y; a.dd = e->dd;
    a.lb = e->lb; a.ub = e->ub; a.kind = e->kindv; a.lrow = e->binfo; a.ldelta = e->bmin; a.lside = e->bmin + e->m;
    a.st = e->st; a.m = e->m; a.ld = e->ld; a.nN = e->nN; a.nblocks = e->nblocks; a.eps = e->eps;
    a.seq = e->dual_seq;
    a.maxviol = e->dual_maxviol;
    a.guard_abs = (e->cert_large && !e->guard_off) ? e->guard_abs : 0.0;
    const dim3 g((unsigned)((e->m + UPD_ROWS - 1) / UPD_ROWS) + DFU_BOOK), b(256);
    const int64_t nr = ((e->ld >> 1) + 255) / 256;  // double2 per thread per row
    if (nr <= 1) hipLaunchKernelGGL((k_dual_fu<1>), g, b, 0, e->stream, a);
    else if (nr <= 2) hipLaunchKernelGGL((k_dual_fu<2>), g, b, 0, e->stream, a);
    else if (nr <= 4) hipLaunchKernelGGL((k_dual_fu<4>), g, b, 0, e->stream, a);
    else hipLaunchKernelGGL((k_dual_fu<8>), g, b, 0, e->stream, a);
    e->dual_open = true;
}

void launch_dual_iteration(ellp_engine *e) {
    e->dual_seq += 1;
    {
        Prof p(e, ELLP_K_DPRICE);
        // closes the fused iteration in front of it if this engine folds the closing work (dual_fold) — unless it
        // returns at entry on a stop, so the host keeps dual_open until a k_dual_close has been enqueued (which does
        // nothing on the device when the iteration is closed already)
        launch_price<1>(e);
    }
    // the drift monitor compares A_B alpha_q with a_q between FTRAN and the update: those iterations keep the
    // three-launch form (both forms leave the same state behind)
    const bool drift_now = e->drift_every > 0 && e->since_drift + 1 >= (uint64_t)e->drift_every;
    if (e->dual_fused && !drift_now) {
        if (e->drift_every > 0) e->since_drift += 1;
        launch_dual_fu(e);
        if (!e->dual_fold) launch_dual_close(e);
        e->since_refactor += 1;
        e->enqueued += 1;
        return;
    }
    {
        Prof p(e, ELLP_K_FTRAN);
        launch_ftran2<1>(e);
    }
    launch_drift_check(e);
    {
        Prof p(e, ELLP_K_DUPDATE);
        launch_update2<1>(e, 0);
    }
    e->since_refactor += 1;
    e->enqueued += 1;
}

void launch_dleave(ellp_engine *e) {
    Prof p(e, ELLP_K_DLEAVE);
    DLeaveArgs a{e->x, e->lb, e->ub, e->kindv, e->B_index, e->st, e->m, e->eps, e->dual_maxviol};
    hipLaunchKernelGGL(k_dleave, dim3(1), dim3(256), 0, e->stream, a);
}

ellp_status status_message(const DevState &s, char *errbuf, size_t errlen) {
    switch (s.status) {
    case ELLP_ERR_SINGULAR: set_err(errbuf, errlen, "invalid B, A_B is not invertible"); break;
    case ELLP_ERR_NAN: set_err(errbuf, errlen, "NaN detected"); break;
    case ELLP_ERR_PANIC:
        if (s.panic_code == 402) set_err(errbuf, errlen, "assertion failed: lambda >= 0.");
        else if (s.panic_code == 229) set_err(errbuf, errlen, "pivot should have been unbounded");
        else if (s.panic_code == 293) set_err(errbuf, errlen, "assertion failed: reduced cost of a nonbasic variable has the wrong sign (dual phase 2 construction)");
        else if (s.panic_code >= 9100 && s.panic_code < 9200) set_err(errbuf, errlen, "debug build: an index left its range (check %d)", s.panic_code);
        else if (s.panic_code == 201) set_err(errbuf, errlen, "bounds should always be fixed or two-sided");
        else if (s.panic_code == 187) set_err(errbuf, errlen, "unwrap() on None in BTRAN");
        else if (s.panic_code == 295) set_err(errbuf, errlen, "unwrap() on None in FTRAN");
        else if (s.panic_code == 249) set_err(errbuf, errlen, "unwrap() on None in dual BTRAN");
        else if (s.panic_code == 294) set_err(errbuf, errlen, "unwrap() on None in dual FTRAN");
        else set_err(errbuf, errlen, "reference invariant violated (code %d)", s.panic_code);
        break;
    default: break;
    }
    return (ellp_status)s.status;
}

double host_dual_obj(int64_t m, int64_t n_c, const double *b, const uint8_t *kind, const double *lb,
                     const double *ub, const double *y, const double *d) {
    // standard_form.rs:52-68
    double obj = 0.0;
    for (int64_t i = 0; i < m; ++i) obj += b[i] * y[i];
    for (int64_t i = 0; i < n_c; ++i) {
        switch (kind[i]) {
        case ELLP_BOUND_LOWER: obj += lb[i] * d[i]; break;
        case ELLP_BOUND_UPPER: obj += ub[i] * d[i]; break;
        case ELLP_BOUND_TWOSIDED: obj += (d[i] > 0.0) ? lb[i] * d[i] : ub[i] * d[i]; break;
        case ELLP_BOUND_FIXED: obj += lb[i] * d[i]; break;
        default: break;
        }
    }
    return obj;
}

// ellp_stats from the last read-back; for a primal engine obj = c.x is computed now (one small launch)
void fill_stats(ellp_engine *e, ellp_stats *stats) {
    if (!stats) return;
    memset(stats, 0, sizeof(*stats));
    stats->iters = e->h_st->iters;
    stats->pivots = e->h_st->pivots;
    stats->bound_flips = e->h_st->flips;
    stats->refactors = e->refactors + e->refreshes;
    stats->obj = e->h_st->obj;
    if (e->obj_fresh) {
        e->obj_fresh = false;  // ellp_engine_run computed c . x in front of the read-back this state came from
    } else if (e->kind == ELLP_ENGINE_PRIMAL && e->st) {
        hipLaunchKernelGGL(k_primal_obj, dim3(1), dim3(1024), 0, e->stream, e->c_B, e->c_N, e->x, e->B_index,
                           e->N_index, e->m, e->nN, e->st);
        double v = 0.0;
        if (hipMemcpyAsync(&v, &e->st->obj, sizeof(double), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
            hipStreamSynchronize(e->stream) == hipSuccess)
            stats->obj = v;
    }
    stats->t_setup_s = e->t_setup;
    for (int k = 0; k < ELLP_K_COUNT; ++k) {
        stats->kernel_ms[k] = e->kernel_ms[k];
        stats->kernel_calls[k] = e->kernel_calls[k];
    }
    if (e->opts.profile && e->ev_overhead_ms >= 0.0) {
        stats->kernel_ms[ELLP_K_EVENT_COST] = e->ev_overhead_ms;
        stats->kernel_calls[ELLP_K_EVENT_COST] = 1;
    }
}

}  // namespace

extern "C" {

void ellp_default_opts(ellp_opts *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->max_iter = 1000;  // Default::default(), primal…:21 / dual…:22
    o->eps = 1e-10;      // util.rs:1
    o->device = -1;
}

#ifdef ELLP_BL_STAMPS
int ellp_debug_bl_stamps(unsigned long long *out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bl_stamps), 8 * sizeof(unsigned long long));
}
#endif
int ellp_hip_abi_version(void) { return ELLP_HIP_ABI_VERSION; }

int ellp_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Streams and pinned status buffers outlive engines: hipStreamCreate + hipStreamDestroy + hipHostMalloc/Free were
// 3 of the 4.8 ms of an AFIRO solve (rocprofv3 --hip-trace: ~1.4 ms per stream call).  An engine takes a set for
// its device from this process-wide pool and gives it back, drained, when it is destroyed.
struct HostPool {
    std::mutex mu;
    std::vector<HostSet> free_sets;
};
static HostPool &host_pool() {
    static HostPool *p = new HostPool;  // never destroyed: the HIP runtime may be gone before static destructors run
    return *p;
}
static hipError_t host_set_acquire(int device, HostSet *out) {
    {
        HostPool &hp = host_pool();
        std::lock_guard<std::mutex> g(hp.mu);
        for (size_t k = 0; k < hp.free_sets.size(); ++k)
            if (hp.free_sets[k].device == device) {
                *out = hp.free_sets[k];
                hp.free_sets.erase(hp.free_sets.begin() + (long)k);
                return hipSuccess;
            }
    }
    HostSet hs;
    hs.device = device;
    hipError_t rc = hipStreamCreateWithFlags(&hs.stream, hipStreamNonBlocking);
    if (rc != hipSuccess) return rc;
    rc = hipHostMalloc(reinterpret_cast<void **>(&hs.h_st), sizeof(DevState), hipHostMallocDefault);
    if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void **>(&hs.h_look), 2 * sizeof(DevState), hipHostMallocDefault);
    if (rc != hipSuccess) {
        if (hs.h_st) (void)hipHostFree(hs.h_st);
        (void)hipStreamDestroy(hs.stream);
        return rc;
    }
    *out = hs;
    return hipSuccess;
}
static void host_set_release(const HostSet &hs) {
    if (!hs.stream) return;
    HostPool &hp = host_pool();
    std::lock_guard<std::mutex> g(hp.mu);
    if (hp.free_sets.size() < 16) {
        hp.free_sets.push_back(hs);
        return;
    }
    (void)hipHostFree(hs.h_st);
    (void)hipHostFree(hs.h_look);
    (void)hipStreamDestroy(hs.stream);
}

void ellp_engine_destroy(ellp_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->luw_ready) {
        ellp_lu_rows_free(&e->luw);
        e->luw_ready = false;
    }
#ifdef ELLP_DBG_STAMPS
    if (e->h_st && hipMemcpy(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost) == hipSuccess) {
        long long t0 = e->h_st->dbg[0][0][0] ? e->h_st->dbg[0][0][0] : e->h_st->dbg[1][0][0];
        for (int k = 0; k < 3; ++k)
            for (int b = 0; b < 4; ++b) {
                fprintf(stderr, "stamps kernel %d blocksel %d:", k, b);
                for (int sl = 0; sl < 8; ++sl) fprintf(stderr, " %8.2f", e->h_st->dbg[k][b][sl] ? (double)(e->h_st->dbg[k][b][sl] - t0) / 100.0 : -1.0);
                fprintf(stderr, "\n");
            }
    }
#endif
    if (e->small_stamps) {
        unsigned long long t[32] = {0};
        if (hipMemcpy(t, e->small_stamps, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess && t[8] > 0) {
            static const char *nm_s[8] = {"leaving row", "copy A_B", "LU", "BTRAN", "pricing", "entering fold", "FTRAN",
                                          "ratio test + updates"};
            static const char *nm_m[8] = {"leaving row", "LU", "BTRAN U^T", "BTRAN L^T", "pricing", "entering fold + FTRAN",
                                          "ratio test + updates", "-"};
            const char **nm = e->mid_lds > 0 && e->small_lds == 0 ? nm_m : nm_s;
            fprintf(stderr, "%s m=%lld nN=%lld threads=%d, us per iteration over %llu iterations:", nm == nm_m ? "k_mid" : "k_small",
                    (long long)e->m, (long long)e->nN, nm == nm_m ? e->mid_nt : e->small_nt, t[8]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.2f;", nm[k], (double)t[k] / 100.0 / (double)t[8]);
            double tot = 0.0;
            for (int k = 0; k < 8; ++k) tot += (double)t[k];
            fprintf(stderr, " total %.2f; shader clock %.0f MHz\n", tot / 100.0 / (double)t[8], (double)t[9] / (tot / 100.0));
            if (nm == nm_m)
                fprintf(stderr, "   LU: panel load %.2f; 16 steps %.2f; store + publish %.2f; pivot rows of the trailing columns %.2f; trailing update %.2f\n",
                        (double)t[10] / 100.0 / (double)t[8], (double)t[11] / 100.0 / (double)t[8], (double)t[12] / 100.0 / (double)t[8],
                        (double)t[13] / 100.0 / (double)t[8], (double)t[14] / 100.0 / (double)t[8]);
        }
    }
    e->stream = e->own_stream;
    if (e->comm && e->rccl) (void)e->rccl->CommDestroy(e->comm);
    for (void *p : e->ipc_opened) (void)hipIpcCloseMemHandle(p);
    if (e->mbox) (void)hipFree(e->mbox);
    if (e->mflags) (void)hipFree(e->mflags);
    for (void *p : e->allocs) (void)hipFree(p);
    for (auto ev : e->look_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    host_set_release(e->host_set);
    delete e;
}

// the instantiation of k_mid for an engine kind and a workgroup size
static const void *mid_kernel(int kind, int nt) {
    if (kind == ELLP_ENGINE_PRIMAL) {
        if (nt == 256) return reinterpret_cast<const void *>(&k_mid<0, 256>);
        if (nt == 512) return reinterpret_cast<const void *>(&k_mid<0, 512>);
        return reinterpret_cast<const void *>(&k_mid<0, 1024>);
    }
    if (nt == 256) return reinterpret_cast<const void *>(&k_mid<1, 256>);
    if (nt == 512) return reinterpret_cast<const void *>(&k_mid<1, 512>);
    return reinterpret_cast<const void *>(&k_mid<1, 1024>);
}

// the instantiation of k_small for an engine kind and a workgroup size
static const void *small_kernel(int kind, int nt) {
    if (kind == ELLP_ENGINE_PRIMAL) {
        if (nt == 64) return reinterpret_cast<const void *>(&k_small<0, 64>);
        if (nt == 128) return reinterpret_cast<const void *>(&k_small<0, 128>);
        return reinterpret_cast<const void *>(&k_small<0, 256>);
    }
    if (nt == 64) return reinterpret_cast<const void *>(&k_small<1, 64>);
    if (nt == 128) return reinterpret_cast<const void *>(&k_small<1, 128>);
    return reinterpret_cast<const void *>(&k_small<1, 256>);
}

// n_explicit: columns of A the caller passes; columns n_explicit .. n-1 (m of them, or none) are the
// artificial columns of primal phase 1 and are made on the device (build_phase1)
static ellp_status engine_create_impl(int kind, int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                               const double *b, const uint8_t *bound_kind, const double *lb, const double *ub,
                               const double *x, const int64_t *B_index, int64_t n_B, const int64_t *N_index,
                               const uint8_t *N_bound, int64_t n_N, const double *y, const double *d,
                               const ellp_opts *opts_in, ellp_engine **out, char *errbuf, size_t errlen,
                               int64_t n_explicit, bool build_phase1) {
    if (!out) return ELLP_ERR_ARG;
    *out = nullptr;
    if (errbuf && errlen) errbuf[0] = 0;
    auto t0 = std::chrono::steady_clock::now();
    if (kind != ELLP_ENGINE_PRIMAL && kind != ELLP_ENGINE_DUAL) {
        set_err(errbuf, errlen, "unknown engine kind %d", kind);
        return ELLP_ERR_ARG;
    }
    if (m <= 0) {
        set_err(errbuf, errlen, "m == 0: the trivial problem is solved on the host (primal…:118-122)");
        return ELLP_ERR_ARG;
    }
    if (n < 0 || n_c < n || !A || !c || !b || !bound_kind || !lb || !ub || !x || !B_index || (n_N > 0 && (!N_index || !N_bound))) {
        set_err(errbuf, errlen, "null pointer or inconsistent sizes");
        return ELLP_ERR_ARG;
    }
    if (kind == ELLP_ENGINE_DUAL && (!y || !d)) {
        set_err(errbuf, errlen, "dual engine needs y and d");
        return ELLP_ERR_ARG;
    }
    if (n_B != m) {  // primal…:124-130
        set_err(errbuf, errlen, "invalid B, has %lld elements but %lld expected", (long long)n_B, (long long)m);
        return ELLP_ERR_BAD_DIMS;
    }
    if (n < m) {
        set_err(errbuf, errlen, "cols < rows");
        return ELLP_ERR_PANIC;
    }
    if (n_N != n - m) {  // primal…:132-140
        set_err(errbuf, errlen, "invalid N, has %lld elements but %lld expected", (long long)n_N, (long long)(n - m));
        return ELLP_ERR_BAD_DIMS;
    }
    if (n_c >= (int64_t)1 << 31) {
        set_err(errbuf, errlen, "n_c too large");
        return ELLP_ERR_ARG;
    }
    for (int64_t i = 0; i < m; ++i)
        if (B_index[i] < 0 || B_index[i] >= n) {
            set_err(errbuf, errlen, "B_index[%lld] out of range", (long long)i);
            return ELLP_ERR_ARG;
        }
    for (int64_t j = 0; j < n_N; ++j)
        if (N_index[j] < 0 || N_index[j] >= n || N_bound[j] > 2) {
            set_err(errbuf, errlen, "N_index/N_bound[%lld] out of range", (long long)j);
            return ELLP_ERR_ARG;
        }
    for (int64_t i = 0; i < n_c; ++i)
        if (bound_kind[i] > 4) {
            set_err(errbuf, errlen, "bound_kind[%lld] out of range", (long long)i);
            return ELLP_ERR_ARG;
        }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_err(errbuf, errlen, "no HIP device available (this library has no CPU path)");
        return ELLP_ERR_DEVICE;
    }
    ellp_engine *e = new ellp_engine();
    ellp_opts defaults;
    ellp_default_opts(&defaults);
    e->opts = opts_in ? *opts_in : defaults;
    e->eps = e->opts.eps > 0.0 ? e->opts.eps : 1e-10;
    e->kind = kind;
    e->m = m;
    e->n = n;
    e->n_c = n_c;
    e->nN = n_N;
    e->ld = round_up(m, 16);
    int dev = e->opts.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    e->device = dev;

#define ECHK(expr)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            set_err(errbuf, errlen, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__,   \
                    __LINE__, #expr);                                                                \
            ellp_engine_destroy(e);                                                                  \
            return ELLP_ERR_DEVICE;                                                                  \
        }                                                                                            \
    } while (0)

    ECHK(hipSetDevice(dev));
    ECHK(host_set_acquire(dev, &e->host_set));
    e->stream = e->host_set.stream;
    e->own_stream = e->stream;
    e->h_st = e->host_set.h_st;
    e->h_look = e->host_set.h_look;
    const int64_t ld = e->ld;
    const int64_t nNa = n_N > 0 ? n_N : 1;
    // geometry
    {
        // A_N far beyond the Infinity Cache -> stream it with nt loads from ~2048 blocks; otherwise
        // ~1024 blocks (the entering fold stages one maximum per block).  tools/price_bench.hip
        e->price_nt = 8.0 * (double)ld * (double)nNa > 160e6;
        int64_t cpb = e->price_nt ? (n_N + 2047) / 2048 : (n_N + 1023) / 1024;
        if (cpb < 1) cpb = 1;
        if (cpb > 64) cpb = 64;
        // wave-per-column pricing: cache-resident A_N, rows that fill a wave, and >= 5 columns per
        // block so that all four waves of a block have a column pair (measured at C3/C4: 18.9 -> 18.0
        // and 15.8 -> 14.3 us; the number of columns per block between 5 and 16 makes no difference)
        e->price_wave = !e->price_nt && ld >= 512 && cpb >= 5;
        e->cpb = (int)cpb;
        e->nblocks = (int)((nNa + cpb - 1) / cpb);
        const int64_t need = ((ld >> 1) + 255) / 256;
        e->priceT = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
        if (need > 16) {
            set_err(errbuf, errlen, "m = %lld exceeds this build's pricing tile (m <= 8192)", (long long)m);
            ellp_engine_destroy(e);
            return ELLP_ERR_ARG;
        }
        e->upd_rows = m >= 1024 ? 4 : (m >= 256 ? 2 : 1);
        e->upd_blocks = (int)((m + e->upd_rows - 1) / e->upd_rows);
        // 8 rows per block once m is large: half as many blocks re-stage the fold inputs (13 m bytes
        // each) and re-read the pivot row; measured -9 us of 58 at m=4000, +0.4 us at m=2000
        e->upd2_rows = m >= 3072 ? 8 : e->upd_rows;
        e->upd2_blocks = (int)((m + e->upd2_rows - 1) / e->upd2_rows);
        int64_t fw = m < 2048 ? m : 2048;
        e->ftran_blocks = (int)((fw + 3) / 4);
        e->btran_rows = (int)((m + 63) / 64);
        if (e->btran_rows < 8) e->btran_rows = 8;
        e->btran_tiles = (int)((m + e->btran_rows - 1) / e->btran_rows);
        const int nchunks = (int)((m + 63) >> 6);
        const size_t staged = sizeof(double) * (size_t)nchunks + (size_t)m * (8 + 4 + 1) + 16;
        if (staged <= 40 * 1024) {  // keep >= 4 update blocks per CU resident
            e->upd_stage = 1;
            e->upd_lds = staged;
        } else {
            e->upd_stage = 0;
            e->upd_lds = sizeof(double) * (size_t)nchunks + 16;
        }
        e->ftran_lds = sizeof(double) * (size_t)e->nblocks + 16;
        e->refactor_period = e->opts.refactor_period > 0 ? e->opts.refactor_period : 0;
        // reactive maintenance after a tiny pivot (|alpha_r| < ill_tol * max|alpha|): small LPs, and steepest edge at any
        // size — its longer steps meet such pivots often enough that at config 5 a variable once left the basis 0.03
        // off its bound between two looks of the drift monitor; costs the look-ahead of the run loop
        const bool se_wanted = kind == ELLP_ENGINE_PRIMAL && (e->opts.flags & ELLP_FLAG_PRIMAL_STEEPEST_EDGE);
        e->ill_tol = (m <= 512 || se_wanted) ? 1e-3 : 0.0;
        if (const char *it = getenv("ELLP_ILL_TOL")) e->ill_tol = atof(it);  // diagnostics
        // drift monitor: only where refreshes are rare (period > 64), four checks per period; one GEMV
        // over A_B + a one-block fold, ~70 us at config 3 (0.7 % of the loop)
        {
            const int64_t p = e->refactor_period > 0 ? e->refactor_period : default_period(e);
            e->drift_every = p > 64 ? (int)(p / 4 > 64 ? p / 4 : 64) : 0;
        }
        e->drift_tol = 1e-10;  // a backstop: config 3 drifts to 1.4e-11 in 3000 updates without any refresh
        if (const char *v = getenv("ELLP_DRIFT_EVERY"); v && v[0]) e->drift_every = atoi(v);  // diagnostics
        if (const char *v = getenv("ELLP_DRIFT_TOL"); v && v[0]) e->drift_tol = atof(v);        // diagnostics
    }

    ECHK(dmalloc(e, &e->A_B, (size_t)(ld * m)));
    ECHK(dmalloc(e, &e->A_N, (size_t)(ld * nNa)));
    ECHK(dmalloc(e, &e->W, (size_t)(m * ld)));
    ECHK(dmalloc(e, &e->W2, (size_t)(m * ld)));
    ECHK(dmalloc(e, &e->c_B, (size_t)m));
    ECHK(dmalloc(e, &e->c_N, (size_t)nNa));
    ECHK(dmalloc(e, &e->u, (size_t)(2 * ld)));  // two buffers (DevState::usel): the two-launch pipeline reads one, writes the other
    ECHK(dmalloc(e, &e->aq_save, (size_t)ld));
    ECHK(dmalloc(e, &e->bmin, (size_t)(2 * m)));  // smallest | second smallest per row block
    ECHK(dmalloc(e, &e->binfo, (size_t)m));
    e->trace_len = e->opts.trace_len > 0 ? e->opts.trace_len : 0;
    if (e->trace_len > 0) {
        ECHK(dmalloc(e, &e->trace_obj, (size_t)e->trace_len));
        ECHK(dmalloc(e, &e->trace_it, (size_t)e->trace_len));
        ECHK(hipMemsetAsync(e->trace_it, 0, sizeof(unsigned long long) * (size_t)e->trace_len, e->stream));
    }
    if (m <= 8192) {  // blocked rebuild (k_bl_factor keeps m / 1024 rows per thread)
        const int64_t nrb64 = (m + 63) / 64;
        int sp = (int)((256 + nrb64 - 1) / nrb64);
        e->bl_splits = sp < 1 ? 1 : (sp > 8 ? 8 : sp);
        ECHK(dmalloc(e, &e->bl_Cpart, (size_t)(e->bl_splits * m * BL_NB)));
        ECHK(dmalloc(e, &e->bl_C, (size_t)(2 * m * BL_NB)));
        ECHK(dmalloc(e, &e->bl_V, (size_t)(2 * m * BL_NB)));
        ECHK(dmalloc(e, &e->bl_Vs, (size_t)(m * 16)));
        ECHK(dmalloc(e, &e->bl_Wp, (size_t)(BL_NB * ld)));
        ECHK(dmalloc(e, &e->bl_Pm, (size_t)BL_NB));
        ECHK(dmalloc(e, &e->bl_nzval, (size_t)m));
        ECHK(dmalloc(e, &e->bl_nzrow, (size_t)m));
        ECHK(dmalloc(e, &e->bl_nzcnt, (size_t)m));
    }
    e->nbs = e->nblocks;
    e->seg = 2 * (int64_t)e->nbs + 2 * (int64_t)e->nbs * e->cpb;
    ECHK(dmalloc(e, &e->X, (size_t)e->seg));
    ECHK(dmalloc(e, &e->x, (size_t)n_c));
    ECHK(dmalloc(e, &e->lb, (size_t)n_c));
    ECHK(dmalloc(e, &e->ub, (size_t)n_c));
    ECHK(dmalloc(e, &e->kindv, (size_t)n_c));
    ECHK(dmalloc(e, &e->d, (size_t)ld));
    ECHK(dmalloc(e, &e->b_dev, (size_t)ld));
    ECHK(dmalloc(e, &e->tvec, (size_t)ld));
    ECHK(dmalloc(e, &e->cand, (size_t)ld));
    ECHK(dmalloc(e, &e->maxbits, (size_t)2));
    ECHK(dmalloc(e, &e->xg, (size_t)nNa));
    ECHK(hipMemsetAsync(e->b_dev, 0, sizeof(double) * (size_t)ld, e->stream));
    ECHK(dmalloc(e, &e->upart, (size_t)(e->btran_tiles * ld)));
    ECHK(dmalloc(e, &e->B_index, (size_t)m));
    ECHK(dmalloc(e, &e->N_index, (size_t)nNa));
    ECHK(dmalloc(e, &e->Nb, (size_t)nNa));
    ECHK(dmalloc(e, &e->perm, (size_t)m));
    ECHK(dmalloc(e, &e->used, (size_t)m));
    ECHK(dmalloc(e, &e->lam, (size_t)m));
    ECHK(dmalloc(e, &e->bidx, (size_t)m));
    ECHK(dmalloc(e, &e->dpos, (size_t)m));
    ECHK(dmalloc(e, &e->resid, (size_t)(m > 4096 ? m : 4096)));
    ECHK(hipMemsetAsync(e->resid, 0, sizeof(double) * (size_t)(m > 4096 ? m : 4096), e->stream));
    ECHK(dmalloc(e, &e->T, (size_t)(m * ld)));
    ECHK(hipMemsetAsync(e->T, 0, sizeof(double) * (size_t)(m * ld), e->stream));
    ECHK(dmalloc(e, &e->st, 1));
    if (kind == ELLP_ENGINE_DUAL) {
        ECHK(dmalloc(e, &e->y, (size_t)ld));
        ECHK(dmalloc(e, &e->dd, (size_t)n_c));
    }
    ECHK(hipMemsetAsync(e->u, 0, sizeof(double) * (size_t)(2 * ld), e->stream));
    ECHK(hipMemsetAsync(e->aq_save, 0, sizeof(double) * (size_t)ld, e->stream));
    ECHK(hipMemsetAsync(e->d, 0, sizeof(double) * (size_t)ld, e->stream));

    // upload (A goes through a temporary full copy, then columns are gathered on the device)
    {
        double *A_full = nullptr, *c_full = nullptr;
        ECHK(hipMalloc(reinterpret_cast<void **>(&A_full), sizeof(double) * (size_t)(m * n > 0 ? m * n : 1)));
        hipError_t rc = hipMalloc(reinterpret_cast<void **>(&c_full), sizeof(double) * (size_t)n_c);
        if (rc != hipSuccess) {
            (void)hipFree(A_full);
            ECHK(rc);
        }
        auto fail = [&](hipError_t err) {
            (void)hipStreamSynchronize(e->stream);
            (void)hipFree(A_full);
            (void)hipFree(c_full);
            return err;
        };
#define UCHK(expr)                              \
    do {                                        \
        hipError_t _u = (expr);                 \
        if (_u != hipSuccess) { ECHK(fail(_u)); } \
    } while (0)
        UCHK(hipMemcpyAsync(A_full, A, sizeof(double) * (size_t)(m * n_explicit), hipMemcpyHostToDevice, e->stream));
        if (n_explicit < n)  // the artificial columns: zero here, their +-1 is set once b~ is known (k_phase1_art)
            UCHK(hipMemsetAsync(A_full + m * n_explicit, 0, sizeof(double) * (size_t)(m * (n - n_explicit)), e->stream));
        UCHK(hipMemcpyAsync(c_full, c, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
        UCHK(hipMemcpyAsync(e->B_index, B_index, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, e->stream));
        if (n_N > 0) {
            UCHK(hipMemcpyAsync(e->N_index, N_index, sizeof(int64_t) * (size_t)n_N, hipMemcpyHostToDevice, e->stream));
            UCHK(hipMemcpyAsync(e->Nb, N_bound, (size_t)n_N, hipMemcpyHostToDevice, e->stream));
        }
        UCHK(hipMemcpyAsync(e->x, x, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
        UCHK(hipMemcpyAsync(e->b_dev, b, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, e->stream));
        UCHK(hipMemcpyAsync(e->lb, lb, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
        UCHK(hipMemcpyAsync(e->ub, ub, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
        UCHK(hipMemcpyAsync(e->kindv, bound_kind, (size_t)n_c, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)m), dim3(256), 0, e->stream, A_full, m, e->B_index, e->A_B, ld);
        hipLaunchKernelGGL(k_gather_vec, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, e->stream, c_full,
                           e->B_index, e->c_B, m);
        if (n_N > 0) {
            hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)n_N), dim3(256), 0, e->stream, A_full, m, e->N_index,
                               e->A_N, ld);
            hipLaunchKernelGGL(k_gather_vec, dim3((unsigned)((n_N + 255) / 256)), dim3(256), 0, e->stream, c_full,
                               e->N_index, e->c_N, n_N);
        }
        UCHK(hipGetLastError());
        DevState init;
        memset(&init, 0, sizeof(init));
        init.status = ST_RUNNING;
        init.r = -1;
        init.lr = -1;
        init.pp_hi = n_N;
        init.dp_seq = ~0ull;
        init.dp_applied = ~0ull;
        if (kind == ELLP_ENGINE_PRIMAL && e->opts.partial_segments > 1 && n_N > 0) {
            // partial pricing (f4): segments of ceil(|N| / P) positions, never more segments than positions
            int P = e->opts.partial_segments;
            if ((int64_t)P > n_N) P = (int)n_N;
            const int64_t S = (n_N + P - 1) / P;
            P = (int)((n_N + S - 1) / S);
            if (P > 1) {
                e->pp_P = P;
                e->pp_S = S;
                init.pp_P = P;
                init.pp_S = S;
                init.pp_hi = S;
            }
        }
        if (kind == ELLP_ENGINE_DUAL) {
            UCHK(hipMemsetAsync(e->y, 0, sizeof(double) * (size_t)ld, e->stream));
            UCHK(hipMemcpyAsync(e->y, y, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, e->stream));
            UCHK(hipMemcpyAsync(e->dd, d, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
            init.obj = host_dual_obj(m, n_c, b, bound_kind, lb, ub, y, d);  // dual…:184
        } else {
            double o = 0.0;  // standard_form.rs:48 `c.dot(x)`
            for (int64_t i = 0; i < n_c; ++i) o += c[i] * x[i];
            init.obj = o;
        }
        *e->h_st = init;
        UCHK(hipMemcpyAsync(e->st, e->h_st, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
        UCHK(hipStreamSynchronize(e->stream));
        (void)hipFree(A_full);
        (void)hipFree(c_full);
#undef UCHK
    }
    if (build_phase1) {
        // primal_problem.rs:236-246 on the device: b~ = b - A v over the nonbasic (= all structural) columns,
        // artificial column i = signum(b~_i) e_i with value |b~_i| (the kernels of launch_resync form b - A_N x_N)
        ResyncArgs ra{e->A_N, e->W, e->W2, e->b_dev, e->x, e->xg, e->tvec, e->upart, e->cand, e->maxbits, e->B_index,
                      e->N_index, e->st, e->m, e->ld, e->nN, 0, e->btran_tiles, 0};
        ra.cols_per_tile = (int)((e->nN + e->btran_tiles - 1) / e->btran_tiles);
        const int64_t half = e->ld >> 1;
        hipLaunchKernelGGL(k_resync_gather, dim3((unsigned)((e->nN + 255) / 256)), dim3(256), 0, e->stream, ra);
        hipLaunchKernelGGL(k_resync_part, dim3((unsigned)((half + 255) / 256), (unsigned)e->btran_tiles), dim3(256), 0, e->stream, ra);
        hipLaunchKernelGGL(k_resync_rhs, dim3((unsigned)((e->ld + 255) / 256)), dim3(256), 0, e->stream, ra);
        hipLaunchKernelGGL(k_phase1_art, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, e->stream, e->tvec, e->A_B, e->x,
                           e->B_index, m, e->ld);
        // the carried objective (and the trace) start from c . x WITH the artificials at |b~_i|, not from the host's value
        hipLaunchKernelGGL(k_primal_obj, dim3(1), dim3(1024), 0, e->stream, e->c_B, e->c_N, e->x, e->B_index, e->N_index, e->m,
                           e->nN, e->st);
    }
    // dual: initial dual feasibility assertion (dual…:139-151) — host side, data is in hand
    if (kind == ELLP_ENGINE_DUAL) {
        for (int64_t j = 0; j < n_N; ++j) {
            const double di = d[N_index[j]];
            bool infeasible;
            if (N_bound[j] == ELLP_NB_LOWER) infeasible = di < -e->eps;
            else if (N_bound[j] == ELLP_NB_UPPER) infeasible = di > e->eps;
            else infeasible = std::fabs(di) > e->eps;
            if (infeasible) {
                set_err(errbuf, errlen, "initial point of dual phase 2 is dual infeasible");
                ellp_engine_destroy(e);
                return ELLP_ERR_PANIC;
            }
        }
    }
    e->dual_maxviol = (kind == ELLP_ENGINE_DUAL && (e->opts.flags & ELLP_FLAG_DUAL_MAX_VIOLATION)) ? 1 : 0;
    e->dual_bflip = (kind == ELLP_ENGINE_DUAL && (e->opts.flags & ELLP_FLAG_DUAL_BOUND_FLIPPING)) ? 1 : 0;
    if (e->dual_bflip && (e->opts.pipeline == 1 || e->opts.pipeline == 2 || e->pp_P > 1)) {
        set_err(errbuf, errlen, "ELLP_FLAG_DUAL_BOUND_FLIPPING runs on the LU-per-iteration kernels (pipeline 0 or 3, up to 1,024 rows), "
                                "not on the explicit-inverse pipelines 1 / 2");
        ellp_engine_destroy(e);
        return ELLP_ERR_ARG;
    }
    if (kind == ELLP_ENGINE_DUAL) {
        bool box = true;
        for (int64_t i = 0; i < n_c && box; ++i) box = bound_kind[i] == ELLP_BOUND_TWOSIDED || bound_kind[i] == ELLP_BOUND_FIXED;
        for (int64_t i = 0; i < m && box; ++i) box = b[i] == 0.0;
        e->box_problem = box;
    }
    if (kind == ELLP_ENGINE_PRIMAL && (e->opts.flags & ELLP_FLAG_PRIMAL_STEEPEST_EDGE) &&
        (e->pp_P > 1 || e->opts.pipeline == 2 || e->opts.pipeline == 3 || e->opts.btran_mode == 1)) {
        // the caller must be able to tell which rule ran: steepest edge exists on the three-launch explicit-inverse engine only
        set_err(errbuf, errlen, "ELLP_FLAG_PRIMAL_STEEPEST_EDGE runs on the three-launch pipeline (pipeline 0 or 1) with full "
                                "pricing and the incremental BTRAN: not with partial_segments > 1, pipeline 2 / 3 or btran_mode 1");
        ellp_engine_destroy(e);
        return ELLP_ERR_ARG;
    }
    e->se = kind == ELLP_ENGINE_PRIMAL && (e->opts.flags & ELLP_FLAG_PRIMAL_STEEPEST_EDGE) && n_N > 0 && e->pp_P <= 1;
    // unit columns of the matrix (slacks, artificials, any other column with a single nonzero): the table the pricing
    // kernels of both loops consult (PriceArgs::vs_row).  ellp_opts.flags bit 0 or ELLP_NO_UNIT_COLUMNS=1: off.
    if (n_N > 0 && (e->se || (!(e->opts.flags & ELLP_FLAG_DENSE_PRICING) && getenv("ELLP_NO_UNIT_COLUMNS") == nullptr))) {
        if (dmalloc(e, &e->vs_row, (size_t)n_c) == hipSuccess && dmalloc(e, &e->vs_val, (size_t)n_c) == hipSuccess &&
            dmalloc(e, &e->pos_hint, (size_t)n_N + 64) == hipSuccess) {
            ECHK(hipMemsetAsync(e->vs_row, 0xff, sizeof(int32_t) * (size_t)n_c, e->stream));
            ECHK(hipMemsetAsync(e->pos_hint, 0, (size_t)n_N + 64, e->stream));
            hipLaunchKernelGGL(k_scan_singletons, dim3((unsigned)((n_N + 3) / 4)), dim3(256), 0, e->stream, e->A_N, e->ld, m, n_N,
                               e->N_index, e->vs_row, e->vs_val);
            hipLaunchKernelGGL(k_scan_singletons, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, e->stream, e->A_B, e->ld, m, m,
                               e->B_index, e->vs_row, e->vs_val);
        } else {
            (void)hipGetLastError();
            e->vs_row = nullptr;
            e->vs_val = nullptr;
        }
    }
    // small LPs: the reference's own loop in one persistent workgroup (ellp_small.inc) unless the caller
    // asked for the explicit-inverse engine (a maintenance period, a launch structure, profiling)
    {
        const int pl = e->opts.pipeline;
        e->small_lds = small_lds_bytes(m, n_N);
        e->mid_lds = mid_lds_bytes(m, n_N);
        // up to which size the exact loop ALONE is the default: m <= 128 (k_small, factors in LDS, 1-5 x slower per iteration
        // than the explicit-inverse engine).  Above that the default is the certified hybrid (below); pipeline = 3 or
        // ELLP_MID_AUTO_MAX still select the LU-per-iteration kernel k_mid for whole solves up to 1024 rows (15-40 x slower).
        int64_t mid_auto = SMALL_MAX_M;
        if (const char *ev = getenv("ELLP_MID_AUTO_MAX")) mid_auto = atoll(ev);
        const bool fits = e->small_lds > 0 || e->mid_lds > 0;
        const bool wanted = e->pp_P <= 1 && !e->se && (pl == 3 || e->dual_bflip || (pl == 0 && e->opts.refactor_period <= 0 && e->opts.btran_mode == 0 &&
                                                         e->opts.profile == 0 && (m <= SMALL_MAX_M || m <= mid_auto)));
        e->small = wanted && fits;
        e->mid = e->small && e->small_lds == 0;
        if (e->small && getenv("ELLP_SMALL_STAMPS") && !e->small_stamps) {
            if (dmalloc(e, &e->small_stamps, 32) == hipSuccess) (void)hipMemsetAsync(e->small_stamps, 0, 256, e->stream);
            else e->small_stamps = nullptr;
        }
        if (e->mid) {
            e->mid_nt = mid_threads(m);
            e->ldn = (n_N + 15) / 16 * 16;
            const void *fn = mid_kernel(e->kind, e->mid_nt);
            hipError_t ra = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->mid_lds);
            if (ra == hipSuccess) ra = dmalloc(e, &e->LUa, (size_t)(ld * ld));
            if (ra == hipSuccess) ra = dmalloc(e, &e->Ut, (size_t)(ld * ld));
            if (ra == hipSuccess) ra = dmalloc(e, &e->A_Nt, (size_t)(ld * e->ldn));
            if (ra != hipSuccess) {
                (void)hipGetLastError();
                if (pl == 3 || e->dual_bflip) {
                    set_err(errbuf, errlen, "pipeline 3: no device memory for the factors of the persistent-workgroup loop");
                    ellp_engine_destroy(e);
                    return ELLP_ERR_DEVICE;
                }
                e->small = e->mid = false;  // the large engine handles it
            }
        } else if (e->small) {
            int nt = small_threads(m, n_N);
            if (const char *ev = getenv("ELLP_SMALL_NT")) {  // measurement: force a workgroup size that still has a thread per row
                const int v = atoi(ev);
                if ((v == 64 || v == 128 || v == 256) && v >= m) nt = v;
            }
            e->small_nt = nt;
            const void *fn = small_kernel(e->kind, nt);
            hipError_t ra = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->small_lds);
            if (ra != hipSuccess) {
                (void)hipGetLastError();
                e->small = false;  // the large engine handles it
            }
        }
    }
    if (e->dual_bflip) {
        // the long-step ratio test exists in k_small / k_mid only (a selection walk and one more solve with the iteration's LU)
        if (!e->small && n_N > 0) {
            set_err(errbuf, errlen, "ELLP_FLAG_DUAL_BOUND_FLIPPING: the LU-per-iteration kernels take up to 1,024 rows (this LP: %lld)", (long long)m);
            ellp_engine_destroy(e);
            return ELLP_ERR_ARG;
        }
        if (n_N > 0 && dmalloc(e, &e->bf_list, (size_t)n_N + 1) != hipSuccess) {
            (void)hipGetLastError();
            set_err(errbuf, errlen, "ELLP_FLAG_DUAL_BOUND_FLIPPING: no device memory for the list of passed positions");
            ellp_engine_destroy(e);
            return ELLP_ERR_DEVICE;
        }
    }
    // Certified hybrid (restated in oracle/ellp_oracle.c: hybrid_run), the default for 128 < m <= 1024: the
    // explicit-inverse loop on the three-launch pipeline (its ratio-test fold sits in front of every store of the iteration,
    // so a guarded pivot can be refused by all blocks alike) with the pivot guard on; every terminal status and every
    // guarded iteration is re-examined by the LU-per-iteration kernel k_mid from the same arrays (exact_takeover).  Off with
    // an explicit pipeline, ELLP_FLAG_NO_CERTIFY, partial pricing, steepest edge, btran_mode 1 and on sharded engines.
    if (!e->small && e->opts.pipeline == 0 && !(e->opts.flags & ELLP_FLAG_NO_CERTIFY) && e->mid_lds > 0 && n_N > 0 && e->pp_P <= 1 &&
        !e->se && e->opts.btran_mode == 0 && getenv("ELLP_NO_HYBRID") == nullptr) {
        e->mid_nt = mid_threads(m);
        e->ldn = (n_N + 15) / 16 * 16;
        const void *fn = mid_kernel(e->kind, e->mid_nt);
        hipError_t ra = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->mid_lds);
        if (ra == hipSuccess) ra = dmalloc(e, &e->LUa, (size_t)(ld * ld));
        if (ra == hipSuccess) ra = dmalloc(e, &e->Ut, (size_t)(ld * ld));
        if (ra == hipSuccess) ra = dmalloc(e, &e->A_Nt, (size_t)(ld * e->ldn));
        if (ra == hipSuccess) {
            e->hybrid = true;
            e->guard_abs = 1e-7;
            if (const char *v = getenv("ELLP_GUARD_ABS"); v && v[0]) e->guard_abs = atof(v);  // diagnostics
            if (const char *v = getenv("ELLP_EXACT_K"); v && v[0] && atoi(v) > 0) e->exact_K = atoi(v);  // diagnostics
        } else {
            (void)hipGetLastError();  // no memory for the factors: the plain explicit-inverse engine
        }
    }
    // above 1,024 rows (the persistent kernel's limit): terminal statuses certified by an exact-LU iteration (ellp_exact.inc)
    if (!e->small && !e->hybrid && e->opts.pipeline == 0 && !(e->opts.flags & ELLP_FLAG_NO_CERTIFY) && m > MID_MAX_M && m <= 8192 &&
        n_N > 0 && e->pp_P <= 1 && !e->se && e->opts.btran_mode == 0 && getenv("ELLP_NO_HYBRID") == nullptr)
    {
        e->cert_large = true;
        e->guard_abs = 1e-7;
        if (const char *v = getenv("ELLP_GUARD_ABS"); v && v[0]) e->guard_abs = atof(v);  // diagnostics
        if (const char *v = getenv("ELLP_EXACT_K"); v && v[0] && atoi(v) > 0) e->exact_K = atoi(v);  // diagnostics
    }
    // two launches per primal iteration from m = 1024 (ellp_lagged.inc), or on request
    {
        const int pl = e->opts.pipeline;
        e->lagged = !e->small && !e->hybrid && !e->se && kind == ELLP_ENGINE_PRIMAL && e->opts.btran_mode == 0 && n_N > 0 && e->pp_P <= 1 &&
                    (pl == 2 || (pl == 0 && m >= 384));  // partial pricing runs on the three-launch pipeline; tools/pipeline_threshold.py for the size
        e->dual_fused = !e->small && !e->hybrid && kind == ELLP_ENGINE_DUAL && n_N > 0 && ld <= 4096 && (pl == 2 || (pl == 0 && m >= 384));
        e->dual_fold = e->dual_fused && e->ill_tol <= 0.0 && getenv("ELLP_DUAL_FOLD_OFF") == nullptr;
        if (e->lagged && e->price_wave && ld > 4096) e->price_wave = false;  // k_price2_wave keeps u in 8 double2 per thread
        e->price2_lds = sizeof(double) * (size_t)((m + 63) / 64) + 16;
    }
    // steepest-edge weights (ellp_se.inc): exact — at a signed-permutation basis from the columns alone, otherwise from B^-1 (below)
    if (e->se) {
        if (!e->vs_row) {
            set_err(errbuf, errlen, "no device memory for the steepest-edge tables");
            ellp_engine_destroy(e);
            return ELLP_ERR_DEVICE;
        }
        int32_t *perm = nullptr;
        ECHK(dmalloc(e, &e->se_gamma, (size_t)n_N));
        ECHK(dmalloc(e, &e->se_rho, (size_t)ld));
        ECHK(dmalloc(e, &e->se_v, (size_t)ld));
        // v = B^-T d accumulated by k_update2's row blocks (their rows of the old inverse are in registers there) where every
        // row of a block goes through the register path: <= 4 rows per block and NR > 0 (ld <= 4096); otherwise the GEMV
        if (ld <= 4096 && getenv("ELLP_SE_BTRAN") == nullptr) {
            if (e->upd2_rows > UPD_ROWS) {
                e->upd2_rows = UPD_ROWS;
                e->upd2_blocks = (int)((m + e->upd2_rows - 1) / e->upd2_rows);
            }
            ECHK(dmalloc(e, &e->se_vpart, (size_t)((int64_t)e->upd2_blocks * ld)));
        }
        ECHK(dmalloc(e, &perm, 4));
        ECHK(hipMemsetAsync(e->se_rho, 0, sizeof(double) * (size_t)ld, e->stream));
        ECHK(hipMemsetAsync(e->se_v, 0, sizeof(double) * (size_t)ld, e->stream));
        e->se_perm = perm;
        hipLaunchKernelGGL(k_se_perm, dim3(1), dim3(256), 0, e->stream, e->B_index, m, e->vs_row, e->vs_val, perm);
        hipLaunchKernelGGL(k_se_init, dim3((unsigned)((n_N + 3) / 4)), dim3(256), 0, e->stream, e->A_N, ld, m, n_N, perm, e->se_gamma);
        if (e->opts.flags & ELLP_FLAG_DENSE_PRICING) {  // the table was only needed for the permutation test
            e->pos_hint = nullptr;
        }
    }
    // initial B^-1 (k_small keeps none: its LU is redone every iteration, with the reference's guard)
    if (e->small) e->w_valid = false;
    else launch_refactor(e);
    if (e->se && e->se_perm) {  // steepest-edge weights at a basis that is not a signed permutation: exact, from the inverse just built
        const int64_t rt = (m + 63) / 64;
        double *part = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&part), sizeof(double) * (size_t)(rt * n_N)) == hipSuccess) {
            hipLaunchKernelGGL(k_se_exact_part, dim3((unsigned)((n_N + 63) / 64), (unsigned)rt), dim3(256), 0, e->stream, e->W, e->W2, e->A_N,
                               e->st, m, ld, n_N, e->se_perm, part);
            hipLaunchKernelGGL(k_se_exact_reduce, dim3((unsigned)((n_N + 255) / 256)), dim3(256), 0, e->stream, e->st, n_N, (int)rt,
                               e->se_perm, part, e->se_gamma);
            (void)hipStreamSynchronize(e->stream);
            (void)hipFree(part);
        } else {
            (void)hipGetLastError();  // no scratch memory: the weights stay at 1 (a reference-framework start)
        }
    }
    ECHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    ECHK(hipStreamSynchronize(e->stream));
    ECHK(hipGetLastError());
    prof_collect(e);
    if (e->h_st->status != ST_RUNNING) {
        ellp_status s = status_message(*e->h_st, errbuf, errlen);
        if (getenv("ELLP_DEBUG"))
            fprintf(stderr, "ellp: initial rebuild failed: status %d, column %lld, pivot row %lld, |pivot| %.3e\n", (int)s,
                    (long long)e->h_st->refk, (long long)e->h_st->r, e->h_st->d_r);
        ellp_engine_destroy(e);
        return s;
    }
#undef ECHK
    e->t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = e;
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_create(int kind, int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                               const double *b, const uint8_t *bound_kind, const double *lb, const double *ub,
                               const double *x, const int64_t *B_index, int64_t n_B, const int64_t *N_index,
                               const uint8_t *N_bound, int64_t n_N, const double *y, const double *d,
                               const ellp_opts *opts_in, ellp_engine **out, char *errbuf, size_t errlen) {
    return engine_create_impl(kind, m, n, n_c, A, c, b, bound_kind, lb, ub, x, B_index, n_B, N_index, N_bound, n_N, y, d,
                              opts_in, out, errbuf, errlen, n, false);
}

ellp_status ellp_engine_create_primal_phase1(int64_t m, int64_t n, const double *A, const double *b,
                                             const uint8_t *bound_kind, const double *lb, const double *ub,
                                             const double *x, const uint8_t *N_bound, const ellp_opts *opts_in,
                                             ellp_engine **out, char *errbuf, size_t errlen) {
    if (!out) return ELLP_ERR_ARG;
    *out = nullptr;
    if (m <= 0 || n <= 0 || !A || !b || !bound_kind || !lb || !ub || !x || !N_bound) {
        set_err(errbuf, errlen, "null pointer or inconsistent sizes");
        return ELLP_ERR_ARG;
    }
    // the phase-1 problem (primal_problem.rs:137-141, :236-253): costs 0 on the originals and 1 on the m
    // artificials, the artificials Lower(0) and basic, every original variable nonbasic at the bound the
    // caller names
    const int64_t nt = n + m;
    std::vector<double> c((size_t)nt, 0.0), lb2((size_t)nt, 0.0), ub2((size_t)nt, 0.0), x2((size_t)nt, 0.0);
    std::vector<uint8_t> kind2((size_t)nt, (uint8_t)ELLP_BOUND_LOWER), Nb(N_bound, N_bound + n);
    std::vector<int64_t> B((size_t)m), N((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        kind2[(size_t)j] = bound_kind[j];
        lb2[(size_t)j] = lb[j];
        ub2[(size_t)j] = ub[j];
        x2[(size_t)j] = x[j];
        N[(size_t)j] = j;
    }
    for (int64_t i = 0; i < m; ++i) {
        c[(size_t)(n + i)] = 1.0;
        B[(size_t)i] = n + i;
    }
    return engine_create_impl(ELLP_ENGINE_PRIMAL, m, nt, nt, A, c.data(), b, kind2.data(), lb2.data(), ub2.data(),
                              x2.data(), B.data(), m, N.data(), Nb.data(), n, nullptr, nullptr, opts_in, out, errbuf, errlen,
                              n, true);
}

// The explicit-inverse engine is about to be used on an engine that has been running k_small: build
// B^-1 from the current A_B and leave the small path for good.
static ellp_status ensure_inverse(ellp_engine *e, char *errbuf, size_t errlen) {
    if (e->dual_bflip) {  // the caller must be able to tell which rule runs: the explicit-inverse engine has no long-step ratio test
        set_err(errbuf, errlen, "ELLP_FLAG_DUAL_BOUND_FLIPPING: this call needs the explicit-inverse engine, which has no long-step ratio test");
        return ELLP_ERR_ARG;
    }
    launch_flush(e);  // two-launch pipeline: B^-1 is complete only once the open iteration has been booked
    if (e->w_valid) {
        e->small = false;
        return ELLP_OPTIMAL;
    }
    HIPCHK(hipSetDevice(e->device));
    launch_refactor(e);
    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    prof_collect(e);
    e->w_valid = true;
    e->small = false;
    e->need_dleave = true;
    e->u_valid = false;
    if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
    return ELLP_OPTIMAL;
}

// one launch of the LU-per-iteration kernel of ellp_mid.inc for up to `iters` loop bodies (run_small; exact_takeover)
static hipError_t launch_mid(ellp_engine *e, uint64_t iters, int resync) {
    const int64_t per = (int64_t)e->nbs * e->cpb;  // layout of the pricing buffer (Xchg, one segment)
    MidArgs a{};
    a.A_B = e->A_B; a.A_N = e->A_N; a.A_Nt = e->A_Nt; a.c_B = e->c_B; a.c_N = e->c_N; a.x = e->x; a.y = e->y; a.dd = e->dd;
    a.lb = e->lb; a.ub = e->ub; a.kind = e->kindv; a.B_index = e->B_index; a.N_index = e->N_index; a.Nb = e->Nb;
    a.kbuf = e->X + 2 * e->nbs;
    a.rbuf = e->X + 2 * e->nbs + per;
    a.LUa = e->LUa; a.Ut = e->Ut;
    a.st = e->st; a.m = e->m; a.ld = e->ld; a.nN = e->nN; a.ldn = e->ldn;
    a.max_iters = iters;
    a.nch = (int)((e->nN + 63) / 64);
    a.eps = e->eps;
    a.trace = Trace{e->trace_obj, e->trace_it, e->trace_len};
    a.stamps = e->small_stamps;
    a.maxviol = e->dual_maxviol;
    a.bflip = e->dual_bflip;
    a.flist = e->bf_list;
    a.resync = resync;
    a.b = e->b_dev;
    // the row-major copy of A_N the pricing pass reads: made afresh at every launch (anything may have
    // touched A_N in between: a hand-off, a sharding call, the other engine)
    hipLaunchKernelGGL(k_mid_transpose, dim3((unsigned)((e->m + 31) / 32), (unsigned)((e->nN + 31) / 32)), dim3(256), 0,
                       e->stream, e->A_N, e->A_Nt, e->m, e->ld, e->nN, e->ldn);
    void *kargs[] = {&a};
    return hipLaunchKernel(mid_kernel(e->kind, e->mid_nt), dim3(1), dim3((unsigned)e->mid_nt), kargs, e->mid_lds, e->stream);
}

// ellp_engine_run for the small path: launches of k_small, each good for up to 16384 iterations
static ellp_status run_small(ellp_engine *e, uint64_t max_iters, char *errbuf, size_t errlen) {
    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
    const uint64_t iters0 = e->h_st->iters;
    uint64_t remaining = max_iters;
    const int64_t per = (int64_t)e->nbs * e->cpb;  // layout of the pricing buffer (Xchg, one segment)
    while (remaining > 0) {
        if (e->mid) {
            HIPCHK(launch_mid(e, remaining < 4096 ? remaining : 4096, 0));
        } else {
            SmallArgs a{};
            a.A_B = e->A_B; a.A_N = e->A_N; a.c_B = e->c_B; a.c_N = e->c_N; a.x = e->x; a.y = e->y; a.dd = e->dd;
            a.lb = e->lb; a.ub = e->ub; a.kind = e->kindv; a.B_index = e->B_index; a.N_index = e->N_index; a.Nb = e->Nb;
            a.kbuf = e->X + 2 * e->nbs;
            a.rbuf = e->X + 2 * e->nbs + per;
            a.st = e->st; a.m = e->m; a.ld = e->ld; a.nN = e->nN;
            a.max_iters = remaining < 16384 ? remaining : 16384;
            a.nch = (int)((e->nN + 63) / 64);
            a.eps = e->eps;
            a.trace = Trace{e->trace_obj, e->trace_it, e->trace_len};
            a.stamps = e->small_stamps;
            a.maxviol = e->dual_maxviol;
            a.bflip = e->dual_bflip;
            a.flist = e->bf_list;
            void *kargs[] = {&a};
            HIPCHK(hipLaunchKernel(small_kernel(e->kind, e->small_nt), dim3(1), dim3((unsigned)e->small_nt), kargs,
                                   e->small_lds, e->stream));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->refactors = e->h_st->iters;  // one LU per loop body, as the reference
        if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
        const uint64_t done = e->h_st->iters - iters0;
        remaining = done < max_iters ? max_iters - done : 0;
    }
    return ELLP_MAXITER;
}

double ellp_engine_refresh(ellp_engine *e) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return NAN;
    if (hipSetDevice(e->device) != hipSuccess) return NAN;
    if (ensure_inverse(e, nullptr, 0) != ELLP_OPTIMAL) return NAN;
    launch_refresh(e);
    if (hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream) != hipSuccess) return NAN;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return NAN;
    prof_collect(e);
    const double res = e->h_st->resid;
    e->last_residual = res;
    if (e->h_st->need_rebuild) {  // an explicit refresh that was refused changes nothing and stops nothing
        const int32_t zero = 0;
        (void)hipMemcpyAsync(&e->st->need_rebuild, &zero, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        (void)hipMemcpyAsync(&e->st->tiny, &zero, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        (void)hipStreamSynchronize(e->stream);
        e->h_st->need_rebuild = 0;
        e->h_st->tiny = 0;
    }
    return res;
}

ellp_status ellp_engine_refactor(ellp_engine *e, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    e->w_valid = true;  // about to be
    e->small = false;
    e->need_dleave = true;
    launch_flush(e);
    launch_refactor(e);
    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    prof_collect(e);
    if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
    return ELLP_OPTIMAL;
}

static ellp_status run_colsharded(ellp_engine *e, uint64_t max_iters, ellp_stats *stats, char *errbuf, size_t errlen);

// ---- "certify or redo" (certified hybrid): the start of a phase is kept on the host; a solve that ends Optimal on a point
// that violates an invariant of the reference's loop (see k_invariants) is repeated from that start by the exact kernel
static hipError_t take_snapshot(ellp_engine *e) {
    auto &sn = e->snap;
    const size_t m = (size_t)e->m, nN = (size_t)e->nN, n_c = (size_t)e->n_c;
    sn.x.resize(n_c); sn.B.resize(m); sn.N.resize(nN); sn.Nb.resize(nN); sn.c_B.resize(m); sn.c_N.resize(nN);
    hipError_t rc;
#define SNAP(dst, src, bytes) if ((rc = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream)) != hipSuccess) return rc
    SNAP(sn.x.data(), e->x, 8 * n_c);
    SNAP(sn.B.data(), e->B_index, 8 * m);
    SNAP(sn.N.data(), e->N_index, 8 * nN);
    SNAP(sn.Nb.data(), e->Nb, nN);
    SNAP(sn.c_B.data(), e->c_B, 8 * m);
    SNAP(sn.c_N.data(), e->c_N, 8 * nN);
    if (e->kind == ELLP_ENGINE_DUAL) {
        sn.y.resize(m); sn.d.resize(n_c);
        SNAP(sn.y.data(), e->y, 8 * m);
        SNAP(sn.d.data(), e->dd, 8 * n_c);
    }
    SNAP(&sn.obj, &e->st->obj, sizeof(double));
#undef SNAP
    if ((rc = hipStreamSynchronize(e->stream)) != hipSuccess) return rc;
    sn.valid = true;
    return hipSuccess;
}

// true if the end point keeps the invariants of the reference's loop to within EPS (or cannot be examined)
static bool end_point_ok(ellp_engine *e, double *detail3) {
    if (!e->inv_out && dmalloc(e, &e->inv_out, 4) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    InvArgs a{e->x, e->lb, e->ub, e->b_dev, e->y, e->dd, e->kindv, e->Nb, e->N_index, e->m, e->n_c, e->nN,
              e->kind == ELLP_ENGINE_DUAL ? 1 : 0, e->inv_out};
    hipLaunchKernelGGL(k_invariants, dim3(1), dim3(1024), 0, e->stream, a);
    double out[3] = {0.0, 0.0, 0.0};
    if (hipMemcpyAsync(out, e->inv_out, sizeof(out), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    if (detail3) { detail3[0] = out[0]; detail3[1] = out[1]; detail3[2] = out[2]; }
    if (e->kind == ELLP_ENGINE_PRIMAL) return out[0] <= e->eps;
    if (!(out[1] <= e->eps)) return false;
    // a box problem's dual objective (= minus the original problem's dual infeasibility, the number the caller tests against
    // EPS, dual…:45-50) that lies BELOW -EPS but within what the drift of the carried d can produce: the exact loop decides
    if (e->box_problem && out[2] <= -e->eps && out[2] > -1e-6) return false;
    return true;
}

// back to the start of the phase, and from now on the LU-per-iteration kernel alone (ellp_mid.inc; `large`: run_exact_large)
static ellp_status redo_from_snapshot(ellp_engine *e, bool large, char *errbuf, size_t errlen) {
    const auto &sn = e->snap;
    const int64_t m = e->m, nN = e->nN, ld = e->ld;
    std::vector<int64_t> B((size_t)m), N((size_t)nN), where((size_t)e->n, INT64_MIN), src((size_t)(m + nN));
    HIPCHK(hipMemcpy(B.data(), e->B_index, 8 * (size_t)m, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(N.data(), e->N_index, 8 * (size_t)nN, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < m; ++i) where[(size_t)B[(size_t)i]] = i;
    for (int64_t j = 0; j < nN; ++j) where[(size_t)N[(size_t)j]] = -1 - j;
    for (int64_t i = 0; i < m; ++i) src[(size_t)i] = where[(size_t)sn.B[(size_t)i]];
    for (int64_t j = 0; j < nN; ++j) src[(size_t)(m + j)] = where[(size_t)sn.N[(size_t)j]];
    for (int64_t k = 0; k < m + nN; ++k)
        if (src[(size_t)k] == INT64_MIN) {
            set_err(errbuf, errlen, "redo: the index sets of the snapshot and of the engine differ");
            return ELLP_ERR_PANIC;
        }
    double *A2 = nullptr;
    int64_t *src_dev = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&A2), sizeof(double) * (size_t)(ld * (m + nN))));
    hipError_t rc = hipMalloc(reinterpret_cast<void **>(&src_dev), 8 * (size_t)(m + nN));
    if (rc != hipSuccess) { (void)hipFree(A2); HIPCHK(rc); }
    auto done = [&](hipError_t err) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipFree(A2);
        (void)hipFree(src_dev);
        return err;
    };
#define RCHK(expr) do { hipError_t _r = (expr); if (_r != hipSuccess) { HIPCHK(done(_r)); } } while (0)
    RCHK(hipMemcpyAsync(src_dev, src.data(), 8 * (size_t)(m + nN), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_restore_cols, dim3((unsigned)(m + nN)), dim3(256), 0, e->stream, e->A_B, e->A_N, src_dev, A2, ld);
    RCHK(hipMemcpyAsync(e->A_B, A2, sizeof(double) * (size_t)(ld * m), hipMemcpyDeviceToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->A_N, A2 + ld * m, sizeof(double) * (size_t)(ld * nN), hipMemcpyDeviceToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->x, sn.x.data(), 8 * (size_t)e->n_c, hipMemcpyHostToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->B_index, sn.B.data(), 8 * (size_t)m, hipMemcpyHostToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->N_index, sn.N.data(), 8 * (size_t)nN, hipMemcpyHostToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->Nb, sn.Nb.data(), (size_t)nN, hipMemcpyHostToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->c_B, sn.c_B.data(), 8 * (size_t)m, hipMemcpyHostToDevice, e->stream));
    RCHK(hipMemcpyAsync(e->c_N, sn.c_N.data(), 8 * (size_t)nN, hipMemcpyHostToDevice, e->stream));
    if (e->kind == ELLP_ENGINE_DUAL) {
        RCHK(hipMemcpyAsync(e->y, sn.y.data(), 8 * (size_t)m, hipMemcpyHostToDevice, e->stream));
        RCHK(hipMemcpyAsync(e->dd, sn.d.data(), 8 * (size_t)e->n_c, hipMemcpyHostToDevice, e->stream));
    }
    DevState ns = *e->h_st;
    ns.status = ST_RUNNING;
    ns.nan_flag = 0; ns.tiny = 0; ns.tiny_p = 0; ns.fin = 0; ns.need_rebuild = 0; ns.panic_code = 0; ns.open = 0;
    ns.pe_valid = 0; ns.mv_pending = 0; ns.se_valid = 0;
    ns.iters = ns.pivots = ns.flips = 0;
    ns.lambda = 0.0;
    ns.obj = sn.obj;
    ns.lr = -1;
    *e->h_st = ns;
    RCHK(hipMemcpyAsync(e->st, e->h_st, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
    if (e->trace_len > 0) RCHK(hipMemsetAsync(e->trace_it, 0, sizeof(unsigned long long) * (size_t)e->trace_len, e->stream));
    RCHK(done(hipSuccess));
#undef RCHK
    if (large) {
        // above 1,024 rows the exact loop is run_exact_large; the explicit inverse follows the restored basis (the exact
        // iterations keep updating it, and the next phase's fast loop starts from it)
        e->exact_large_only = true;
        launch_refactor(e);
        HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        prof_collect(e);
        e->hy_rebuilds += 1;
        if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
    } else {
        e->hybrid = false;
        e->small = true;
        e->mid = true;
        e->w_valid = false;
        e->lagged = false;
        e->dual_fused = e->dual_fold = false;
    }
    e->lag_open = e->dual_open = false;
    e->u_valid = false;
    e->need_dleave = true;
    e->maint_chain = 0;
    e->enqueued = 0;
    e->iters_seen = 0;
    e->since_refactor = e->since_btran = e->since_drift = 0;
    e->hy_redos += 1;
    return ELLP_OPTIMAL;
}

// One iteration with u / rho and B^-1 a_q from a fresh LU of the CURRENT basis (ellp_exact.inc), enqueued; the state is complete
// afterwards.  The workspace must be there (exact_workspace) and the caller has switched the pivot guard off (guard_off).
static void enqueue_exact_iteration(ellp_engine *e) {
    const int64_t m = e->m, ld = e->ld;
    const unsigned gm = (unsigned)((m + 255) / 256), gld = (unsigned)((ld + 255) / 256);
    const size_t lds0 = sizeof(double) * (size_t)m, lds1 = 2 * sizeof(double) * (size_t)m;
    const bool was_lagged = e->lagged, was_fused = e->dual_fused, was_fold = e->dual_fold;
        hipLaunchKernelGGL(k_rows_from_cols, dim3((unsigned)((m + 31) / 32), (unsigned)((m + 31) / 32)), dim3(256), 0, e->stream, e->A_B, e->luw.M, m, ld);
        ellp_lu_rows_factor(&e->luw, e->stream);
        if (e->kind == ELLP_ENGINE_PRIMAL) {
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), lds1, e->stream, e->luw.M, e->luw.piv, m, e->c_B, e->ex_sol, 1, e->ex_fail);
            hipLaunchKernelGGL(k_exact_put_u, dim3(gld), dim3(256), 0, e->stream, e->ex_sol, e->u, m, ld, e->ex_fail);
            e->lagged = false;
            launch_price<0>(e);
            launch_ftran2<0>(e);
            hipLaunchKernelGGL(k_exact_gather_aq, dim3(gm), dim3(256), 0, e->stream, e->A_N, e->st, m, ld, e->ex_rhs);
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), lds0, e->stream, e->luw.M, e->luw.piv, m, e->ex_rhs, e->ex_sol, 0, e->ex_fail);
            ExactLamArgs la{e->ex_sol, e->d, e->lam, e->bidx, e->dpos, e->B_index, e->x, e->lb, e->ub, e->kindv, e->st, m, e->eps, e->ex_fail};
            hipLaunchKernelGGL(k_exact_relam, dim3(gm), dim3(256), 0, e->stream, la);
            launch_update2<0>(e, 1);
            e->lagged = was_lagged;
        } else {
            // x_B = A_B^-1 (b - A_N x_N) from the LU (the kernels of launch_resync form the right-hand side)
            ResyncArgs ra{e->A_N, e->W, e->W2, e->b_dev, e->x, e->xg, e->tvec, e->upart, e->cand, e->maxbits, e->B_index,
                          e->N_index, e->st, e->m, e->ld, e->nN, 0, e->btran_tiles, 1};
            ra.cols_per_tile = (int)((e->nN + e->btran_tiles - 1) / e->btran_tiles);
            const int64_t half = ld >> 1;
            hipLaunchKernelGGL(k_resync_gather, dim3((unsigned)((e->nN + 255) / 256)), dim3(256), 0, e->stream, ra);
            hipLaunchKernelGGL(k_resync_part, dim3((unsigned)((half + 255) / 256), (unsigned)e->btran_tiles), dim3(256), 0, e->stream, ra);
            hipLaunchKernelGGL(k_resync_rhs, dim3(gld), dim3(256), 0, e->stream, ra);
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), lds0, e->stream, e->luw.M, e->luw.piv, m, e->tvec, e->cand, 0, e->ex_fail);
            hipLaunchKernelGGL(k_resync_apply, dim3(gm), dim3(256), 0, e->stream, ra);
            launch_dleave(e);
            hipLaunchKernelGGL(k_exact_unit, dim3(gm), dim3(256), 0, e->stream, e->ex_rhs, m, e->st);
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), lds1, e->stream, e->luw.M, e->luw.piv, m, e->ex_rhs, e->ex_rho, 1, e->ex_fail);
            e->dual_fused = e->dual_fold = false;
            e->price_rho_ovr = e->ex_rho;
            e->dual_seq += 1;
            launch_price<1>(e);
            launch_ftran2<1>(e);
            hipLaunchKernelGGL(k_exact_gather_aq, dim3(gm), dim3(256), 0, e->stream, e->A_N, e->st, m, ld, e->ex_rhs);
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), lds0, e->stream, e->luw.M, e->luw.piv, m, e->ex_rhs, e->ex_sol, 0, e->ex_fail);
            hipLaunchKernelGGL(k_exact_copy, dim3(gm), dim3(256), 0, e->stream, e->ex_sol, e->d, m, e->st, e->ex_fail);
            launch_update2<1>(e, 0);
            e->price_rho_ovr = nullptr;
            e->dual_fused = was_fused;
            e->dual_fold = was_fold;
        }
}

// the LU workspace and the vectors of the exact iteration (allocated at the first use)
static hipError_t exact_workspace(ellp_engine *e) {
    if (e->luw_ready) return hipSuccess;
    const int64_t m = e->m, ld = e->ld;
    hipError_t rc;
    if ((rc = ellp_lu_rows_alloc(&e->luw, m)) != hipSuccess) {
        ellp_lu_rows_free(&e->luw);
        (void)hipGetLastError();
        return rc;
    }
    e->luw_ready = true;
    if ((rc = dmalloc(e, &e->ex_rhs, (size_t)ld)) != hipSuccess) return rc;
    if ((rc = dmalloc(e, &e->ex_sol, (size_t)ld)) != hipSuccess) return rc;
    if ((rc = dmalloc(e, &e->ex_rho, (size_t)ld)) != hipSuccess) return rc;
    if ((rc = dmalloc(e, &e->ex_fail, 4)) != hipSuccess) return rc;
    (void)hipMemsetAsync(e->ex_rho, 0, sizeof(double) * (size_t)ld, e->stream);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lu_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * m));
    return hipSuccess;
}

// "certify or redo" above 1,024 rows: after a redo every loop body runs on a fresh LU (ellp_engine::exact_large_only) — the
// reference's LU-per-iteration loop on all CUs instead of in one workgroup: an LU (2 m launches) and two (primal) / three (dual) solves per iteration,
// 20-40 ms at 1,000-2,000 rows; the explicit inverse is still updated along (a later phase goes back to the fast loop).
static ellp_status run_exact_large(ellp_engine *e, uint64_t max_iters, char *errbuf, size_t errlen) {
    HIPCHK(exact_workspace(e));
    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
    const uint64_t iters0 = e->h_st->iters;
    e->lag_open = e->dual_open = false;
    e->guard_off = true;
    ellp_status result = ELLP_MAXITER;
    while (e->h_st->iters - iters0 < max_iters) {
        HIPCHK(hipMemsetAsync(e->ex_fail, 0, sizeof(int), e->stream));
        enqueue_exact_iteration(e);
        int failed = 0;
        HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&failed, e->ex_fail, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipGetLastError());
        prof_collect(e);
        if (failed) {  // a zero on U's diagonal: the reference's unwrap() on None in BTRAN / FTRAN
            set_err(errbuf, errlen, "unwrap() on None: the basis is exactly singular");
            result = ELLP_ERR_PANIC;
            break;
        }
        if (e->h_st->tiny) {  // raised by the update kernel for the explicit inverse's sake: nothing to maintain here
            static const int32_t zero = 0;
            (void)hipMemcpyAsync(&e->st->tiny, &zero, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
            e->h_st->tiny = 0;
        }
        if (e->h_st->status != ST_RUNNING) {
            result = status_message(*e->h_st, errbuf, errlen);
            break;
        }
    }
    e->guard_off = false;
    e->u_valid = false;
    e->enqueued = 0;
    e->iters_seen = e->h_st->iters;
    e->hy_exact_iters += e->h_st->iters - iters0;
    e->need_dleave = true;
    if (result != ELLP_MAXITER && e->h_st->status != ST_RUNNING) {
        // the solve has ended: the explicit inverse, updated along through pivots no guard has looked at, is rebuilt from
        // the final basis (a phase hand-off and the dual point of read_point read it)
        static const int32_t running = ST_RUNNING;
        static int32_t keep;
        keep = e->h_st->status;
        (void)hipMemcpyAsync(&e->st->status, &running, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        launch_refactor(e);
        e->hy_rebuilds += 1;
        DevState after;
        HIPCHK(hipMemcpyAsync(&after, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        prof_collect(e);
        if (after.status != ST_RUNNING) e->w_valid = false;
        e->h_st->cur = after.cur;
        (void)hipMemcpyAsync(&e->st->status, &keep, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        (void)hipMemcpyAsync(&e->st->panic_code, &e->h_st->panic_code, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        (void)hipStreamSynchronize(e->stream);
    }
    return result;
}

// The certificate above 1,024 rows (ellp_exact.inc): the explicit-inverse loop has reported a terminal status; one iteration
// is run with u / rho and d = B^-1 a_q from a fresh LU of the basis (dual: x_B recomputed from it first).  Returns as
// exact_takeover does.
static int exact_certify_large(ellp_engine *e, uint64_t remaining, ellp_status *result, char *errbuf, size_t errlen) {
    const int s = e->h_st->status;
    const bool guard = s == ST_NEED_EXACT;  // a refused pivot: nothing of that iteration is committed or counted
    if (!(guard || s == ELLP_OPTIMAL || s == ELLP_INFEASIBLE || s == ELLP_UNBOUNDED)) return 0;
    if (guard && remaining == 0) return 0;  // the slice is used up: the next one starts here
    if (!guard) remaining += 1;  // the loop body that found the status is examined again, not counted twice
    auto fail = [&](hipError_t rc) {
        set_err(errbuf, errlen, "HIP error %s in the certificate of the terminal status", hipGetErrorString(rc));
        *result = ELLP_ERR_DEVICE;
        return 2;
    };
    hipError_t rc;
    if ((rc = exact_workspace(e)) != hipSuccess) {
        e->cert_large = false;  // no memory for the factors: the status goes out uncertified
        e->hy_uncertified += 1;
        return 0;
    }
    // re-arm: the state is complete (a status found by k_ftran_eta comes with that pass's eta update done; one found by a
    // ratio-test fold comes before anything of its iteration is committed); the loop body that found it is not counted twice
    DevState ns = *e->h_st;
    ns.status = ST_RUNNING;
    ns.nan_flag = 0; ns.tiny = 0; ns.tiny_p = 0; ns.fin = 0; ns.need_rebuild = 0; ns.panic_code = 0;
    ns.open = 0; ns.pe_valid = 0; ns.mv_pending = 0; ns.usel = 0; ns.usel_next = 0;
    if (!guard && ns.iters > 0) ns.iters -= 1;
    *e->h_st = ns;
    if ((rc = hipMemcpyAsync(e->st, e->h_st, sizeof(DevState), hipMemcpyHostToDevice, e->stream)) != hipSuccess) return fail(rc);
    if ((rc = hipMemsetAsync(e->ex_fail, 0, sizeof(int), e->stream)) != hipSuccess) return fail(rc);
    e->guard_off = true;
    e->lag_open = false;
    e->dual_open = false;
    // the first iteration examines the status; if it does NOT confirm it (it pivots), up to exact_K - 1 more follow before the
    // explicit-inverse loop takes over again — the policy of the certified hybrid (oracle/ellp_oracle.c, hybrid_run)
    int failed = 0;
    uint64_t ran = 0;
    int s_first = ST_RUNNING;
    for (int k = 0; k < e->exact_K; ++k) {
        enqueue_exact_iteration(e);
        if ((rc = hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream)) != hipSuccess) return fail(rc);
        if ((rc = hipMemcpyAsync(&failed, e->ex_fail, sizeof(int), hipMemcpyDeviceToHost, e->stream)) != hipSuccess) return fail(rc);
        if ((rc = hipStreamSynchronize(e->stream)) != hipSuccess) return fail(rc);
        if ((rc = hipGetLastError()) != hipSuccess) return fail(rc);
        ran += 1;
        if (k == 0) s_first = e->h_st->status;
        if (e->h_st->status != ST_RUNNING || e->h_st->tiny || failed) break;
        if (ran >= remaining) break;  // the caller's budget
    }
    e->guard_off = false;
    prof_collect(e);
    const int s2 = e->h_st->status;
    e->hy_exact_iters += ran;
    if (failed) e->hy_uncertified += 1;  // an exactly singular basis: the iteration ran on the explicit inverse's numbers
    if (guard) e->hy_guards += 1;
    else {
        e->hy_certs += 1;
        if (s_first != s) e->hy_disagree += 1;
    }
    if (getenv("ELLP_HYBRID_DEBUG"))
        fprintf(stderr, "ellp hybrid: fast status %d at iteration %llu -> exact-LU iteration, status %d%s\n", s,
                (unsigned long long)ns.iters, s2, failed ? " (LU singular: not certified)" : "");
    e->u_valid = false;
    e->since_btran = 0;
    e->maint_chain = 0;
    e->enqueued = 0;
    e->iters_seen = e->h_st->iters;
    e->need_dleave = false;
    if (s2 == ST_RUNNING && !e->h_st->tiny) return 1;
    if (s2 == ST_RUNNING) return 1;  // a maintenance request raised by the iteration: the run loop services it
    *result = status_message(*e->h_st, errbuf, errlen);
    return 2;
}

// Certified hybrid (restated in oracle/ellp_oracle.c, hybrid_run): the explicit-inverse loop has stopped — on a guarded
// pivot (ST_NEED_EXACT: nothing of that iteration is committed), on a terminal status, or on an error of its own arithmetic
// (singular rebuild, NaN, an assertion of the reference).  h_st is the drained device state.  The LU-per-iteration kernel
// (k_mid: the reference's arithmetic on a fresh factorisation, primal…:173-189,289-292,404-406; dual…:241-246,281-284)
// runs up to exact_K loop bodies from the same arrays; dual engines recompute x_B = A_B^-1 (b - A_N x_N) from that LU
// first.  The loop body in which a terminal status was found is examined again, not counted twice.
// Returns 0: nothing to do; 1: the loop goes on (status RUNNING, B^-1 rebuilt from the basis k_mid left);
// 2: the solve has ended, *result holds the status k_mid found (certified, or corrected).
static int exact_takeover(ellp_engine *e, uint64_t remaining, ellp_status *result, char *errbuf, size_t errlen) {
    if (e->cert_large && e->world == 1 && !e->colshard) return exact_certify_large(e, remaining, result, errbuf, errlen);
    if (!e->hybrid || e->world != 1 || e->colshard) return 0;
    const int s = e->h_st->status;
    const bool guard = s == ST_NEED_EXACT;
    const bool terminal = s == ELLP_OPTIMAL || s == ELLP_INFEASIBLE || s == ELLP_UNBOUNDED;
    const bool failed = s == ELLP_ERR_SINGULAR || s == ELLP_ERR_NAN || s == ELLP_ERR_PANIC;
    if (!guard && !terminal && !failed) return 0;
    DevState ns = *e->h_st;
    uint64_t budget = remaining;
    if (terminal && ns.iters > 0) {
        ns.iters -= 1;
        budget += 1;
    }
    if (budget == 0) return 0;  // a guarded pivot at the very end of the caller's slice: the next slice starts here
    ns.status = ST_RUNNING;
    ns.nan_flag = 0;
    ns.tiny = 0;
    ns.tiny_p = 0;
    ns.fin = 0;
    ns.need_rebuild = 0;
    ns.panic_code = 0;
    auto fail = [&](hipError_t rc) {
        set_err(errbuf, errlen, "HIP error %s in the hand-over to the exact kernel", hipGetErrorString(rc));
        *result = ELLP_ERR_DEVICE;
        return 2;
    };
    hipError_t rc;
    *e->h_st = ns;
    if ((rc = hipMemcpyAsync(e->st, e->h_st, sizeof(DevState), hipMemcpyHostToDevice, e->stream)) != hipSuccess) return fail(rc);
    const uint64_t K = budget < (uint64_t)e->exact_K ? budget : (uint64_t)e->exact_K;
    static const int dual_resync = [] {  // diagnostics: ELLP_HYBRID_RESYNC = 1 (default) recomputes x_B only, 2 x_B, y and d
        const char *v = getenv("ELLP_HYBRID_RESYNC");
        return v && v[0] ? atoi(v) : 1;
    }();
    if ((rc = launch_mid(e, K, e->kind == ELLP_ENGINE_DUAL ? dual_resync : 0)) != hipSuccess) return fail(rc);
    if ((rc = hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream)) != hipSuccess) return fail(rc);
    if ((rc = hipStreamSynchronize(e->stream)) != hipSuccess) return fail(rc);
    const uint64_t did = e->h_st->iters - ns.iters;
    const int s2 = e->h_st->status;
    e->hy_exact_iters += did;
    if (guard) e->hy_guards += 1;
    else {
        e->hy_certs += 1;
        if (!(s2 == s && did <= 1)) e->hy_disagree += 1;
    }
    if (getenv("ELLP_HYBRID_DEBUG"))
        fprintf(stderr, "ellp hybrid: fast status %d at iteration %llu -> exact kernel, %llu iterations, status %d\n", s,
                (unsigned long long)ns.iters, (unsigned long long)did, s2);
    e->u_valid = false;
    e->since_btran = 0;
    e->maint_chain = 0;
    e->enqueued = 0;
    e->iters_seen = e->h_st->iters;
    if (e->h_st->pivots != ns.pivots || failed) {
        // the explicit inverse follows the basis (also when the solve has ended: a phase hand-off reads it)
        static const int32_t running = ST_RUNNING;
        if (s2 != ST_RUNNING) (void)hipMemcpyAsync(&e->st->status, &running, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        launch_refactor(e);
        e->hy_rebuilds += 1;
        DevState after;
        if ((rc = hipMemcpyAsync(&after, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream)) != hipSuccess) return fail(rc);
        if ((rc = hipStreamSynchronize(e->stream)) != hipSuccess) return fail(rc);
        prof_collect(e);
        if (after.status != ST_RUNNING) {
            if (s2 == ST_RUNNING) {  // the basis the exact kernel left fails the rebuild's guard: that is the loop's next LU
                *e->h_st = after;
                *result = status_message(after, errbuf, errlen);
                return 2;
            }
            e->w_valid = false;
        }
        e->h_st->cur = after.cur;
        if (s2 != ST_RUNNING) {
            static int32_t keep;
            keep = s2;
            (void)hipMemcpyAsync(&e->st->status, &keep, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
            (void)hipMemcpyAsync(&e->st->panic_code, &e->h_st->panic_code, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
            (void)hipStreamSynchronize(e->stream);
        }
    }
    if (s2 == ST_RUNNING) {
        if (e->kind == ELLP_ENGINE_DUAL) launch_dleave(e);  // the three-launch loop starts from DevState::lr
        return 1;
    }
    *result = status_message(*e->h_st, errbuf, errlen);
    return 2;
}

ellp_status ellp_engine_run(ellp_engine *e, uint64_t max_iters, ellp_stats *stats, char *errbuf, size_t errlen) {
    if (!e) return ELLP_ERR_ARG;
    if (errbuf && errlen) errbuf[0] = 0;
    HIPCHK(hipSetDevice(e->device));
    e->obj_fresh = false;
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t poll = e->opts.poll_interval > 0 ? (uint64_t)e->opts.poll_interval : (e->m <= 256 ? 64 : 16);
    int64_t period = e->refactor_period;
    if (period <= 0) period = default_period(e);
    ellp_status result = ELLP_MAXITER;
    if ((e->hybrid || e->cert_large) && !e->snap.valid && e->world == 1 && !e->colshard && e->nN > 0 && getenv("ELLP_NO_REDO") == nullptr) {
        launch_flush(e);
        HIPCHK(take_snapshot(e));  // the start of the phase (certify or redo, see end_point_ok)
    }
    if (e->nN == 0) {
        result = ELLP_OPTIMAL;  // primal…:149-151 / dual…:175-177
    } else if (e->small && e->world == 1) {
        result = run_small(e, max_iters, errbuf, errlen);
    } else if (e->exact_large_only && e->world == 1 && !e->colshard) {
        result = run_exact_large(e, max_iters, errbuf, errlen);
    } else {
        uint64_t remaining = max_iters;
        if (e->colshard) return run_colsharded(e, max_iters, stats, errbuf, errlen);
        if (e->world != 1) {
            set_err(errbuf, errlen, "a sharded engine is driven with ellp_engine_step + an all-gather (ellp_amd/dist.py)");
            return ELLP_ERR_ARG;
        }
        if (e->kind == ELLP_ENGINE_DUAL && remaining > 0 && e->need_dleave) {
            launch_dleave(e);
            e->need_dleave = false;
        }
        // a previous slice may already have terminated (h_st is current if the previous call was a run that read it)
        if (!e->hst_fresh) {
            HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
        e->hst_fresh = false;
        reconcile_counters(e);
        adopt_fin(e);
        const uint64_t iters0 = e->h_st->iters;
        if (e->h_st->status == ST_NEED_EXACT && remaining > 0) {  // a guarded pivot closed the previous slice
            const int tk = exact_takeover(e, remaining, &result, errbuf, errlen);
            const uint64_t done0 = e->h_st->iters - iters0;
            remaining = (tk == 2 || done0 >= max_iters) ? 0 : max_iters - done0;
        } else if (e->h_st->status == ST_NEED_EXACT) {
            remaining = 0;  // nothing to run: the status stays for the next slice
        } else if (e->h_st->status != ST_RUNNING) {
            result = status_message(*e->h_st, errbuf, errlen);
            remaining = 0;
        }
        const bool can_look_ahead = e->ill_tol <= 0.0 && !e->opts.profile;
        while (remaining > 0 && result == ELLP_MAXITER) {
            // Look-ahead polling (no tiny-pivot maintenance, no profiling, no follow-up refresh due):
            // batch k+1 is enqueued BEFORE the host waits for the status of batch k, so the stream never
            // drains while the host looks at a read-back.  A batch enqueued after termination — or after a
            // maintenance request of the drift monitor — is a few no-op launches (every kernel returns at
            // once when status != RUNNING).  Iterations are counted on the device afterwards.
            if (can_look_ahead && e->maint_chain == 0) {
                if (!e->look_ev[0]) {
                    HIPCHK(hipEventCreateWithFlags(&e->look_ev[0], hipEventDisableTiming));
                    HIPCHK(hipEventCreateWithFlags(&e->look_ev[1], hipEventDisableTiming));
                }
                const uint64_t lpoll = e->opts.poll_interval > 0 ? (uint64_t)e->opts.poll_interval : 64;
                bool pending[2] = {false, false};
                bool stop = false;
                int slot = 0, last_slot = -1, waited_slot = -1;
                bool last_is_final = false;
                uint64_t to_launch = remaining;
                while (!stop) {
                    if (to_launch > 0) {
                        const uint64_t batch = to_launch < lpoll ? to_launch : lpoll;
                        for (uint64_t it = 0; it < batch; ++it) {
                            if (e->since_refactor >= (uint64_t)period) maintain_inverse(e);
                            if (e->kind == ELLP_ENGINE_PRIMAL) launch_primal_iteration(e);
                            else launch_dual_iteration(e);
                        }
                        to_launch -= batch;
                        launch_dual_close(e);  // a batch ends on a complete state (a fused dual iteration may be open)
                        if (to_launch == 0 && e->kind == ELLP_ENGINE_PRIMAL) {
                            // the objective ellp_stats reports (c . x) rides on the last read-back instead of costing a
                            // launch and a synchronisation of its own after the loop (fill_stats)
                            hipLaunchKernelGGL(k_primal_obj, dim3(1), dim3(1024), 0, e->stream, e->c_B, e->c_N, e->x, e->B_index,
                                               e->N_index, e->m, e->nN, e->st);
                        }
                        HIPCHK(hipMemcpyAsync(&e->h_look[slot], e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
                        HIPCHK(hipEventRecord(e->look_ev[slot], e->stream));
                        pending[slot] = true;
                        last_slot = slot;
                        last_is_final = to_launch == 0;
                    }
                    const int other = slot ^ 1;
                    const int wait_on = pending[other] ? other : (to_launch == 0 && pending[slot] ? slot : -1);
                    if (wait_on >= 0) {
                        HIPCHK(hipEventSynchronize(e->look_ev[wait_on]));
                        pending[wait_on] = false;
                        waited_slot = wait_on;
                        if (e->h_look[wait_on].status != ST_RUNNING || e->h_look[wait_on].tiny || e->h_look[wait_on].fin) stop = true;
                    }
                    if (to_launch == 0 && !pending[0] && !pending[1]) break;
                    slot ^= 1;
                }
                if (last_is_final && waited_slot == last_slot && !pending[0] && !pending[1]) {
                    // the read-back just waited for came behind everything that was enqueued: it IS the final state
                    *e->h_st = e->h_look[last_slot];
                    e->obj_fresh = e->kind == ELLP_ENGINE_PRIMAL;
                } else {
                    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
                    HIPCHK(hipStreamSynchronize(e->stream));
                }
                HIPCHK(hipGetLastError());
                reconcile_counters(e);
                adopt_fin(e);
                if (getenv("ELLP_RUN_DEBUG"))
                    fprintf(stderr, "ellp run (look-ahead): status %d iters %llu tiny %d fin %d need_rebuild %d lr %lld obj %.17g\n", e->h_st->status,
                            (unsigned long long)e->h_st->iters, e->h_st->tiny, e->h_st->fin, e->h_st->need_rebuild, (long long)e->h_st->lr, e->h_st->obj);
                const uint64_t done = e->h_st->iters - iters0;
                remaining = done < max_iters ? max_iters - done : 0;
                if (service_maintenance_request(e)) continue;  // refreshed; the follow-up runs in the loop below
                if (e->h_st->status != ST_RUNNING) {
                    const int tk = exact_takeover(e, remaining, &result, errbuf, errlen);
                    if (tk == 1) {
                        const uint64_t done2 = e->h_st->iters - iters0;
                        remaining = done2 < max_iters ? max_iters - done2 : 0;
                    } else if (tk == 0) {
                        if (e->h_st->status == ST_NEED_EXACT) break;  // the slice is used up: the next one starts with the hand-over
                        result = status_message(*e->h_st, errbuf, errlen);
                    }
                }
                continue;
            }
            const bool chained = e->maint_chain > 0;
            const uint64_t batch = chained ? 1 : (remaining < poll ? remaining : poll);
            for (uint64_t it = 0; it < batch; ++it) {
                if (e->since_refactor >= (uint64_t)period) maintain_inverse(e);
                if (e->kind == ELLP_ENGINE_PRIMAL) launch_primal_iteration(e);
                else launch_dual_iteration(e);
            }
            // this path serves the reactive maintenance of small LPs (and profiling): its follow-up refresh is
            // meant to come AFTER the iteration that follows a tiny pivot, so that iteration is completed here
            if (e->ill_tol > 0.0) launch_flush(e);
            launch_dual_close(e);
            HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            HIPCHK(hipGetLastError());
            prof_collect(e);
            reconcile_counters(e);
            adopt_fin(e);
            if (getenv("ELLP_RUN_DEBUG"))
                fprintf(stderr, "ellp run (polled): status %d iters %llu tiny %d fin %d need_rebuild %d lr %lld obj %.17g\n", e->h_st->status,
                        (unsigned long long)e->h_st->iters, e->h_st->tiny, e->h_st->fin, e->h_st->need_rebuild, (long long)e->h_st->lr, e->h_st->obj);
            // iterations that really ran (a maintenance request voids the rest of its batch)
            const uint64_t done = e->h_st->iters - iters0;
            remaining = done < max_iters ? max_iters - done : 0;
            if (chained) e->maint_chain = 0;
            if (service_maintenance_request(e)) continue;
            if (chained && e->h_st->status == ST_RUNNING) maintain_inverse(e, true);  // the follow-up refresh
            if (e->h_st->status != ST_RUNNING) {
                const int tk = exact_takeover(e, remaining, &result, errbuf, errlen);
                if (tk == 1) {
                    const uint64_t done2 = e->h_st->iters - iters0;
                    remaining = done2 < max_iters ? max_iters - done2 : 0;
                } else if (tk == 0) {
                    if (e->h_st->status == ST_NEED_EXACT) break;  // the slice is used up: the next one starts with the hand-over
                    result = status_message(*e->h_st, errbuf, errlen);
                }
            }
        }
    }
    // Certify or redo: a hybrid solve that has ended Optimal (certified by the exact kernel) on a point that does not keep the
    // invariants of the reference's loop — the explicit-inverse stretch has let x, or the dual's carried d, drift past EPS:
    // the caller's next test on that point (the phase-1 objective against EPS, primal…:42-50 / dual…:45-50; the assertions of
    // DualPhase2::from, dual_problem.rs:293-310) would fail where the reference's own arithmetic passes — is repeated from the
    // start of the phase by the LU-per-iteration kernel alone: from there on the engine IS the reference's loop, bit for bit.
    if (result == ELLP_OPTIMAL && (e->hybrid || (e->cert_large && !e->exact_large_only)) && e->snap.valid && e->world == 1) {
        double det[3] = {0.0, 0.0, 0.0};
        bool ok = end_point_ok(e, det);
        if (getenv("ELLP_FORCE_REDO")) ok = false;  // tests: the redo path itself (snapshot, restore, the exact kernel from the start)
        const bool large = !e->hybrid;
        if (!ok && large) {
            // Above 1,024 rows the exact loop costs an LU of 2 m launches and two or three solves of m steps per iteration (about
            // 18 us x m): the redo is taken when the iterations this phase needed, at that price, stay within
            // ELLP_REDO_MAX_SECONDS (default 900 — a 1,850-row phase of 3,000 iterations: 100 s; config 3's 600,000: never);
            // otherwise the point goes out as it is, counted as uncertified (ELLP_TAP_STATE).
            const char *capv = getenv("ELLP_REDO_MAX_SECONDS");  // read when it matters: once per phase end that fails the check
            const double cap = capv && capv[0] ? atof(capv) : 900.0;
            const double est = (double)e->h_st->iters * 18e-6 * (double)e->m;
            if (est > cap) {
                if (getenv("ELLP_HYBRID_DEBUG"))
                    fprintf(stderr, "ellp hybrid: end point violates an invariant (x %.3e, d %.3e); a redo would take about %.0f s: not done\n", det[0], det[1], est);
                e->hy_uncertified += 1;
                ok = true;
            }
        }
        if (!ok && large) {
            if (getenv("ELLP_HYBRID_DEBUG"))
                fprintf(stderr, "ellp hybrid: end point violates an invariant (x %.3e, d %.3e, objective %.17g): redo on fresh LUs\n", det[0], det[1], det[2]);
            const ellp_status rs = redo_from_snapshot(e, true, errbuf, errlen);
            if (rs != ELLP_OPTIMAL) return rs;
            result = run_exact_large(e, e->opts.max_iter, errbuf, errlen);
            e->obj_fresh = false;
        } else if (!ok) {
            if (getenv("ELLP_HYBRID_DEBUG"))
                fprintf(stderr, "ellp hybrid: end point violates an invariant (x %.3e, d %.3e, objective %.17g against the carried %.17g): redo\n",
                        det[0], det[1], det[2], e->h_st->obj);
            const ellp_status rs = redo_from_snapshot(e, false, errbuf, errlen);
            if (rs != ELLP_OPTIMAL) return rs;
            result = run_small(e, e->opts.max_iter, errbuf, errlen);
            e->obj_fresh = false;
        }
    }
    // The caller's budget (ellp_opts.max_iter) is spent: the reference has run that many FULL loop bodies
    // (primal…:162-202), so an unbounded ray, a panic or a NaN found by the ratio test of the last one is its
    // result, not MaxIter.  On the two-launch pipeline that ratio test is still open (its fold belongs to the next
    // pricing launch, which will not come): close it and take the status from behind the closing kernel.  Slices
    // inside the budget stay open — that is what makes slicing free.
    if (result == ELLP_MAXITER && e->lag_open && !e->small && e->world == 1 && e->h_st->iters >= e->opts.max_iter) {
        launch_flush(e);
        hipLaunchKernelGGL(k_primal_obj, dim3(1), dim3(1024), 0, e->stream, e->c_B, e->c_N, e->x, e->B_index, e->N_index, e->m,
                           e->nN, e->st);
        HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        reconcile_counters(e);
        adopt_fin(e);
        e->obj_fresh = true;
        if (e->h_st->status != ST_RUNNING && e->h_st->status != ST_NEED_MAINT) result = status_message(*e->h_st, errbuf, errlen);
    }
    if (stats) {
        fill_stats(e, stats);
        stats->t_loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // A slice that ends normally has read h_st behind its last iteration; what may have been enqueued after that
    // read (fill_stats' objective kernel, a follow-up refresh) changes neither the status nor the iteration
    // counter, and a request it raises is seen by the next slice's first read-back.  So the next ellp_engine_run
    // need not start with a read-back of its own (15 us of a 20-iteration slice).
    e->hst_fresh = result == ELLP_MAXITER && !e->small && e->world == 1 && !e->colshard && e->nN > 0 &&
                   e->h_st->status == ST_RUNNING;
    return result;
}

ellp_status ellp_engine_read_point(ellp_engine *e, double *x, int64_t *B_index, int64_t *N_index, uint8_t *N_bound,
                                   double *y, double *d, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    const bool was_open = e->lag_open;
    launch_flush(e);  // two-launch pipeline: fold and book the iteration that is still open
    if (was_open) HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    if (x) HIPCHK(hipMemcpyAsync(x, e->x, sizeof(double) * (size_t)e->n_c, hipMemcpyDeviceToHost, e->stream));
    if (B_index) HIPCHK(hipMemcpyAsync(B_index, e->B_index, sizeof(int64_t) * (size_t)e->m, hipMemcpyDeviceToHost, e->stream));
    if (N_index && e->nN) HIPCHK(hipMemcpyAsync(N_index, e->N_index, sizeof(int64_t) * (size_t)e->nN, hipMemcpyDeviceToHost, e->stream));
    if (N_bound && e->nN) HIPCHK(hipMemcpyAsync(N_bound, e->Nb, (size_t)e->nN, hipMemcpyDeviceToHost, e->stream));
    if (e->kind == ELLP_ENGINE_DUAL) {
        if (y) HIPCHK(hipMemcpyAsync(y, e->y, sizeof(double) * (size_t)e->m, hipMemcpyDeviceToHost, e->stream));
        if (d) HIPCHK(hipMemcpyAsync(d, e->dd, sizeof(double) * (size_t)e->n_c, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (was_open) {
        // completing the open iteration may have ended the solve (its ratio test found an unbounded ray, a panic of
        // the reference, a NaN): that status is reported here, with the point as it stands
        adopt_fin(e);
        const int32_t stt = e->h_st->status;
        if (stt != ST_RUNNING && stt != ST_NEED_MAINT && stt != ELLP_OPTIMAL) return status_message(*e->h_st, errbuf, errlen);
    }
    return ELLP_OPTIMAL;
}

int64_t ellp_engine_tap(ellp_engine *e, int what, double *dst, int64_t cap) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !dst) return ELLP_ERR_ARG;
    if (hipSetDevice(e->device) != hipSuccess) return ELLP_ERR_DEVICE;
    if (what != ELLP_TAP_STATE) launch_flush(e);
    const double *src = nullptr;
    int64_t count = 0;
    switch (what) {
    case ELLP_TAP_U:
        if (hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess)
            return ELLP_ERR_DEVICE;
        src = e->u + (e->h_st->usel ? e->ld : 0);
        count = e->m;
        break;
    case ELLP_TAP_R:
    case ELLP_TAP_ALPHA:
        if (e->world != 1) return ELLP_ERR_ARG;
        src = e->X + 2 * e->nbs + (int64_t)e->nbs * e->cpb;
        count = e->nN;
        break;
    case ELLP_TAP_D: src = e->d; count = e->m; break;
    case 7:  // steepest-edge weights by nonbasic position (diagnostics; valid up to the last pricing launch)
        if (!e->se) return ELLP_ERR_ARG;
        src = e->se_gamma;
        count = e->nN;
        break;
    case ELLP_TAP_KEY:
        if (e->world != 1) return ELLP_ERR_ARG;
        src = e->X + 2 * e->nbs;
        count = e->nN;
        break;
    case ELLP_TAP_STATE: {
        if (cap < 12) return ELLP_ERR_ARG;
        if (hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess)
            return ELLP_ERR_DEVICE;
        const DevState &h = *e->h_st;
        const double v[12] = {(double)h.status, (double)h.cur, (double)h.s_q, (double)h.s_r, h.s_theta_d, h.s_delta,
                              (double)h.lr, h.ldelta, (double)h.iters, (double)h.pivots, h.lambda, h.s_rq};
        for (int k = 0; k < 12; ++k) dst[k] = v[k];
        if (cap >= 14) {  // + last drift estimate of B^-1 (k_drift_reduce) and the number of checks so far
            dst[12] = h.drift;
            dst[13] = (double)e->drift_checks;
            if (cap >= 20) {
                dst[14] = (double)e->maint_requests;
                dst[15] = (double)e->refreshes;
                dst[16] = (double)e->refactors;
                dst[17] = (double)e->resyncs;
                dst[18] = h.resid;
                dst[19] = e->small ? 0.0 : ((e->lagged || e->dual_fused) ? 2.0 : 3.0);  // 0: whole iterations inside one persistent launch
                if (cap >= 22) {
                    dst[20] = (double)e->rebuild_shortcuts;
                    dst[21] = e->t_setup;
                    if (cap >= 28) {  // certified hybrid: on?, guarded pivots handed over, terminal statuses examined, of those not
                                      // confirmed, loop bodies run by the exact kernel, rebuilds of B^-1 after a hand-over
                        dst[22] = e->hybrid ? 1.0 : (e->cert_large ? 2.0 : 0.0);  // 2: terminal statuses certified by an exact-LU iteration (m > 1024)
                        dst[23] = (double)e->hy_guards;
                        dst[24] = (double)e->hy_certs;
                        dst[25] = (double)e->hy_disagree;
                        dst[26] = (double)e->hy_exact_iters;
                        dst[27] = (double)e->hy_rebuilds;
                        if (cap >= 29) {
                            dst[28] = (double)e->hy_redos;  // solves repeated by the exact kernel from the start of the phase
                            if (cap >= 30) {
                                // end points that went out NOT certified: a redo above 1,024 rows that would have cost more than
                                // ELLP_REDO_MAX_SECONDS, an exactly singular LU in the certificate, no memory for its factors
                                dst[29] = (double)e->hy_uncertified;
                                return 30;
                            }
                            return 29;
                        }
                        return 28;
                    }
                    return 22;
                }
                return 20;
            }
            return 14;
        }
        return 12;
    }
    case ELLP_TAP_BINV: {
        // row-major m x m without the padding
        count = e->m * e->m;
        if (count > cap) return ELLP_ERR_ARG;
        if (ensure_inverse(e, nullptr, 0) != ELLP_OPTIMAL) return ELLP_ERR_DEVICE;
        if (hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess)
            return ELLP_ERR_DEVICE;
        if (hipMemcpy2DAsync(dst, sizeof(double) * (size_t)e->m, e->h_st->cur ? e->W2 : e->W, sizeof(double) * (size_t)e->ld,
                             sizeof(double) * (size_t)e->m, (size_t)e->m, hipMemcpyDeviceToHost, e->stream) != hipSuccess)
            return ELLP_ERR_DEVICE;
        if (hipStreamSynchronize(e->stream) != hipSuccess) return ELLP_ERR_DEVICE;
        return count;
    }
    default: return ELLP_ERR_ARG;
    }
    if (count > cap) count = cap;
    if (count > 0) {
        if (hipMemcpyAsync(dst, src, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, e->stream) != hipSuccess)
            return ELLP_ERR_DEVICE;
        if (hipStreamSynchronize(e->stream) != hipSuccess) return ELLP_ERR_DEVICE;
    }
    return count;
}

int64_t ellp_engine_read_trace(ellp_engine *e, uint64_t *iters_out, double *obj_out, int64_t cap) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !iters_out || !obj_out || cap < 0) return ELLP_ERR_ARG;
    if (e->trace_len <= 0) return 0;
    if (hipSetDevice(e->device) != hipSuccess) return ELLP_ERR_DEVICE;
    launch_flush(e);
    std::vector<unsigned long long> it((size_t)e->trace_len);
    std::vector<double> ob((size_t)e->trace_len);
    if (hipMemcpyAsync(it.data(), e->trace_it, sizeof(unsigned long long) * it.size(), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipMemcpyAsync(ob.data(), e->trace_obj, sizeof(double) * ob.size(), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
        return ELLP_ERR_DEVICE;
    // entries in iteration order: the ring position of iteration k is k % len; 0 marks an empty slot
    unsigned long long newest = 0;
    for (auto v : it) newest = v > newest ? v : newest;
    int64_t n = 0;
    const unsigned long long len = (unsigned long long)e->trace_len;
    const unsigned long long first = newest >= len ? newest - len + 1 : 1;
    for (unsigned long long k = first; k <= newest && n < cap; ++k) {
        const size_t slot = (size_t)(k % len);
        if (it[slot] != k) continue;  // an iteration that completed nothing (dropped for maintenance)
        iters_out[n] = k;
        obj_out[n] = ob[slot];
        ++n;
    }
    return n;
}

ellp_status ellp_engine_request_maintenance(ellp_engine *e) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    if (hipSetDevice(e->device) != hipSuccess) return ELLP_ERR_DEVICE;
    if (e->small) return ELLP_OPTIMAL;  // k_small carries no inverse: every iteration starts from a fresh LU
    const int32_t one = 1;
    if (hipMemcpyAsync(&e->st->tiny, &one, sizeof(int32_t), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
        return ELLP_ERR_DEVICE;
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_debug_scale_inverse(ellp_engine *e, double factor) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    if (hipSetDevice(e->device) != hipSuccess) return ELLP_ERR_DEVICE;
    if (ensure_inverse(e, nullptr, 0) != ELLP_OPTIMAL) return ELLP_ERR_DEVICE;
    hipLaunchKernelGGL(k_scale_inverse, dim3(512), dim3(256), 0, e->stream, e->W, e->W2, e->st, e->m * e->ld, factor);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return ELLP_ERR_DEVICE;
    e->u_valid = false;
    return ELLP_OPTIMAL;
}

double ellp_engine_inverse_residual(ellp_engine *e) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return NAN;
    if (hipSetDevice(e->device) != hipSuccess) return NAN;
    if (ensure_inverse(e, nullptr, 0) != ELLP_OPTIMAL) return NAN;
    hipLaunchKernelGGL(k_inv_residual, dim3((unsigned)e->m), dim3(256), 0, e->stream, e->W, e->W2, e->st, e->A_B,
                       e->m, e->ld, e->resid);
    std::vector<double> h((size_t)e->m);
    if (hipMemcpyAsync(h.data(), e->resid, sizeof(double) * (size_t)e->m, hipMemcpyDeviceToHost, e->stream) != hipSuccess)
        return NAN;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return NAN;
    double w = 0.0;
    for (double v : h) w = (v > w || v != v) ? v : w;
    return w;
}

ellp_status ellp_engine_set_shard(ellp_engine *e, int rank, int world, void *exchange_buffer, char *errbuf,
                                  size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || world < 1 || rank < 0 || rank >= world) {
        set_err(errbuf, errlen, "bad rank/world");
        return ELLP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    {
        const ellp_status si = ensure_inverse(e, errbuf, errlen);  // the stepped loop is the explicit-inverse engine's
        if (si != ELLP_OPTIMAL) return si;
    }
    if (e->lagged) {  // the stepped loop drives the three-launch kernels, which use u buffer 0
        e->lagged = false;
        e->lag_open = false;
        const int32_t zero = 0;
        HIPCHK(hipMemcpy(&e->st->usel, &zero, sizeof(int32_t), hipMemcpyHostToDevice));
        e->u_valid = false;
    }
    e->rank = rank;
    e->world = world;
    if (world > 1) e->hybrid = e->cert_large = false;  // certification is for unsharded engines
    e->nbs = (e->nblocks + world - 1) / world;
    e->seg = 2 * (int64_t)e->nbs + 2 * (int64_t)e->nbs * e->cpb;
    double *nx = static_cast<double *>(exchange_buffer);  // caller-owned (e.g. a torch tensor) ...
    if (!nx) HIPCHK(dmalloc(e, &nx, (size_t)(e->seg * world)));  // ... or ours (released with the engine)
    HIPCHK(hipMemset(nx, 0, sizeof(double) * (size_t)(e->seg * world)));
    e->X = nx;
    return ELLP_OPTIMAL;
}

int64_t ellp_engine_segment_doubles(ellp_engine *e, int world) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || world < 1) return ELLP_ERR_ARG;
    const int64_t nbs = (e->nblocks + world - 1) / world;
    return 2 * nbs + 2 * nbs * e->cpb;
}

ellp_status ellp_engine_exchange_info(ellp_engine *e, void **base, int64_t *seg_doubles, int *rank, int *world) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    if (base) *base = e->X;
    if (seg_doubles) *seg_doubles = e->seg;
    if (rank) *rank = e->rank;
    if (world) *world = e->world;
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_set_stream(ellp_engine *e, void *hip_stream) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    if (hipSetDevice(e->device) != hipSuccess) return ELLP_ERR_DEVICE;
    (void)hipStreamSynchronize(e->stream);
    e->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->own_stream;
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_step(ellp_engine *e, int phase, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    if (phase == 2) {  // second half of this iteration + first half of the next, one host call
        const ellp_status s1 = ellp_engine_step(e, 1, errbuf, errlen);
        if (s1 != ELLP_OPTIMAL) return s1;
        return ellp_engine_step(e, 0, errbuf, errlen);
    }
    HIPCHK(hipSetDevice(e->device));
    if (e->nN == 0) return ELLP_OPTIMAL;
    e->hybrid = e->cert_large = false;  // the stepped API drives the plain explicit-inverse engine (no guard: nobody would service its stop)
    if (e->small || !e->w_valid) {
        const ellp_status si = ensure_inverse(e, errbuf, errlen);
        if (si != ELLP_OPTIMAL) return si;
    }
    if (phase == 0) {
        int64_t period = e->refactor_period > 0 ? e->refactor_period : default_period(e);
        if (e->since_refactor >= (uint64_t)period) maintain_inverse(e);
        // the follow-up of a serviced maintenance request (see ellp_engine_run): B^-1 is refreshed once
        // more after the single iteration that follows the request.  maint_chain: 1 = that iteration
        // starts now, 2 = it has run
        if (e->maint_chain == 2) {
            maintain_inverse(e, true);
            e->maint_chain = 0;
        } else if (e->maint_chain == 1) {
            e->maint_chain = 2;
        }
        if (e->kind == ELLP_ENGINE_PRIMAL) {
            const bool full_btran =
                (e->opts.btran_mode == 1) || !e->u_valid || e->since_btran >= (uint64_t)e->btran_refresh;
            if (full_btran) {
                Prof p(e, ELLP_K_BTRAN);
                launch_btran(e);
                e->since_btran = 0;
                e->u_valid = true;
            }
            Prof p(e, ELLP_K_PRICE);
            launch_price<0>(e);
        } else {
            if (e->need_dleave) {
                launch_dleave(e);
                e->need_dleave = false;
            }
            Prof p(e, ELLP_K_DPRICE);
            launch_price<1>(e);
        }
    } else {
        if (e->kind == ELLP_ENGINE_PRIMAL) {
            {
                Prof p(e, ELLP_K_FTRAN);
                launch_ftran2<0>(e);
            }
            launch_drift_check(e);
            Prof p(e, ELLP_K_UPDATE);
            launch_update2<0>(e, e->opts.btran_mode == 1 ? 0 : 1);
            e->since_btran += 1;
        } else {
            {
                Prof p(e, ELLP_K_FTRAN);
                launch_ftran2<1>(e);
            }
            launch_drift_check(e);
            Prof p(e, ELLP_K_DUPDATE);
            launch_update2<1>(e, 0);
        }
        e->since_refactor += 1;
        e->enqueued += 1;
    }
    HIPCHK(hipGetLastError());
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_poll(ellp_engine *e, ellp_stats *stats, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    prof_collect(e);
    reconcile_counters(e);
    (void)service_maintenance_request(e);
    fill_stats(e, stats);
    if (e->h_st->status == ST_RUNNING) return ELLP_MAXITER;  // still running: the slice is simply used up
    return status_message(*e->h_st, errbuf, errlen);
}

ellp_status ellp_engine_rephase(ellp_engine *e, const double *c, const uint8_t *bound_kind, const double *lb,
                                const double *ub, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !c || !bound_kind || !lb || !ub) return ELLP_ERR_ARG;
    if (errbuf && errlen) errbuf[0] = 0;
    if (e->kind != ELLP_ENGINE_PRIMAL) {
        set_err(errbuf, errlen, "rephase is the primal phase-1 -> phase-2 hand-off");
        return ELLP_ERR_ARG;
    }
    for (int64_t i = 0; i < e->n_c; ++i)
        if (bound_kind[i] > 4) {
            set_err(errbuf, errlen, "bound_kind[%lld] out of range", (long long)i);
            return ELLP_ERR_ARG;
        }
    HIPCHK(hipSetDevice(e->device));
    launch_flush(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    double *c_dev = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c_dev), sizeof(double) * (size_t)e->n_c));
    auto bail = [&](hipError_t err) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipFree(c_dev);
        set_err(errbuf, errlen, "HIP error %s in ellp_engine_rephase", hipGetErrorString(err));
        return ELLP_ERR_DEVICE;
    };
    hipError_t rc;
    if ((rc = hipMemcpyAsync(c_dev, c, sizeof(double) * (size_t)e->n_c, hipMemcpyHostToDevice, e->stream)) != hipSuccess) return bail(rc);
    if ((rc = hipMemcpyAsync(e->lb, lb, sizeof(double) * (size_t)e->n_c, hipMemcpyHostToDevice, e->stream)) != hipSuccess) return bail(rc);
    if ((rc = hipMemcpyAsync(e->ub, ub, sizeof(double) * (size_t)e->n_c, hipMemcpyHostToDevice, e->stream)) != hipSuccess) return bail(rc);
    if ((rc = hipMemcpyAsync(e->kindv, bound_kind, (size_t)e->n_c, hipMemcpyHostToDevice, e->stream)) != hipSuccess) return bail(rc);
    const int64_t cnt = e->m > e->nN ? e->m : e->nN;
    hipLaunchKernelGGL(k_rephase, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream, c_dev, e->kindv,
                       e->B_index, e->N_index, e->c_B, e->c_N, e->Nb, e->m, e->nN);
    hipLaunchKernelGGL(k_primal_obj, dim3(1), dim3(1024), 0, e->stream, e->c_B, e->c_N, e->x, e->B_index, e->N_index, e->m,
                       e->nN, e->st);  // c.x with the new costs (the objective trace continues from it)
    if (e->trace_len > 0) (void)hipMemsetAsync(e->trace_it, 0, sizeof(unsigned long long) * (size_t)e->trace_len, e->stream);
    // a new solve_with_initial starts here: status, counters and flags afresh; B^-1 and `cur` stay
    if ((rc = hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream)) != hipSuccess) return bail(rc);
    if ((rc = hipStreamSynchronize(e->stream)) != hipSuccess) return bail(rc);
    DevState ns = *e->h_st;
    ns.status = ST_RUNNING;
    ns.nan_flag = 0;
    ns.tiny = 0;
    ns.tiny_p = 0;
    ns.fin = 0;
    ns.pe_valid = 0;
    ns.open = 0;
    ns.se_valid = 0;  // steepest edge: the last pivot's weight update has been applied by the pricing launch that ended the phase
    ns.mv_pending = 0;
    ns.pp_seg = 0;  // partial pricing starts over with the first segment
    ns.pp_empty = 0;
    ns.pp_skip = 0;
    ns.pp_lo = 0;
    ns.pp_hi = e->pp_P > 1 ? e->pp_S : e->nN;
    ns.need_rebuild = 0;
    ns.panic_code = 0;
    ns.iters = ns.pivots = ns.flips = 0;
    ns.lambda = 0.0;
    *e->h_st = ns;
    if ((rc = hipMemcpyAsync(e->st, e->h_st, sizeof(DevState), hipMemcpyHostToDevice, e->stream)) != hipSuccess) return bail(rc);
    if ((rc = hipStreamSynchronize(e->stream)) != hipSuccess) return bail(rc);
    (void)hipFree(c_dev);
    e->u_valid = false;  // u = B^-T c_B with the new costs
    e->maint_chain = 0;
    e->enqueued = 0;
    e->iters_seen = 0;
    e->snap.valid = false;  // a new phase starts here
    e->exact_large_only = false;  // ... in the fast loop again
    return ELLP_OPTIMAL;
}

// y = B^-T c_B, d = c - A^T y, labels and values of the nonbasic variables, x_B = B^-1 (b - A_N x_N), all from the
// resident B^-1: the common part of DualPhase2::from (dual_problem.rs:278-328; labels :286-323) and of
// DualPhase1::new (:165-214; labels :177-203, `phase1`).  Ends with the loop's own entry assertion
// (dual_simplex_solver.rs:139-151).  c_dev: the costs by variable index, on the device.
static ellp_status dual_point_from_inverse(ellp_engine *e, const double *c_dev, int phase1, const int64_t *N_host,
                                           char *errbuf, size_t errlen) {
    const int64_t m = e->m, nN = e->nN, ld = e->ld, n_c = e->n_c;
    // Certified-hybrid engines take y (and, below, x_B) from a fresh LU of the basis — the LU-per-iteration kernel with zero
    // iterations — as the reference does (dual_problem.rs:162-172, :275-284), not from the explicit inverse: the point is the
    // START of a phase whose carried d the caller will test against EPS at its end, and a d that starts 1e-9 off stays 1e-9 off
    // even when the exact kernel repeats the phase.
    const bool from_lu = e->hybrid && e->LUa != nullptr && e->world == 1;
    if (from_lu) {
        HIPCHK(launch_mid(e, 0, 2));
    } else {
        launch_btran(e);  // into e->u (a dual engine has no other use for it)
        HIPCHK(hipMemcpyAsync(e->y, e->u, sizeof(double) * (size_t)ld, hipMemcpyDeviceToDevice, e->stream));
    }
    HIPCHK(hipMemsetAsync(e->x, 0, sizeof(double) * (size_t)n_c, e->stream));
    DualRephaseArgs da{e->A_N, e->A_B, e->y, c_dev, e->kindv, e->lb, e->ub, e->N_index, e->B_index, e->dd, e->x, e->Nb,
                       e->st, m, ld, nN, e->eps, phase1};
    hipLaunchKernelGGL(k_dual_rephase, dim3((unsigned)((nN + m + 3) / 4)), dim3(256), 0, e->stream, da);
    launch_resync(e, 1);
    if (from_lu) HIPCHK(launch_mid(e, 0, 1));  // x_B = A_B^-1 (b - A_N x_N) from the LU
    std::vector<double> d((size_t)n_c);
    std::vector<uint8_t> Nb((size_t)(nN > 0 ? nN : 1));
    HIPCHK(hipMemcpyAsync(d.data(), e->dd, sizeof(double) * (size_t)n_c, hipMemcpyDeviceToHost, e->stream));
    if (nN > 0) HIPCHK(hipMemcpyAsync(Nb.data(), e->Nb, (size_t)nN, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    if (e->h_st->status != ST_RUNNING) return status_message(*e->h_st, errbuf, errlen);
    for (int64_t j = 0; j < nN; ++j) {
        const double di = d[(size_t)N_host[(size_t)j]];
        bool infeasible;
        if (Nb[(size_t)j] == ELLP_NB_LOWER) infeasible = di < -e->eps;
        else if (Nb[(size_t)j] == ELLP_NB_UPPER) infeasible = di > e->eps;
        else infeasible = std::fabs(di) > e->eps;
        if (infeasible) {
            set_err(errbuf, errlen, "initial point of dual phase 2 is dual infeasible");
            return ELLP_ERR_PANIC;
        }
    }
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_dual_rephase(ellp_engine *e, const double *c, const double *b, const uint8_t *bound_kind,
                                     const double *lb, const double *ub, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !c || !b || !bound_kind || !lb || !ub) return ELLP_ERR_ARG;
    if (errbuf && errlen) errbuf[0] = 0;
    if (e->kind != ELLP_ENGINE_DUAL || e->small || e->colshard || e->world != 1) {
        set_err(errbuf, errlen, "dual_rephase: a resident, unsharded dual engine of the explicit-inverse kind is needed");
        return ELLP_ERR_ARG;
    }
    for (int64_t i = 0; i < e->n_c; ++i)
        if (bound_kind[i] > 4) {
            set_err(errbuf, errlen, "bound_kind[%lld] out of range", (long long)i);
            return ELLP_ERR_ARG;
        }
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int64_t m = e->m, nN = e->nN, ld = e->ld, n_c = e->n_c;
    // ---- N in variable-index order (dual_problem.rs:286-323 enumerates `is_basic`), columns moved along
    std::vector<int64_t> N((size_t)nN), perm((size_t)nN), Nsorted((size_t)nN);
    if (nN > 0) HIPCHK(hipMemcpy(N.data(), e->N_index, sizeof(int64_t) * (size_t)nN, hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < nN; ++j) perm[(size_t)j] = j;
    std::sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b2) { return N[(size_t)a] < N[(size_t)b2]; });
    bool moved = false;
    for (int64_t j = 0; j < nN; ++j) {
        Nsorted[(size_t)j] = N[(size_t)perm[(size_t)j]];
        moved = moved || perm[(size_t)j] != j;
    }
    double *c_dev = nullptr, *A2 = nullptr;
    int64_t *perm_dev = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(e->stream);
        if (c_dev) (void)hipFree(c_dev);
        if (perm_dev) (void)hipFree(perm_dev);
    };
#define DCHK(expr)                                                                                        \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            cleanup();                                                                                    \
            set_err(errbuf, errlen, "HIP error %s in ellp_engine_dual_rephase (%s)", hipGetErrorString(_e), #expr); \
            return ELLP_ERR_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)
    if (moved && nN > 0) {
        DCHK(hipMalloc(reinterpret_cast<void **>(&A2), sizeof(double) * (size_t)(ld * nN)));
        DCHK(hipMalloc(reinterpret_cast<void **>(&perm_dev), sizeof(int64_t) * (size_t)nN));
        DCHK(hipMemcpyAsync(perm_dev, perm.data(), sizeof(int64_t) * (size_t)nN, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)nN), dim3(256), 0, e->stream, e->A_N, A2, perm_dev, ld);
        DCHK(hipMemcpyAsync(e->N_index, Nsorted.data(), sizeof(int64_t) * (size_t)nN, hipMemcpyHostToDevice, e->stream));
        DCHK(hipStreamSynchronize(e->stream));
        replace_alloc(e, e->A_N, A2);
        e->A_N = A2;
    }
    // ---- new costs, right-hand side, bounds
    DCHK(hipMalloc(reinterpret_cast<void **>(&c_dev), sizeof(double) * (size_t)n_c));
    DCHK(hipMemcpyAsync(c_dev, c, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
    DCHK(hipMemcpyAsync(e->b_dev, b, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, e->stream));
    DCHK(hipMemcpyAsync(e->lb, lb, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
    DCHK(hipMemcpyAsync(e->ub, ub, sizeof(double) * (size_t)n_c, hipMemcpyHostToDevice, e->stream));
    DCHK(hipMemcpyAsync(e->kindv, bound_kind, (size_t)n_c, hipMemcpyHostToDevice, e->stream));
    const int64_t cnt = m > nN ? m : nN;
    hipLaunchKernelGGL(k_rephase, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream, c_dev, e->kindv, e->B_index,
                       e->N_index, e->c_B, e->c_N, e->Nb, m, nN);
    // a new solve_with_initial: status and counters afresh (the maintenance kernels need a RUNNING engine)
    DCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    DCHK(hipStreamSynchronize(e->stream));
    DevState ns = *e->h_st;
    ns.status = ST_RUNNING;
    ns.nan_flag = 0;
    ns.tiny = 0;
    ns.need_rebuild = 0;
    ns.panic_code = 0;
    ns.iters = ns.pivots = ns.flips = 0;
    ns.lr = -1;
    *e->h_st = ns;
    DCHK(hipMemcpyAsync(e->st, e->h_st, sizeof(DevState), hipMemcpyHostToDevice, e->stream));
    // y, d and the labels are made from B^-1 and checked against the reference's EPS assertions (dual_problem.rs:275-323);
    // the reference takes a FRESH LU there, so the resident inverse (up to a maintenance period of eta updates old) is
    // rebuilt from A_B first — once per solve
    launch_dual_close(e);
    launch_refactor(e);
    DCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    DCHK(hipStreamSynchronize(e->stream));
    if (e->h_st->status != ST_RUNNING) {
        cleanup();
        return status_message(*e->h_st, errbuf, errlen);
    }
    e->since_refactor = 0;
    const ellp_status ps = dual_point_from_inverse(e, c_dev, 0, Nsorted.data(), errbuf, errlen);
    cleanup();
#undef DCHK
    if (ps != ELLP_OPTIMAL) return ps;
    std::vector<double> y((size_t)m), d((size_t)n_c);
    HIPCHK(hipMemcpy(y.data(), e->y, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(d.data(), e->dd, sizeof(double) * (size_t)n_c, hipMemcpyDeviceToHost));
    const double obj = host_dual_obj(m, n_c, b, bound_kind, lb, ub, y.data(), d.data());
    HIPCHK(hipMemcpy(&e->st->obj, &obj, sizeof(double), hipMemcpyHostToDevice));
    e->h_st->obj = obj;
    e->need_dleave = true;
    e->maint_chain = 0;
    e->enqueued = 0;
    e->iters_seen = 0;
    e->snap.valid = false;  // a new phase starts here
    e->exact_large_only = false;  // ... in the fast loop again
    {
        bool box = true;
        for (int64_t i = 0; i < n_c && box; ++i) box = bound_kind[i] == ELLP_BOUND_TWOSIDED || bound_kind[i] == ELLP_BOUND_FIXED;
        for (int64_t i = 0; i < m && box; ++i) box = b[i] == 0.0;
        e->box_problem = box;
    }
    if (e->trace_len > 0) HIPCHK(hipMemset(e->trace_it, 0, sizeof(unsigned long long) * (size_t)e->trace_len));
    return ELLP_OPTIMAL;
}

// DualPhase1::new's point (dual_problem.rs:162-214) made on the device: the caller has the box problem's standard
// form and the basis the LU of A^T picked (B_index, and N_index in the order of the permutation, :153-160); the
// engine builds B^-1 and from it y = B^-T c_B, d = c - A^T y, the nonbasic labels and values by the sign of d,
// b~ = b - A x and x_B = B^-1 b~.  No LU of A_B on the host, no solves, no A x product there.
ellp_status ellp_engine_create_dual_phase1(int64_t m, int64_t n, const double *A, const double *c, const double *b,
                                           const uint8_t *bound_kind, const double *lb, const double *ub,
                                           const int64_t *B_index, const int64_t *N_index, const ellp_opts *opts_in,
                                           ellp_engine **out, char *errbuf, size_t errlen) {
    if (errbuf && errlen) errbuf[0] = 0;
    if (!out) return ELLP_ERR_ARG;
    *out = nullptr;
    if (m <= 0 || n <= m || !A || !c || !b || !bound_kind || !lb || !ub || !B_index || !N_index) {
        set_err(errbuf, errlen, "bad arguments (a nonbasic variable is needed: n > m)");
        return ELLP_ERR_ARG;
    }
    const int64_t nN = n - m;
    std::vector<double> zeros((size_t)(n > m ? n : m), 0.0);
    std::vector<uint8_t> Nb((size_t)(nN > 0 ? nN : 1), (uint8_t)ELLP_NB_LOWER);
    ellp_engine *e = nullptr;
    ellp_status s = engine_create_impl(ELLP_ENGINE_DUAL, m, n, n, A, c, b, bound_kind, lb, ub, zeros.data(), B_index, m,
                                       N_index, Nb.data(), nN, zeros.data(), zeros.data(), opts_in, &e, errbuf, errlen, n,
                                       false);
    if (s != ELLP_OPTIMAL) return s;
    auto fail = [&](ellp_status st) {
        ellp_engine_destroy(e);
        return st;
    };
    if (e->small) {  // k_small keeps no inverse; the point below is made from one all the same
        launch_refactor(e);
        if (hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess) {
            set_err(errbuf, errlen, "HIP error in ellp_engine_create_dual_phase1");
            return fail(ELLP_ERR_DEVICE);
        }
        if (e->h_st->status != ST_RUNNING) return fail(status_message(*e->h_st, errbuf, errlen));
    }
    double *c_dev = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&c_dev), sizeof(double) * (size_t)n) != hipSuccess ||
        hipMemcpyAsync(c_dev, c, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, e->stream) != hipSuccess) {
        if (c_dev) (void)hipFree(c_dev);
        set_err(errbuf, errlen, "HIP error in ellp_engine_create_dual_phase1");
        return fail(ELLP_ERR_DEVICE);
    }
    s = dual_point_from_inverse(e, c_dev, 1, N_index, errbuf, errlen);
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(c_dev);
    if (s != ELLP_OPTIMAL) return fail(s);
    std::vector<double> y((size_t)m), d((size_t)n);
    if (hipMemcpy(y.data(), e->y, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(d.data(), e->dd, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
        set_err(errbuf, errlen, "HIP error in ellp_engine_create_dual_phase1");
        return fail(ELLP_ERR_DEVICE);
    }
    const double obj = host_dual_obj(m, n, b, bound_kind, lb, ub, y.data(), d.data());
    if (hipMemcpy(&e->st->obj, &obj, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(ELLP_ERR_DEVICE);
    e->h_st->obj = obj;
    e->need_dleave = true;
    e->u_valid = false;
    *out = e;
    return ELLP_OPTIMAL;
}

// ---- direct RCCL exchange -------------------------------------------------------------------
static const RcclApi *load_rccl(const char *path, char *errbuf, size_t errlen) {
    // bound once per process; the initialisation of a function-local static is thread-safe, and
    // dlopen of a library the process already has is a lookup
    static const RcclApi api = [path] {
        RcclApi a;
        const char *names[] = {path, "librccl.so.1", "librccl.so"};
        for (const char *nm : names) {
            if (!nm || !nm[0]) continue;
            a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (a.handle) break;
        }
        if (a.handle) {
            a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
            a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
            a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
            a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
            a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
        }
        return a;
    }();
    if (!(api.handle && api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy && api.GetErrorString)) {
        set_err(errbuf, errlen, "RCCL is not available (dlopen/dlsym of librccl failed)");
        return nullptr;
    }
    return &api;
}

ellp_status ellp_comm_unique_id(const char *rccl_path, void *id_out, char *errbuf, size_t errlen) {
    if (!id_out) return ELLP_ERR_ARG;
    const RcclApi *api = load_rccl(rccl_path, errbuf, errlen);
    if (!api) return ELLP_ERR_DEVICE;
    ncclUniqueId id;
    const ncclResult_t rc = api->GetUniqueId(&id);
    if (rc != ncclSuccess) {
        set_err(errbuf, errlen, "ncclGetUniqueId: %s", api->GetErrorString(rc));
        return ELLP_ERR_DEVICE;
    }
    static_assert(sizeof(ncclUniqueId) == ELLP_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_comm_init(ellp_engine *e, const char *rccl_path, const void *id, int rank, int world,
                                  char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !id || world < 1 || rank < 0 || rank >= world) return ELLP_ERR_ARG;
    const RcclApi *api = load_rccl(rccl_path, errbuf, errlen);
    if (!api) return ELLP_ERR_DEVICE;
    HIPCHK(hipSetDevice(e->device));
    if (!e->colshard) {
        const ellp_status s = ellp_engine_set_shard(e, rank, world, nullptr, errbuf, errlen);  // engine-owned buffer
        if (s != ELLP_OPTIMAL) return s;
    } else if (rank != e->rank || world != e->world) {
        set_err(errbuf, errlen, "rank/world differ from ellp_engine_shard_columns");
        return ELLP_ERR_ARG;
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    if (e->comm) {
        (void)api->CommDestroy(e->comm);
        e->comm = nullptr;
    }
    const ncclResult_t rc = api->CommInitRank(&e->comm, world, uid, rank);
    if (rc != ncclSuccess) {
        set_err(errbuf, errlen, "ncclCommInitRank: %s", api->GetErrorString(rc));
        e->comm = nullptr;
        return ELLP_ERR_DEVICE;
    }
    e->rccl = api;
    if (e->colshard && e->transport == 0) e->transport = 1;
    return ELLP_OPTIMAL;
}

// ---- column-sharded storage: setup, transports, loop (kernels: ellp_shard.inc) ---------------------
ellp_status ellp_engine_shard_columns(ellp_engine *e, int rank, int world, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || world < 1 || rank < 0 || rank >= world) return ELLP_ERR_ARG;
    if (errbuf && errlen) errbuf[0] = 0;
    if (e->kind != ELLP_ENGINE_PRIMAL) {
        set_err(errbuf, errlen, "column-sharded storage is implemented for the primal loop");
        return ELLP_ERR_ARG;
    }
    if (world * SH_KC > WAVE) {
        set_err(errbuf, errlen, "at most %d ranks", WAVE / SH_KC);
        return ELLP_ERR_ARG;
    }
    if (e->pp_P > 1) {
        set_err(errbuf, errlen, "partial pricing is implemented for one GPU");
        return ELLP_ERR_ARG;
    }
    if (e->se) {
        set_err(errbuf, errlen, "steepest-edge pricing is implemented for one GPU");
        return ELLP_ERR_ARG;
    }
    if (e->colshard) {
        set_err(errbuf, errlen, "the columns are already sharded");
        return ELLP_ERR_ARG;
    }
    // the two-launch pipeline stays on for a column-sharded engine (ELLP_SHARD_LAGGED=0: the three-launch kernels of rounds 1-2)
    const bool keep_lagged = e->lagged && !(getenv("ELLP_SHARD_LAGGED") && getenv("ELLP_SHARD_LAGGED")[0] == '0') &&
                             !(getenv("ELLP_SHARD_SPLIT") && getenv("ELLP_SHARD_SPLIT")[0] == '1');
    const ellp_status s0 = ellp_engine_set_shard(e, rank, world, nullptr, errbuf, errlen);  // rank, world, nbs, seg, X
    if (s0 != ELLP_OPTIMAL) return s0;
    e->hybrid = e->cert_large = false;  // certification is for unsharded engines
    if (keep_lagged) {
        e->lagged = true;
        e->lag_open = false;
    }
    HIPCHK(hipSetDevice(e->device));
    const int64_t ld = e->ld;
    int64_t a0 = (int64_t)rank * e->nbs * e->cpb, a1 = (int64_t)(rank + 1) * e->nbs * e->cpb;
    if (a0 > e->nN) a0 = e->nN;
    if (a1 > e->nN) a1 = e->nN;
    const int64_t nloc = a1 - a0;
    double *loc = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&loc), sizeof(double) * (size_t)(ld * (nloc > 0 ? nloc : 1))));
    if (nloc > 0)
        HIPCHK(hipMemcpy(loc, e->A_N + a0 * ld, sizeof(double) * (size_t)(ld * nloc), hipMemcpyDeviceToDevice));
    // release the full A_N: it is one of e->allocs
    replace_alloc(e, e->A_N, loc);
    e->A_N_store = loc;
    e->A_N = loc - a0 * ld;  // virtual base: only [own0, own1) may be dereferenced
    e->own0 = a0;
    e->own1 = a1;
    const int64_t pd = pack_doubles(ld);
    e->slot_doubles = pd > e->seg ? pd : e->seg;
    HIPCHK(dmalloc(e, &e->packs, (size_t)(pd * world)));
    HIPCHK(dmalloc(e, &e->aq_cur, (size_t)ld));
    HIPCHK(hipMemset(e->packs, 0, sizeof(double) * (size_t)(pd * world)));
    HIPCHK(hipMemset(e->aq_cur, 0, sizeof(double) * (size_t)ld));
    e->colshard = true;
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_set_exchange_callback(ellp_engine *e, ellp_exchange_fn fn, void *user) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    e->xfn = fn;
    e->xuser = user;
    if (fn) e->transport = 3;
    return ELLP_OPTIMAL;
}

int ellp_shard_select_compact(const double *packs, int world, int64_t ld, double eps, int64_t *q, int *src_rank, int *src_slot) {
    if (!packs || !q || !src_rank || !src_slot || world < 1 || world * SH_KC > WAVE) return -1;
    long long qq = -1;
    const int v = shard_select_compact(packs, world, ld, eps, &qq, src_rank, src_slot);
    *q = qq;
    return v;
}
int64_t ellp_shard_pack_doubles(int64_t ld) { return pack_doubles(ld); }

ellp_status ellp_engine_shard_info(ellp_engine *e, double *out6) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !out6) return ELLP_ERR_ARG;
    out6[0] = (double)e->full_exchanges;
    out6[1] = (double)e->column_requests;
    out6[2] = (double)e->transport;
    out6[3] = (double)pack_doubles(e->ld);
    out6[4] = (double)e->own0;
    out6[5] = (double)e->own1;
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_mailbox_export(ellp_engine *e, void *handles_out, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !handles_out || !e->colshard) return ELLP_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    static_assert(sizeof(hipIpcMemHandle_t) == ELLP_IPC_HANDLE_BYTES, "IPC handle size");
    if (!e->mbox) {
        const size_t slots = sizeof(double) * (size_t)(2 * e->world * e->slot_doubles);
        void *p = nullptr, *f = nullptr;
        // uncached: stores of a peer must not be hidden by this GPU's L2, polls must not hit stale lines
        HIPCHK(hipExtMallocWithFlags(&p, slots, hipDeviceMallocUncached));
        HIPCHK(hipExtMallocWithFlags(&f, sizeof(unsigned long long) * 2 * 64, hipDeviceMallocUncached));
        HIPCHK(hipMemset(p, 0, slots));
        HIPCHK(hipMemset(f, 0, sizeof(unsigned long long) * 2 * 64));
        e->mbox = static_cast<double *>(p);
        e->mflags = static_cast<unsigned long long *>(f);
    }
    hipIpcMemHandle_t h[2];
    HIPCHK(hipIpcGetMemHandle(&h[0], e->mbox));
    HIPCHK(hipIpcGetMemHandle(&h[1], e->mflags));
    memcpy(handles_out, h, sizeof(h));
    return ELLP_OPTIMAL;
}

ellp_status ellp_engine_mailbox_connect(ellp_engine *e, const void *all_handles, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || !all_handles || !e->mbox) return ELLP_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    std::vector<double *> slots((size_t)e->world);
    std::vector<unsigned long long *> flags((size_t)e->world);
    const hipIpcMemHandle_t *h = static_cast<const hipIpcMemHandle_t *>(all_handles);
    for (int r = 0; r < e->world; ++r) {
        if (r == e->rank) {
            slots[r] = e->mbox;
            flags[r] = e->mflags;
            continue;
        }
        void *p = nullptr, *f = nullptr;
        HIPCHK(hipIpcOpenMemHandle(&p, h[2 * r], hipIpcMemLazyEnablePeerAccess));
        e->ipc_opened.push_back(p);
        HIPCHK(hipIpcOpenMemHandle(&f, h[2 * r + 1], hipIpcMemLazyEnablePeerAccess));
        e->ipc_opened.push_back(f);
        slots[r] = static_cast<double *>(p);
        flags[r] = static_cast<unsigned long long *>(f);
    }
    HIPCHK(dmalloc(e, &e->d_peer_slots, (size_t)e->world));
    HIPCHK(dmalloc(e, &e->d_peer_flags, (size_t)e->world));
    HIPCHK(hipMemcpy(e->d_peer_slots, slots.data(), sizeof(double *) * (size_t)e->world, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_peer_flags, flags.data(), sizeof(unsigned long long *) * (size_t)e->world, hipMemcpyHostToDevice));
    e->transport = 2;
    return ELLP_OPTIMAL;
}

namespace {
// all-gather `n` doubles per rank: src = this rank's segment, dst = world * n doubles (dst + rank * n may be src)
ellp_status shard_exchange(ellp_engine *e, const double *src, double *dst, int64_t n, char *errbuf, size_t errlen, bool commit = true) {
    if (e->world == 1) {
        if (dst + (int64_t)e->rank * n != src)
            HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, e->stream));
        return ELLP_OPTIMAL;
    }
    if (e->transport == 2) {
        MboxArgs a{e->d_peer_slots, e->d_peer_flags, src, dst, e->st, n, e->rank, e->world, 300000000LL /* 3 s */};
        // the slot stride is slot_doubles, not n: both sides use slot_doubles
        a.n = n;
        hipLaunchKernelGGL(k_mbox_push, dim3((unsigned)e->world), dim3(256), 0, e->stream, a, e->slot_doubles);
        hipLaunchKernelGGL(k_mbox_wait, dim3((unsigned)e->world), dim3(256), 0, e->stream, a, e->slot_doubles);
        if (commit) hipLaunchKernelGGL(k_mbox_commit, dim3(1), dim3(1), 0, e->stream, e->st);
        return ELLP_OPTIMAL;
    }
    if (e->transport == 1 && e->comm && e->rccl) {
        if (dst + (int64_t)e->rank * n != src)
            HIPCHK(hipMemcpyAsync(dst + (int64_t)e->rank * n, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, e->stream));
        const ncclResult_t rc = e->rccl->AllGather(dst + (int64_t)e->rank * n, dst, (size_t)n, ncclDouble, e->comm, e->stream);
        if (rc != ncclSuccess) {
            set_err(errbuf, errlen, "ncclAllGather: %s", e->rccl->GetErrorString(rc));
            return ELLP_ERR_DEVICE;
        }
        return ELLP_OPTIMAL;
    }
    if (e->transport == 3 && e->xfn) {
        const size_t segb = sizeof(double) * (size_t)n;
        e->xhost.resize(segb * (size_t)e->world);
        HIPCHK(hipMemcpyAsync(e->xhost.data() + segb * (size_t)e->rank, src, segb, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->xfn(e->xuser, e->xhost.data(), (int64_t)segb, e->world) != 0) {
            set_err(errbuf, errlen, "the exchange callback failed");
            return ELLP_ERR_DEVICE;
        }
        HIPCHK(hipMemcpyAsync(dst, e->xhost.data(), segb * (size_t)e->world, hipMemcpyHostToDevice, e->stream));
        return ELLP_OPTIMAL;
    }
    set_err(errbuf, errlen, "no exchange transport has been set up (RCCL communicator, mailbox or callback)");
    return ELLP_ERR_ARG;
}

PackArgs pack_args(ellp_engine *e, int forced) {
    PackArgs a{};
    a.A_N = e->A_N; a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb}; a.N_index = e->N_index;
    a.pack = e->packs + (int64_t)e->rank * pack_doubles(e->ld);
    a.st = e->st; a.ld = e->ld; a.nN = e->nN; a.own0 = e->own0; a.own1 = e->own1;
    a.block0 = e->rank * e->nbs;
    int mine = e->nblocks - a.block0;
    if (mine > e->nbs) mine = e->nbs;
    if (mine < 0) mine = 0;
    a.nblk = mine; a.cpb = e->cpb; a.forced = forced; a.eps = e->eps;
    return a;
}
void launch_pack(ellp_engine *e, int forced) {
    hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, e->stream, pack_args(e, forced));
}
// mailbox transport: pack, push and wait of an iteration's compact exchange in one launch (k_sh_xchg)
void launch_xchg_fused(ellp_engine *e) {
    const int64_t pd = pack_doubles(e->ld);
    MboxArgs a{e->d_peer_slots, e->d_peer_flags, nullptr, e->packs, e->st, pd, e->rank, e->world, 300000000LL /* 3 s */};
    hipLaunchKernelGGL(k_sh_xchg, dim3((unsigned)(2 * e->world)), dim3(256), 0, e->stream, pack_args(e, 0), a, e->slot_doubles);
}
void launch_select(ellp_engine *e, int mode) {
    SelectArgs a{};
    a.packs = e->packs; a.xc = Xchg{e->X, e->seg, e->nbs, e->cpb}; a.N_index = e->N_index; a.A_N = e->A_N;
    a.aq_cur = e->aq_cur; a.st = e->st; a.ld = e->ld; a.nN = e->nN; a.own0 = e->own0; a.own1 = e->own1;
    a.world = e->world; a.nblocks = e->nblocks; a.cpb = e->cpb; a.mode = mode; a.eps = e->eps;
    hipLaunchKernelGGL(k_sh_select, dim3(1), dim3(256), sizeof(double) * (size_t)e->nblocks + 16, e->stream, a);
}
// the part of an iteration behind the selection: FTRAN with the given column, drift monitor, eta update
void launch_sharded_tail(ellp_engine *e) {
    if (e->lagged) {  // two-launch pipeline: F applies the eta update of the pivot k_price2 has booked and forms d; the iteration stays open
        {
            Prof p(e, ELLP_K_FTRAN);
            launch_ftran_eta(e);
        }
        launch_drift_check(e);
        e->lag_open = true;
        e->since_btran += 1;
        e->since_refactor += 1;
        e->enqueued += 1;
        return;
    }
    {
        Prof p(e, ELLP_K_FTRAN);
        launch_ftran2<0>(e);
    }
    launch_drift_check(e);
    {
        Prof p(e, ELLP_K_UPDATE);
        launch_update2<0>(e, 1);
    }
    e->since_btran += 1;
    e->since_refactor += 1;
    e->enqueued += 1;
}
ellp_status launch_sharded_iteration(ellp_engine *e, char *errbuf, size_t errlen) {
    const bool full_btran = !e->u_valid || e->since_btran >= (uint64_t)e->btran_refresh;
    if (full_btran) {
        Prof p(e, ELLP_K_BTRAN);
        launch_btran(e);
        e->since_btran = 0;
        e->u_valid = true;
    }
    {
        Prof p(e, ELLP_K_PRICE);
        if (e->lagged) launch_price2(e, e->lag_open ? 1 : 0);
        else launch_price<0>(e);
    }
    // Launches of a sharded iteration: pricing | exchange | FTRAN (the selection in its prologue) | update — four on the
    // mailbox transport (k_sh_xchg packs, pushes and waits in one), five over RCCL (k_pack, then the all-gather).
    // ELLP_SHARD_SPLIT=1: the seven-launch form of rounds 1-2 (k_pack, push, wait, commit, k_sh_select), for A/B and tests.
    static const bool split = getenv("ELLP_SHARD_SPLIT") && getenv("ELLP_SHARD_SPLIT")[0] == '1';
    {
        Prof p(e, ELLP_K_SELECT);
        const int64_t pd = pack_doubles(e->ld);
        if (!split && e->world > 1 && e->transport == 2 && 2 * e->world <= 64 && e->world * SH_KC <= 64) {
            launch_xchg_fused(e);
            e->sel_commit = true;
        } else {
            launch_pack(e, 0);
            const ellp_status s = shard_exchange(e, e->packs + (int64_t)e->rank * pd, e->packs, pd, errbuf, errlen, split || e->world * SH_KC > 64);
            if (s != ELLP_OPTIMAL) return s;
            e->sel_commit = !(split || e->world * SH_KC > 64) && e->world > 1 && e->transport == 2;
        }
        if (split || e->world * SH_KC > 64) launch_select(e, 1);
        else e->sel_in_ftran = true;
    }
    launch_sharded_tail(e);
    e->sel_in_ftran = false;
    e->sel_commit = false;
    return ELLP_OPTIMAL;
}
}  // namespace

ellp_status ellp_engine_mailbox_selftest(ellp_engine *e, int rounds, char *errbuf, size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e || e->transport != 2) return ELLP_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    const int64_t n = pack_doubles(e->ld);
    std::vector<double> mine((size_t)n), all((size_t)(n * e->world));
    double *d_src = e->packs + (int64_t)e->rank * n;
    for (int round = 0; round < rounds; ++round) {
        for (int64_t i = 0; i < n; ++i) mine[(size_t)i] = (double)(e->rank * 1000003 + round * 7919) + (double)i * 0.5;
        HIPCHK(hipMemcpyAsync(d_src, mine.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, e->stream));
        const ellp_status s = shard_exchange(e, d_src, e->packs, n, errbuf, errlen);
        if (s != ELLP_OPTIMAL) return s;
        HIPCHK(hipMemcpyAsync(all.data(), e->packs, sizeof(double) * (size_t)(n * e->world), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->h_st->status != ST_RUNNING) {
            // re-arm: the engine is still good, only this transport is not
            const int32_t running = ST_RUNNING;
            (void)hipMemcpy(&e->st->status, &running, sizeof(int32_t), hipMemcpyHostToDevice);
            e->h_st->status = ST_RUNNING;
            e->transport = 0;
            set_err(errbuf, errlen, "mailbox self-test: a wait timed out (round %d)", round);
            return ELLP_ERR_DEVICE;
        }
        for (int r = 0; r < e->world; ++r)
            for (int64_t i = 0; i < n; ++i) {
                const double want = (double)(r * 1000003 + round * 7919) + (double)i * 0.5;
                if (all[(size_t)(r * n + i)] != want) {
                    e->transport = 0;
                    set_err(errbuf, errlen, "mailbox self-test: word %lld of rank %d's segment is wrong in round %d", (long long)i, r, round);
                    return ELLP_ERR_DEVICE;
                }
            }
    }
    HIPCHK(hipMemset(e->packs, 0, sizeof(double) * (size_t)(n * e->world)));
    return ELLP_OPTIMAL;
}

// the loop of a column-sharded engine (collective: every rank passes the same max_iters)
static ellp_status run_colsharded(ellp_engine *e, uint64_t max_iters, ellp_stats *stats, char *errbuf, size_t errlen) {
    auto t0 = std::chrono::steady_clock::now();
    ellp_stats local;
    ellp_stats *sp = stats ? stats : &local;
    ellp_status result = ellp_engine_poll(e, sp, errbuf, errlen);
    const uint64_t iters0 = sp->iters;
    uint64_t remaining = max_iters;
    const uint64_t poll = e->opts.poll_interval > 0 ? (uint64_t)e->opts.poll_interval : 32;
    int64_t period = e->refactor_period > 0 ? e->refactor_period : default_period(e);
    auto rearm = [&]() {
        static const int32_t running = ST_RUNNING;  // static: the copy is asynchronous
        (void)hipMemcpyAsync(&e->st->status, &running, sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        e->h_st->status = ST_RUNNING;
    };
    while (result == ELLP_MAXITER && remaining > 0 && e->nN > 0) {
        const uint64_t batch = remaining < poll ? remaining : poll;
        for (uint64_t k = 0; k < batch; ++k) {
            if (e->since_refactor >= (uint64_t)period) maintain_inverse(e);
            const ellp_status s = launch_sharded_iteration(e, errbuf, errlen);
            if (s != ELLP_OPTIMAL) return s;
        }
        // read back; the two internal requests of the selection are serviced here, one iteration at a time
        for (;;) {
            HIPCHK(hipMemcpyAsync(e->h_st, e->st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            HIPCHK(hipGetLastError());
            prof_collect(e);
            reconcile_counters(e);
            const int stt = e->h_st->status;
            if (stt == ST_NEED_FULL) {
                // ties reach below the gap: gather the complete pricing output and fold it in full
                rearm();
                e->full_exchanges += 1;
                const ellp_status s = shard_exchange(e, e->X + (int64_t)e->rank * e->seg, e->X, e->seg, errbuf, errlen);
                if (s != ELLP_OPTIMAL) return s;
                launch_select(e, 2);
                launch_sharded_tail(e);
                continue;
            }
            if (stt == ST_NEED_COLUMN) {
                rearm();
                e->column_requests += 1;
                launch_pack(e, 1);
                const int64_t pd = pack_doubles(e->ld);
                const ellp_status s = shard_exchange(e, e->packs + (int64_t)e->rank * pd, e->packs, pd, errbuf, errlen);
                if (s != ELLP_OPTIMAL) return s;
                launch_select(e, 3);
                launch_sharded_tail(e);
                continue;
            }
            break;
        }
        (void)service_maintenance_request(e);
        result = e->h_st->status == ST_RUNNING ? ELLP_MAXITER : status_message(*e->h_st, errbuf, errlen);
        // (the statistics — with the objective kernel and a synchronisation of their own — are filled once, after the loop)
        const uint64_t done = e->h_st->iters - iters0;
        remaining = done < max_iters ? max_iters - done : 0;
    }
    fill_stats(e, sp);
    if (e->nN == 0) result = ELLP_OPTIMAL;
    if (stats) stats->t_loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return result;
}

ellp_status ellp_engine_run_sharded(ellp_engine *e, uint64_t max_iters, ellp_stats *stats, char *errbuf,
                                    size_t errlen) {
    if (e) e->hst_fresh = false;  // anything but ellp_engine_run may change the device state behind h_st
    if (!e) return ELLP_ERR_ARG;
    if (e->colshard) {
        if (errbuf && errlen) errbuf[0] = 0;
        HIPCHK(hipSetDevice(e->device));
        return run_colsharded(e, max_iters, stats, errbuf, errlen);
    }
    if (!e->comm || !e->rccl) {
        set_err(errbuf, errlen, "ellp_engine_comm_init has not been called");
        return ELLP_ERR_ARG;
    }
    if (errbuf && errlen) errbuf[0] = 0;
    HIPCHK(hipSetDevice(e->device));
    auto t0 = std::chrono::steady_clock::now();
    ellp_stats local;
    ellp_stats *sp = stats ? stats : &local;
    ellp_status result = ellp_engine_poll(e, sp, errbuf, errlen);  // a previous slice may have terminated
    const uint64_t iters0 = sp->iters;
    uint64_t remaining = max_iters;
    const uint64_t poll = e->opts.poll_interval > 0 ? (uint64_t)e->opts.poll_interval : 32;
    const double *mine = e->X + (int64_t)e->rank * e->seg;
    while (result == ELLP_MAXITER && remaining > 0 && e->nN > 0) {
        const uint64_t batch = remaining < poll ? remaining : poll;
        // one collective per iteration: step(2) = rest of iteration k + pricing of iteration k+1
        ellp_status s = ellp_engine_step(e, 0, errbuf, errlen);
        for (uint64_t k = 0; k < batch && s == ELLP_OPTIMAL; ++k) {
            const ncclResult_t rc =
                e->rccl->AllGather(mine, e->X, (size_t)e->seg, ncclDouble, e->comm, e->stream);  // in place
            if (rc != ncclSuccess) {
                set_err(errbuf, errlen, "ncclAllGather: %s", e->rccl->GetErrorString(rc));
                return ELLP_ERR_DEVICE;
            }
            s = ellp_engine_step(e, k + 1 < batch ? 2 : 1, errbuf, errlen);
        }
        if (s != ELLP_OPTIMAL) return s;
        result = ellp_engine_poll(e, sp, errbuf, errlen);
        // iterations that really ran (a maintenance request voids the rest of its batch); the
        // device state is replicated, so every rank computes the same `remaining`
        const uint64_t done = sp->iters - iters0;
        remaining = done < max_iters ? max_iters - done : 0;
    }
    if (e->nN == 0) result = ELLP_OPTIMAL;
    if (stats) stats->t_loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return result;
}

static ellp_status solve_once(int kind, int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                              const double *b, const uint8_t *bound_kind, const double *lb, const double *ub,
                              double *x, int64_t *B_index, int64_t n_B, int64_t *N_index, uint8_t *N_bound,
                              int64_t n_N, double *y, double *d, const ellp_opts *opts, ellp_stats *stats,
                              char *errbuf, size_t errlen) {
    ellp_engine *e = nullptr;
    ellp_status s = ellp_engine_create(kind, m, n, n_c, A, c, b, bound_kind, lb, ub, x, B_index, n_B, N_index,
                                       N_bound, n_N, y, d, opts, &e, errbuf, errlen);
    if (s != ELLP_OPTIMAL) return s;
    const uint64_t max_iter = opts ? opts->max_iter : 1000;
    s = ellp_engine_run(e, max_iter, stats, errbuf, errlen);
    if (s != ELLP_ERR_DEVICE) {
        ellp_status rs = ellp_engine_read_point(e, x, B_index, N_index, N_bound, y, d, errbuf, errlen);
        if (rs != ELLP_OPTIMAL) s = rs;
    }
    ellp_engine_destroy(e);
    return s;
}

ellp_status ellp_primal_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                                           const double *b, const uint8_t *bound_kind, const double *lb,
                                           const double *ub, double *x, int64_t *B_index, int64_t n_B,
                                           int64_t *N_index, uint8_t *N_bound, int64_t n_N, const ellp_opts *opts,
                                           ellp_stats *stats, char *errbuf, size_t errlen) {
    return solve_once(ELLP_ENGINE_PRIMAL, m, n, n_c, A, c, b, bound_kind, lb, ub, x, B_index, n_B, N_index, N_bound,
                      n_N, nullptr, nullptr, opts, stats, errbuf, errlen);
}

ellp_status ellp_dual_solve_with_initial(int64_t m, int64_t n, int64_t n_c, const double *A, const double *c,
                                         const double *b, const uint8_t *bound_kind, const double *lb,
                                         const double *ub, double *x, int64_t *B_index, int64_t n_B,
                                         int64_t *N_index, uint8_t *N_bound, int64_t n_N, double *y, double *d,
                                         const ellp_opts *opts, ellp_stats *stats, char *errbuf, size_t errlen) {
    return solve_once(ELLP_ENGINE_DUAL, m, n, n_c, A, c, b, bound_kind, lb, ub, x, B_index, n_B, N_index, N_bound,
                      n_N, y, d, opts, stats, errbuf, errlen);
}

}  // extern "C"
