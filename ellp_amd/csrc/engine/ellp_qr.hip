// ellp_qr.hip — column-pivoted Householder QR of A^T on the device.
//
// The reference's standard form runs `A.transpose().col_piv_qr()` (src/standard_form.rs:142) to
// detect redundant rows; all it consumes is the column transposition list and |R_ii|
// (:143-181).  On one host core that is ~2*n*m^2 flop — about a minute at m=2000, n=7000 and a
// quarter of an hour at config 5 — while the simplex loop itself takes seconds on the GPU.
//
// This is the same algorithm as ellp_amd/csrc/host/dense.h ColPivQR / oracle col_piv_qr (pivot =
// column holding the entry of maximal modulus of the trailing block, first such entry in
// column-major order; Householder reflector with nalgebra's normalisation), arranged so that
// every floating-point result is BITWISE what the host loop produces: each dot product and
// each norm is accumulated by ONE thread in the host's order (r ascending, multiply then add,
// compiled with -ffp-contract=off).  Parallelism comes from the independent columns (one
// thread per column, coalesced because M = A^T is walked through A's own column-major storage:
// M[r, j] = A[j + r*m]) and from the element-wise passes.  Not bandwidth-optimal — it does not
// need to be: 4 launches per elimination step (tools/qr_time.py for timings).
//
// That arrangement is kept as the EXACT mode (ELLP_QR_EXACT=1; the bitwise tests run it).  The default is the FAST mode:
// the same algorithm step for step — nalgebra's pivot rule needs the fully updated trailing block before the next
// reflector is known, so the steps cannot be blocked into GEMMs — with the one-accumulator-per-column chains given up:
// norms and dot products are reduced in parallel (fixed, run-to-run reproducible orders), which leaves the three passes
// over the trailing block per step (read for the dots, read + write for the update with the next pivot search fused in)
// running at memory speed.  Results differ from the host loop in the last bits of |R_ii|; pivots differ only where two
// entries of the trailing block tie to within that rounding (rank detection is pinned end to end only).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ellp_hip.h"

namespace {

struct QrState {
    int64_t pj;       // pivot column of the current step
    int32_t skip;     // factor == 0: no reflector to apply (dense.h: `continue`)
    int32_t near_tie; // fast mode: some step's two best pivot candidates were within 1e-12 (relative) of each other
};

// per-column candidate for the next pivot search: cand[j] = max_r |M[r, j]| over the trailing rows
// (the first such r wins inside the column, but only the column matters for the pivot)
__global__ __launch_bounds__(256) void k_qr_scan(const double *A, int64_t m, int64_t nv, int64_t i, double *cand) {
    const int64_t j = i + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double best = -1.0;
    for (int64_t r = i; r < nv; ++r) {
        const double v = fabs(A[j + r * m]);
        if (v > best) best = v;
    }
    cand[j] = best;
}

// every block: pj = first column (lowest j >= i) holding the maximal candidate — host order is
// j ascending then r ascending with a strict '>', and the initial value |M[i,i]| belongs to
// column i, so the lowest j among the maxima wins.  Then the block swaps its slice of columns
// i and pj of M (rows i, pj of A) and copies the pivot column's trailing part to xbuf.
__global__ __launch_bounds__(256) void k_qr_swap(double *A, int64_t m, int64_t nv, int64_t i, const double *cand,
                                                 double *xbuf, QrState *st, int64_t *pivot_out) {
    __shared__ double s_v[4], s_2[4];
    __shared__ long long s_j[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bv = -1.0, sv = -1.0;  // the best candidate and the value of the second best (for the near-tie test of the fast mode)
    long long bj = -1;
    for (int64_t j = i + tid; j < m; j += 256) {
        const double v = cand[j];
        if (bj < 0 || v > bv) {  // this thread's j ascend: strict '>' keeps its lowest j
            sv = bv;
            bv = v;
            bj = j;
        } else if (v > sv) {
            sv = v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o), os = __shfl_xor(sv, o);
        const long long oj = __shfl_xor(bj, o);
        if (oj >= 0 && (bj < 0 || ov > bv || (ov == bv && oj < bj))) {
            sv = fmax(fmax(sv, os), bj >= 0 ? bv : -1.0);
            bv = ov;
            bj = oj;
        } else {
            sv = fmax(fmax(sv, os), oj >= 0 ? ov : -1.0);
        }
    }
    if (lane == 0) {
        s_v[wave] = bv;
        s_2[wave] = sv;
        s_j[wave] = bj;
    }
    __syncthreads();
    bv = s_v[0];
    sv = s_2[0];
    bj = s_j[0];
    for (int w = 1; w < 4; ++w) {
        if (s_j[w] >= 0 && (bj < 0 || s_v[w] > bv || (s_v[w] == bv && s_j[w] < bj))) {
            sv = fmax(fmax(sv, s_2[w]), bj >= 0 ? bv : -1.0);
            bv = s_v[w];
            bj = s_j[w];
        } else {
            sv = fmax(fmax(sv, s_2[w]), s_j[w] >= 0 ? s_v[w] : -1.0);
        }
    }
    // two candidates closer than the parallel reductions of the fast mode can tell apart (they agree with the host loop's
    // sums to ~1e-14 relative): the host re-runs the factorisation in the exact mode (ellp_hip_qr_transposed)
    // (EXACT ties are left to the index rule, as in the host loop: bit-equal candidates are entries no reflection has touched —
    // the slack entries of a standard form end up as the largest entry of every trailing column, all exactly 1 — or the results of
    // identical operations on identical data, equal in both modes)
    if (blockIdx.x == 0 && tid == 0 && sv >= 0.0 && bv > 0.0 && bv - sv > 0.0 && bv - sv <= 1e-12 * bv) st->near_tie = 1;
    const int64_t pj = bj;
    if (blockIdx.x == 0 && tid == 0) {
        st->pj = pj;
        pivot_out[i] = pj;
    }
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    if (r < nv) {
        double a = A[i + r * m];
        if (pj != i) {
            const double b = A[pj + r * m];
            A[pj + r * m] = a;
            A[i + r * m] = b;
            a = b;
        }
        if (r >= i) xbuf[r - i] = a;
    }
}

// the reflector of dense.h ColPivQR: the two norms are accumulated by thread 0 in the host's order
// (from LDS, which the whole block fills chunk by chunk — a single thread reading global memory
// pays a full round trip per element), the element-wise divisions by all threads
constexpr int XCHUNK = 4096;
__device__ __forceinline__ double sequential_sumsq(const double *x, int64_t len, double *s_x, double *s_out) {
    const int tid = threadIdx.x;
    double acc = 0.0;  // meaningful in thread 0
    for (int64_t c0 = 0; c0 < len; c0 += XCHUNK) {
        const int64_t n = (len - c0) < XCHUNK ? (len - c0) : XCHUNK;
        for (int64_t k = tid; k < n; k += 256) {  // the squares are formed by all threads; only the additions are the chain
            const double v = x[c0 + k];
            s_x[k] = v * v;
        }
        __syncthreads();
        if (tid == 0) {
            int64_t k = 0;
            for (; k + 8 <= n; k += 8) {  // eight LDS reads in flight in front of eight dependent additions
                double q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = s_x[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += q[u];
            }
            for (; k < n; ++k) acc += s_x[k];
        }
        __syncthreads();
    }
    if (tid == 0) *s_out = acc;
    __syncthreads();
    return *s_out;
}

__global__ __launch_bounds__(256) void k_qr_reflect(double *xbuf, int64_t len, int64_t i, QrState *st, double *rdiag) {
    __shared__ double s_x[XCHUNK];
    __shared__ double s_sum, s_sf;
    __shared__ int s_skip;
    const int tid = threadIdx.x;
    const double sqn = sequential_sumsq(xbuf, len, s_x, &s_sum);
    if (tid == 0) {
        const double norm = sqrt(sqn);
        const double x0 = xbuf[0];
        const double signed_norm = (x0 < 0.0) ? -norm : norm;
        const double factor = (sqn + fabs(x0) * norm) * 2.0;
        rdiag[i] = norm;
        xbuf[0] = x0 + signed_norm;
        s_skip = factor == 0.0 ? 1 : 0;
        s_sf = sqrt(factor);
        st->skip = s_skip;
    }
    __syncthreads();
    if (s_skip) return;
    const double sf = s_sf;
    for (int64_t r = tid; r < len; r += 256) xbuf[r] = xbuf[r] / sf;
    __syncthreads();
    const double n2 = sqrt(sequential_sumsq(xbuf, len, s_x, &s_sum));
    if (n2 != 0.0)
        for (int64_t r = tid; r < len; r += 256) xbuf[r] = xbuf[r] / n2;
}

// dot[j] = sum_r x[r] * M[i + r, j] for every trailing column j > i, each accumulated by ONE lane in the host's order
// (r ascending, multiply then add).  That fixes one accumulator per column, but not who fetches: a wave takes 16
// columns, ALL 64 lanes fetch (lane = 16 s + c loads rows 4 u + s of column c: four 128-byte segments per load
// instruction, 16 instructions = 64 rows in flight per wave), the tile goes through LDS, and lanes 0..15 do the
// additions from there while the next tile's loads are already in flight.  (One lane per column with 64 columns per
// wave left 62 waves on the whole device at config 5's shape and 0.5 TB/s: 2.3 ms per step.)
// Writes f2[j] = -2 * dot and clears the column's pivot-search candidate for k_qr_update.
constexpr int DOT_COLS = 16;   // columns per wave
constexpr int DOT_ROWS = 64;   // rows per tile
__global__ __launch_bounds__(64) void k_qr_dot(const double *A, int64_t m, int64_t nv, int64_t i, const double *xbuf,
                                               const QrState *st, double *f2, double *cand) {
    __shared__ double s_t[2][DOT_ROWS][DOT_COLS];
    __shared__ double s_x[2][DOT_ROWS];
    const int lane = threadIdx.x, c = lane & 15, s = lane >> 4;
    const int64_t j = i + 1 + (int64_t)blockIdx.x * DOT_COLS + c;
    const bool colok = j < m;
    if (lane < DOT_COLS && colok) cand[j] = 0.0;
    if (st->skip) {
        if (lane < DOT_COLS && colok) f2[j] = 0.0;
        return;
    }
    const int64_t len = nv - i;
    const double *cj = A + (colok ? j : i + 1) + i * m;  // M[i + r, j] = cj[r * m]; clamped column: loaded, never used
    double dot = 0.0;
    double v[DOT_ROWS / 4];
    double xv = 0.0;
    auto load_tile = [&](int64_t r0) {
#pragma unroll
        for (int u = 0; u < DOT_ROWS / 4; ++u) {
            const int64_t r = r0 + 4 * u + s;
            v[u] = cj[(r < len ? r : len - 1) * m];
        }
        const int64_t rx = r0 + lane;
        xv = xbuf[rx < len ? rx : len - 1];
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < DOT_ROWS / 4; ++u) s_t[buf][4 * u + s][c] = v[u];
        s_x[buf][lane] = xv;
    };
    load_tile(0);
    store_tile(0);
    int buf = 0;
    for (int64_t r0 = 0; r0 < len; r0 += DOT_ROWS) {
        const bool more = r0 + DOT_ROWS < len;
        if (more) load_tile(r0 + DOT_ROWS);  // in flight during the additions below
        __builtin_amdgcn_wave_barrier();
        const int n = (len - r0) < DOT_ROWS ? (int)(len - r0) : DOT_ROWS;
        if (lane < DOT_COLS) {
            if (n == DOT_ROWS) {
#pragma unroll 8
                for (int k = 0; k < DOT_ROWS; ++k) dot += s_x[buf][k] * s_t[buf][k][c];
            } else {
                for (int k = 0; k < n; ++k) dot += s_x[buf][k] * s_t[buf][k][c];
            }
        }
        if (more) store_tile(buf ^ 1);  // one wave: LDS operations are in order, the other buffer's readers are done
        buf ^= 1;
    }
    if (lane < DOT_COLS && colok) f2[j] = -2.0 * dot;
}

// M[i + r, j] = f2[j] * x[r] + M[i + r, j] over the whole trailing block (element-wise: any order),
// and the per-column maximum of |.| over rows i+1.. for the next pivot search (a maximum is exact in
// any order too: atomicMax on the bit pattern of a non-negative double)
constexpr int UPD_R = 64;
__global__ __launch_bounds__(256) void k_qr_update(double *A, int64_t m, int64_t nv, int64_t i, const double *xbuf,
                                                   const QrState *st, const double *f2, double *cand) {
    const int64_t j = i + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t len = nv - i;
    const int64_t r0 = (int64_t)blockIdx.y * UPD_R;
    const int64_t r1 = (r0 + UPD_R < len) ? r0 + UPD_R : len;
    double *cj = A + j + i * m;
    const bool skip = st->skip != 0;
    const double f = f2[j];
    double best = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        double v = cj[r * m];
        if (!skip) {
            v = f * xbuf[r] + v;
            cj[r * m] = v;
        }
        if (r > 0) best = fmax(best, fabs(v));
    }
    if (best > 0.0)
        atomicMax(reinterpret_cast<unsigned long long *>(cand + j), (unsigned long long)__double_as_longlong(best));
}

// ---------------------------------------------------------------------------------------------- FAST mode kernels
// cand[j] = max_r |M[r, j]| over all rows: (column tile, row chunk) blocks, atomicMax on the bit pattern (cand zeroed before)
constexpr int FQ_UNROLL = 8;
__global__ __launch_bounds__(256) void k_fq_scan(const double *__restrict__ A, int64_t m, int64_t nv, int64_t rows_per, double *cand) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = (r0 + rows_per < nv) ? r0 + rows_per : nv;
    double best = 0.0;
    for (int64_t r = r0; r < r1; r += FQ_UNROLL) {
        double v[FQ_UNROLL];
#pragma unroll
        for (int u = 0; u < FQ_UNROLL; ++u) v[u] = A[j + ((r + u < r1) ? r + u : r1 - 1) * m];
#pragma unroll
        for (int u = 0; u < FQ_UNROLL; ++u) best = fmax(best, fabs(v[u]));
    }
    atomicMax(reinterpret_cast<unsigned long long *>(cand + j), (unsigned long long)__double_as_longlong(best));
}

__device__ __forceinline__ double block_sum_1024(double v, double *s_w) {  // fixed order: lanes by xor butterfly, then the 16 waves in turn
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s_w[w];
    return t;
}

// the reflector of dense.h ColPivQR with parallel norms
__global__ __launch_bounds__(1024) void k_fq_reflect(double *xbuf, int64_t len, int64_t i, QrState *st, double *rdiag) {
    __shared__ double s_w[16];
    __shared__ double s_sf;
    __shared__ int s_skip;
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int64_t r = tid; r < len; r += 1024) {
        const double v = xbuf[r];
        acc += v * v;
    }
    const double sqn = block_sum_1024(acc, s_w);
    if (tid == 0) {
        const double norm = sqrt(sqn);
        const double x0 = xbuf[0];
        const double signed_norm = (x0 < 0.0) ? -norm : norm;
        const double factor = (sqn + fabs(x0) * norm) * 2.0;
        rdiag[i] = norm;
        xbuf[0] = x0 + signed_norm;
        s_skip = factor == 0.0 ? 1 : 0;
        s_sf = sqrt(factor);
        st->skip = s_skip;
    }
    __syncthreads();
    if (s_skip) return;
    const double sf = s_sf;
    acc = 0.0;
    for (int64_t r = tid; r < len; r += 1024) {
        const double v = xbuf[r] / sf;
        xbuf[r] = v;
        acc += v * v;
    }
    const double n2 = sqrt(block_sum_1024(acc, s_w));
    if (n2 != 0.0)
        for (int64_t r = tid; r < len; r += 1024) xbuf[r] = xbuf[r] / n2;
}

// part[chunk][j] = sum over the chunk's rows of x[r] * M[i + r, j]: one thread per column (coalesced along j), FQ_UNROLL
// rows in flight, the chunks summed in order by k_fq_f2
__global__ __launch_bounds__(256) void k_fq_dot(const double *__restrict__ A, int64_t m, int64_t nv, int64_t i, const double *__restrict__ xbuf,
                                                const QrState *st, int64_t rows_per, double *__restrict__ part) {
    const int64_t j = i + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (st->skip || j >= m) return;
    const int64_t len = nv - i;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = (r0 + rows_per < len) ? r0 + rows_per : len;
    const double *cj = A + j + i * m;
    double dot = 0.0;
    for (int64_t r = r0; r < r1; r += FQ_UNROLL) {
        double v[FQ_UNROLL], x[FQ_UNROLL];
#pragma unroll
        for (int u = 0; u < FQ_UNROLL; ++u) {
            const int64_t rr = (r + u < r1) ? r + u : r1 - 1;
            v[u] = cj[rr * m];
            x[u] = (r + u < r1) ? xbuf[rr] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < FQ_UNROLL; ++u) dot += x[u] * v[u];
    }
    part[(int64_t)blockIdx.y * m + j] = dot;
}
__global__ __launch_bounds__(256) void k_fq_f2(int64_t m, int64_t i, const QrState *st, int nchunk, const double *__restrict__ part, double *f2, double *cand) {
    const int64_t j = i + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    cand[j] = 0.0;
    double dot = 0.0;
    if (!st->skip)
        for (int c = 0; c < nchunk; ++c) dot += part[(int64_t)c * m + j];
    f2[j] = -2.0 * dot;
}

// the update of k_qr_update with FQ_UNROLL rows in flight per thread
__global__ __launch_bounds__(256) void k_fq_update(double *__restrict__ A, int64_t m, int64_t nv, int64_t i, const double *__restrict__ xbuf,
                                                   const QrState *st, const double *__restrict__ f2, double *cand) {
    const int64_t j = i + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t len = nv - i;
    const int64_t r0 = (int64_t)blockIdx.y * UPD_R;
    const int64_t r1 = (r0 + UPD_R < len) ? r0 + UPD_R : len;
    double *cj = A + j + i * m;
    const bool skip = st->skip != 0;
    const double f = f2[j];
    double best = 0.0;
    for (int64_t r = r0; r < r1; r += FQ_UNROLL) {
        double v[FQ_UNROLL], x[FQ_UNROLL];
#pragma unroll
        for (int u = 0; u < FQ_UNROLL; ++u) {
            const int64_t rr = (r + u < r1) ? r + u : r1 - 1;
            v[u] = cj[rr * m];
            x[u] = xbuf[rr];
        }
#pragma unroll
        for (int u = 0; u < FQ_UNROLL; ++u) {
            if (r + u >= r1) continue;
            double w = v[u];
            if (!skip) {
                w = f * x[u] + w;
                cj[(r + u) * m] = w;
            }
            if (r + u > 0) best = fmax(best, fabs(w));
        }
    }
    if (best > 0.0)
        atomicMax(reinterpret_cast<unsigned long long *>(cand + j), (unsigned long long)__double_as_longlong(best));
}

void set_err(char *errbuf, size_t len, const char *msg, hipError_t e) {
    if (errbuf && len) snprintf(errbuf, len, "%s: %s", msg, hipGetErrorString(e));
}

}  // namespace

static ellp_status qr_transposed_impl(int64_t m, int64_t nv, const double *A, int64_t *pivot_out, double *rdiag_out, int device,
                                      char *errbuf, size_t errlen, bool exact, int *near_tie) {
    if (errbuf && errlen) errbuf[0] = 0;
    if (m < 0 || nv < 0 || (m > 0 && nv > 0 && (!A || !pivot_out || !rdiag_out))) return ELLP_ERR_ARG;
    const int64_t mn = m < nv ? m : nv;
    if (mn == 0) return ELLP_OPTIMAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        if (errbuf && errlen) snprintf(errbuf, errlen, "no HIP device available");
        return ELLP_ERR_DEVICE;
    }
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return ELLP_ERR_DEVICE;
    constexpr int MAXCHUNK = 64;
    double *dA = nullptr, *xbuf = nullptr, *cand = nullptr, *rdiag = nullptr, *f2 = nullptr, *part = nullptr;
    int64_t *piv = nullptr;
    QrState *st = nullptr;
    hipStream_t stream = nullptr;
    hipError_t rc = hipSuccess;
    auto cleanup = [&] {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        (void)hipFree(dA); (void)hipFree(xbuf); (void)hipFree(cand); (void)hipFree(rdiag); (void)hipFree(f2); (void)hipFree(part); (void)hipFree(piv); (void)hipFree(st);
    };
#define QCHK(expr)                                      \
    do {                                                \
        rc = (expr);                                    \
        if (rc != hipSuccess) {                         \
            set_err(errbuf, errlen, #expr, rc);         \
            cleanup();                                  \
            return ELLP_ERR_DEVICE;                     \
        }                                               \
    } while (0)
    QCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    QCHK(hipMalloc(reinterpret_cast<void **>(&dA), sizeof(double) * (size_t)(m * nv)));
    QCHK(hipMalloc(reinterpret_cast<void **>(&xbuf), sizeof(double) * (size_t)nv));
    QCHK(hipMalloc(reinterpret_cast<void **>(&cand), sizeof(double) * (size_t)m));
    QCHK(hipMalloc(reinterpret_cast<void **>(&f2), sizeof(double) * (size_t)m));
    QCHK(hipMalloc(reinterpret_cast<void **>(&rdiag), sizeof(double) * (size_t)mn));
    QCHK(hipMalloc(reinterpret_cast<void **>(&piv), sizeof(int64_t) * (size_t)mn));
    QCHK(hipMalloc(reinterpret_cast<void **>(&st), sizeof(QrState)));
    QCHK(hipMemcpyAsync(dA, A, sizeof(double) * (size_t)(m * nv), hipMemcpyHostToDevice, stream));
    QCHK(hipMemsetAsync(st, 0, sizeof(QrState), stream));
    if (exact) {
        hipLaunchKernelGGL(k_qr_scan, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, dA, m, nv, (int64_t)0, cand);
    } else {
        QCHK(hipMalloc(reinterpret_cast<void **>(&part), sizeof(double) * (size_t)(m * MAXCHUNK)));
        QCHK(hipMemsetAsync(cand, 0, sizeof(double) * (size_t)m, stream));
        const int64_t tiles = (m + 255) / 256;
        int64_t nch = 2048 / tiles;
        nch = nch < 1 ? 1 : (nch > (nv + 63) / 64 ? (nv + 63) / 64 : nch);
        const int64_t rows_per = (nv + nch - 1) / nch;
        hipLaunchKernelGGL(k_fq_scan, dim3((unsigned)tiles, (unsigned)((nv + rows_per - 1) / rows_per)), dim3(256), 0, stream, dA, m, nv,
                           rows_per, cand);
    }
    for (int64_t i = 0; i < mn; ++i) {
        hipLaunchKernelGGL(k_qr_swap, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, dA, m, nv, i, cand, xbuf,
                           st, piv);
        if (exact) hipLaunchKernelGGL(k_qr_reflect, dim3(1), dim3(256), 0, stream, xbuf, nv - i, i, st, rdiag);
        else hipLaunchKernelGGL(k_fq_reflect, dim3(1), dim3(1024), 0, stream, xbuf, nv - i, i, st, rdiag);
        if (i + 1 < m) {
            const int64_t ncol = m - i - 1, len = nv - i;
            if (exact) {
                hipLaunchKernelGGL(k_qr_dot, dim3((unsigned)((ncol + DOT_COLS - 1) / DOT_COLS)), dim3(64), 0, stream, dA, m, nv, i, xbuf, st,
                                   f2, cand);
                hipLaunchKernelGGL(k_qr_update, dim3((unsigned)((ncol + 255) / 256), (unsigned)((len + UPD_R - 1) / UPD_R)),
                                   dim3(256), 0, stream, dA, m, nv, i, xbuf, st, f2, cand);
            } else {
                const int64_t tiles = (ncol + 255) / 256;
                int64_t nch = 2048 / tiles;  // enough blocks to fill the device, at most MAXCHUNK partial sums per column
                const int64_t maxch = (len + 63) / 64;
                nch = nch > MAXCHUNK ? MAXCHUNK : nch;
                nch = nch > maxch ? maxch : nch;
                nch = nch < 1 ? 1 : nch;
                const int64_t rows_per = (len + nch - 1) / nch;
                const int64_t nchunk = (len + rows_per - 1) / rows_per;
                hipLaunchKernelGGL(k_fq_dot, dim3((unsigned)tiles, (unsigned)nchunk), dim3(256), 0, stream, dA, m, nv, i, xbuf, st, rows_per, part);
                hipLaunchKernelGGL(k_fq_f2, dim3((unsigned)tiles), dim3(256), 0, stream, m, i, st, (int)nchunk, part, f2, cand);
                hipLaunchKernelGGL(k_fq_update, dim3((unsigned)tiles, (unsigned)((len + UPD_R - 1) / UPD_R)), dim3(256), 0, stream, dA, m, nv,
                                   i, xbuf, st, f2, cand);
            }
        }
    }
    QCHK(hipGetLastError());
    QCHK(hipMemcpyAsync(pivot_out, piv, sizeof(int64_t) * (size_t)mn, hipMemcpyDeviceToHost, stream));
    QCHK(hipMemcpyAsync(rdiag_out, rdiag, sizeof(double) * (size_t)mn, hipMemcpyDeviceToHost, stream));
    QrState hs{};
    QCHK(hipMemcpyAsync(&hs, st, sizeof(QrState), hipMemcpyDeviceToHost, stream));
    QCHK(hipStreamSynchronize(stream));
#undef QCHK
    cleanup();
    if (near_tie) *near_tie = hs.near_tie;
    return ELLP_OPTIMAL;
}

// Fast mode by default; a factorisation in which two pivot candidates came closer than the fast mode's parallel reductions
// can tell apart (different, but within 1e-12 relative; exact ties go by the index rule in both modes) is done again in the exact mode,
// whose every number is the host loop's: the pivot order — which standard_form.rs:142-181 turns into the ROW ORDER of the
// standard form, hence into the basis order and every tie-break downstream — is then the host's and the reference's in
// every case.  ELLP_QR_EXACT=1: exact from the start; ELLP_QR_EXACT=0: fast without the fall-back (measurements).
extern "C" ellp_status ellp_hip_qr_transposed(int64_t m, int64_t nv, const double *A, int64_t *pivot_out,
                                              double *rdiag_out, int device, char *errbuf, size_t errlen) {
    const char *exact_env = getenv("ELLP_QR_EXACT");
    const bool exact = exact_env && exact_env[0] == '1';
    const bool no_fallback = exact_env && exact_env[0] == '0';
    int near_tie = 0;
    ellp_status s = qr_transposed_impl(m, nv, A, pivot_out, rdiag_out, device, errbuf, errlen, exact, &near_tie);
    if (s == ELLP_OPTIMAL && !exact && !no_fallback && near_tie)
        s = qr_transposed_impl(m, nv, A, pivot_out, rdiag_out, device, errbuf, errlen, true, nullptr);
    return s;
}
