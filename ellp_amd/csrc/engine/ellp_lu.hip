// ellp_lu.hip — LU with partial pivoting of A^T on the device (the basis of dual phase 1).
//
// The reference picks the starting basis of dual phase 1 from `std_form.A.transpose().lu()`
// (src/solvers/dual/dual_problem.rs:139-160): all it consumes is the row permutation (which columns of A become
// basic) and the diagonal of U (`< EPS` -> panic).  That is n·m² flop on one host core — 56 GFLOP and about a
// minute at config 3's shape, a quarter of an hour at config 5's — in front of a simplex loop that takes seconds.
//
// Same algorithm as ellp_amd/csrc/host/dense.h LU / oracle lu_factor_inplace (pivot = FIRST entry of maximal
// modulus in the column; a zero pivot column is skipped; multipliers a·(1/diag); trailing update
// c_k[r] = (-c_k[i])·c_i[r] + c_k[r], skipped for a zero c_k[i]), and every stored number is BITWISE the host
// loop's: an entry M[r,k] receives its updates in the order of the steps whatever thread makes them, each one
// a separately rounded multiply and add (compiled with -ffp-contract=off).
//
// M = A^T (nv x m) is walked through A's own column-major storage: row r of M is column r of A, m contiguous
// doubles — so a row swap moves two contiguous rows, the trailing update of a row is a contiguous stream, and
// the matrix needs no transposition.  Two launches per elimination step: in k_lut_step each wave updates one
// row (its multiplier from the pivot row parked in a scratch buffer) and reports |M[r, i+1]| for the next pivot
// search; the one-block k_lut_fold folds the per-block candidates (first maximum by row), parks the next pivot
// row and the row it displaces, and records the pivot — rows are swapped lazily: the displaced row is read
// from its parked copy by the wave that owns the pivot's old position.  (Folding in the block that finishes
// last, one launch per step, was measured first: every block then needs an agent-scope release fence, which on
// this part writes back its XCD's L2 — the per-XCD L2s are not coherent with each other — and a launch never
// took less than 110 us however little there was to eliminate.  A kernel boundary does that write-back once.)
// Right-looking and unblocked: the whole trailing matrix is read and written once per step (about
// 16·nv·m²/2 bytes in all: 0.3 TB at config 3's shape); a blocked variant would divide that by the panel width.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "ellp_hip.h"
#include "ellp_lu_dev.h"

namespace {

struct LuState {
    long long piv;     // pivot row of the step the next k_lut_step eliminates
    double diag;       // its entry in the pivot column
    int32_t skip;      // diag == 0: the column is skipped (dense.h: `continue`)
    int32_t pad;
};
struct LuCand {
    double v;
    long long r;
};

constexpr int LU_RPB = 4;  // rows (= waves) per block

__device__ __forceinline__ bool lu_better(double ov, long long orr, double bv, long long br) {
    return orr >= 0 && (br < 0 || ov > bv || (ov == bv && orr < br));
}

// Launch `i` (i = -1: nothing to eliminate, only the candidates of column 0):
//   block 0            writes the pivot row of step i to position i (the other half of the lazy swap)
//   block 1 + b, wave w eliminates row r = i + 1 + 4 b + w with the pivot of step i and reports |M[r, i+1]|
__global__ __launch_bounds__(256) void k_lut_step(double *M, int64_t m, int64_t nv, int64_t i, const double *prow,
                                                  const double *irow, LuCand *cands, const LuState *st) {
    __shared__ double s_v[LU_RPB];
    __shared__ long long s_r[LU_RPB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long piv = st->piv;
    const bool skip = i < 0 || st->skip != 0;
    const int64_t nxt = i + 1;  // the column searched for the next step
    double cv = -1.0;
    long long cr = -1;
    if (blockIdx.x == 0) {
        if (!skip && piv != i)
            for (int64_t k = tid; k < m; k += 256) M[i * m + k] = prow[k];
    } else {
        const int64_t r = nxt + (int64_t)(blockIdx.x - 1) * LU_RPB + wave;
        if (r < nv) {
            double *dst = M + r * m;
            double first = 0.0;  // M[r, i+1] after this step
            if (!skip) {
                const bool moved = r == piv;  // this position receives the row the pivot displaced
                const double *src = moved ? irow : dst;
                const double inv_diag = 1.0 / st->diag;
                const double l = __dmul_rn(src[i], inv_diag);
                if (moved)
                    for (int64_t k = lane; k < i; k += 64) dst[k] = src[k];
                if (lane == 0) dst[i] = l;
                // four 64-entry chunks in flight per wave: the loop is a chain of dependent round trips otherwise
                for (int64_t k0 = nxt + lane; k0 < m; k0 += 256) {
                    double pv[4], sv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t k = k0 + 64 * u;
                        const int64_t kc = k < m ? k : m - 1;
                        pv[u] = prow[kc];
                        sv[u] = src[kc];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t k = k0 + 64 * u;
                        if (k < m) {
                            const double f = -pv[u];
                            double val = sv[u];
                            if (f != 0.0) val = __dadd_rn(__dmul_rn(f, l), val);
                            if (f != 0.0 || moved) dst[k] = val;
                            if (u == 0 && k == nxt) first = val;
                        }
                    }
                }
            } else if (nxt < m && lane == 0) {
                first = dst[nxt];
            }
            if (nxt < m) {
                first = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(first)),
                                         __builtin_amdgcn_readfirstlane(__double2loint(first)));
                double v = fabs(first);
                if (v != v) v = (r == nxt) ? INFINITY : -1.0;  // `v > best` is false for a NaN: only the diagonal can carry one
                cv = v;
                cr = r;
            }
        }
    }
    if (nxt >= m) return;  // the last elimination: nothing to search
    if (lane == 0) {
        s_v[wave] = cv;
        s_r[wave] = cr;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = s_v[0];
        long long br = s_r[0];
        for (int w = 1; w < LU_RPB; ++w)
            if (lu_better(s_v[w], s_r[w], bv, br)) {
                bv = s_v[w];
                br = s_r[w];
            }
        cands[blockIdx.x] = LuCand{bv, br};
    }
}

// one block of 1024 threads: the pivot of column nxt from the ncand per-block candidates, rows piv and nxt parked
__global__ __launch_bounds__(1024) void k_lut_fold(const double *M, int64_t m, int64_t nxt, const LuCand *cands,
                                                   unsigned ncand, double *prow, double *irow, LuState *st,
                                                   int64_t *pivot_out, double *udiag_out) {
    __shared__ double s_v[16];
    __shared__ long long s_r[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bv = -1.0;
    long long br = -1;
    for (unsigned b = tid; b < ncand; b += 1024) {
        const LuCand c = cands[b];
        if (lu_better(c.v, c.r, bv, br)) {
            bv = c.v;
            br = c.r;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o);
        const long long orr = __shfl_xor(br, o);
        if (lu_better(ov, orr, bv, br)) {
            bv = ov;
            br = orr;
        }
    }
    if (lane == 0) {
        s_v[wave] = bv;
        s_r[wave] = br;
    }
    __syncthreads();
    bv = s_v[0];
    br = s_r[0];
    for (int w = 1; w < 16; ++w)
        if (lu_better(s_v[w], s_r[w], bv, br)) {
            bv = s_v[w];
            br = s_r[w];
        }
    // br >= 0: row nxt itself is always a candidate (nv >= m)
    const double diag = M[br * m + nxt];
    for (int64_t k = tid; k < m; k += 1024) {
        prow[k] = M[br * m + k];
        irow[k] = M[nxt * m + k];
    }
    if (tid == 0) {
        st->piv = br;
        st->diag = diag;
        st->skip = diag == 0.0 ? 1 : 0;
        pivot_out[nxt] = diag == 0.0 ? nxt : br;  // a skipped column appends no transposition (dense.h)
        udiag_out[nxt] = diag;
    }
}

void set_err(char *errbuf, size_t len, const char *msg, hipError_t e) {
    if (errbuf && len) snprintf(errbuf, len, "%s: %s", msg, hipGetErrorString(e));
}

}  // namespace

// ---- the same factorisation of a DEVICE-RESIDENT square matrix stored by rows (ellp_lu_dev.h): the certificate of the
// certified hybrid above 1,024 rows (ellp_exact.inc) factors the basis with it.  M (m x m, row r at M + r m) is overwritten
// by the factors exactly as the oracle's lu_factor_inplace leaves them (L's multipliers below the diagonal, U on and above;
// rows in pivoted order), piv[i] = the row exchanged with row i at step i, udiag[i] = U_ii.
hipError_t ellp_lu_rows_alloc(EllpLuWork *w, int64_t m) {
    memset(w, 0, sizeof(*w));
    w->m = m;
    hipError_t rc;
    const unsigned max_blocks = (unsigned)((m + LU_RPB - 1) / LU_RPB) + 1;
    if ((rc = hipMalloc(reinterpret_cast<void **>(&w->M), sizeof(double) * (size_t)(m * m))) != hipSuccess) return rc;
    if ((rc = hipMalloc(reinterpret_cast<void **>(&w->prow), sizeof(double) * (size_t)m)) != hipSuccess) return rc;
    if ((rc = hipMalloc(reinterpret_cast<void **>(&w->irow), sizeof(double) * (size_t)m)) != hipSuccess) return rc;
    if ((rc = hipMalloc(reinterpret_cast<void **>(&w->udiag), sizeof(double) * (size_t)m)) != hipSuccess) return rc;
    if ((rc = hipMalloc(reinterpret_cast<void **>(&w->piv), sizeof(int64_t) * (size_t)m)) != hipSuccess) return rc;
    if ((rc = hipMalloc(&w->cands, sizeof(LuCand) * (size_t)max_blocks)) != hipSuccess) return rc;
    if ((rc = hipMalloc(&w->st, sizeof(LuState))) != hipSuccess) return rc;
    return hipSuccess;
}
void ellp_lu_rows_free(EllpLuWork *w) {
    (void)hipFree(w->M); (void)hipFree(w->prow); (void)hipFree(w->irow); (void)hipFree(w->udiag); (void)hipFree(w->piv);
    (void)hipFree(w->cands); (void)hipFree(w->st);
    memset(w, 0, sizeof(*w));
}
void ellp_lu_rows_factor(EllpLuWork *w, hipStream_t stream) {
    const int64_t m = w->m;
    (void)hipMemsetAsync(w->st, 0, sizeof(LuState), stream);
    for (int64_t i = -1; i < m; ++i) {
        const int64_t rows = m - i - 1;
        const unsigned grid = (unsigned)((rows + LU_RPB - 1) / LU_RPB) + 1;
        hipLaunchKernelGGL(k_lut_step, dim3(grid), dim3(256), 0, stream, w->M, m, m, i, w->prow, w->irow, static_cast<LuCand *>(w->cands),
                           static_cast<LuState *>(w->st));
        if (i + 1 < m)
            hipLaunchKernelGGL(k_lut_fold, dim3(1), dim3(1024), 0, stream, w->M, m, i + 1, static_cast<const LuCand *>(w->cands), grid, w->prow,
                               w->irow, static_cast<LuState *>(w->st), w->piv, w->udiag);
    }
}

extern "C" ellp_status ellp_hip_lu_transposed(int64_t m, int64_t nv, const double *A, int64_t *pivot_out,
                                              double *udiag_out, int device, char *errbuf, size_t errlen) {
    if (errbuf && errlen) errbuf[0] = 0;
    if (m < 0 || nv < 0 || (m > 0 && nv > 0 && (!A || !pivot_out || !udiag_out))) return ELLP_ERR_ARG;
    if (nv < m) {
        if (errbuf && errlen) snprintf(errbuf, errlen, "ellp_hip_lu_transposed needs nv >= m (every column of A^T gets a pivot row)");
        return ELLP_ERR_ARG;
    }
    if (m == 0) return ELLP_OPTIMAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        if (errbuf && errlen) snprintf(errbuf, errlen, "no HIP device available");
        return ELLP_ERR_DEVICE;
    }
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return ELLP_ERR_DEVICE;
    double *dM = nullptr, *prow = nullptr, *irow = nullptr, *udiag = nullptr;
    int64_t *piv = nullptr;
    LuCand *cands = nullptr;
    LuState *st = nullptr;
    hipStream_t stream = nullptr;
    hipError_t rc = hipSuccess;
    auto cleanup = [&] {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        (void)hipFree(dM); (void)hipFree(prow); (void)hipFree(irow); (void)hipFree(udiag); (void)hipFree(piv); (void)hipFree(cands); (void)hipFree(st);
    };
#define LCHK(expr)                                      \
    do {                                                \
        rc = (expr);                                    \
        if (rc != hipSuccess) {                         \
            set_err(errbuf, errlen, #expr, rc);         \
            cleanup();                                  \
            return ELLP_ERR_DEVICE;                     \
        }                                               \
    } while (0)
    const unsigned max_blocks = (unsigned)((nv + LU_RPB - 1) / LU_RPB) + 1;
    LCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    LCHK(hipMalloc(reinterpret_cast<void **>(&dM), sizeof(double) * (size_t)(m * nv)));
    LCHK(hipMalloc(reinterpret_cast<void **>(&prow), sizeof(double) * (size_t)m));
    LCHK(hipMalloc(reinterpret_cast<void **>(&irow), sizeof(double) * (size_t)m));
    LCHK(hipMalloc(reinterpret_cast<void **>(&udiag), sizeof(double) * (size_t)m));
    LCHK(hipMalloc(reinterpret_cast<void **>(&piv), sizeof(int64_t) * (size_t)m));
    LCHK(hipMalloc(reinterpret_cast<void **>(&cands), sizeof(LuCand) * (size_t)max_blocks));
    LCHK(hipMalloc(reinterpret_cast<void **>(&st), sizeof(LuState)));
    LCHK(hipMemcpyAsync(dM, A, sizeof(double) * (size_t)(m * nv), hipMemcpyHostToDevice, stream));
    LCHK(hipMemsetAsync(st, 0, sizeof(LuState), stream));
    for (int64_t i = -1; i < m; ++i) {
        const int64_t rows = nv - i - 1;
        const unsigned grid = (unsigned)((rows + LU_RPB - 1) / LU_RPB) + 1;
        hipLaunchKernelGGL(k_lut_step, dim3(grid), dim3(256), 0, stream, dM, m, nv, i, prow, irow, cands, st);
        if (i + 1 < m)
            hipLaunchKernelGGL(k_lut_fold, dim3(1), dim3(1024), 0, stream, dM, m, i + 1, cands, grid, prow, irow, st, piv, udiag);
    }
    LCHK(hipGetLastError());
    LCHK(hipMemcpyAsync(pivot_out, piv, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, stream));
    LCHK(hipMemcpyAsync(udiag_out, udiag, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, stream));
    LCHK(hipStreamSynchronize(stream));
#undef LCHK
    cleanup();
    return ELLP_OPTIMAL;
}
