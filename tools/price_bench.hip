// price_bench.hip — A/B micro-benchmark of pricing-kernel shapes (r_j = c_j - A_N[:,j].u).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/price_bench.hip -o /tmp/pb && /tmp/pb
// Variants are interleaved round-robin in ONE process (cdna guide §5.4 rule 24); prints median us
// and GB/s of algorithmic bytes (8*ld*nN) per variant, for the C3 and C5 shapes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                  \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <bool NT>
__device__ __forceinline__ double2 ld2(const double2 *p) {
    if (NT) {
        double2 r;
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
        return r;
    }
    return *p;
}

// block-per-column-group: BT threads stride down CF columns at once, u slice in registers
template <int T, int CF, int BT, bool NT>
__global__ __launch_bounds__(BT) void k_block(const double *A, const double *u, double *r, int64_t ld, int64_t nN,
                                              int cpb) {
    constexpr int NW = BT / 64;
    __shared__ double s_part[2][NW][CF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t half = ld >> 1;
    const double2 *u2 = reinterpret_cast<const double2 *>(u);
    double2 ur[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t idx = tid + BT * t;
        ur[t] = idx < half ? u2[idx] : make_double2(0.0, 0.0);
    }
    const int64_t j0 = (int64_t)blockIdx.x * cpb;
    const int64_t j1 = (j0 + cpb < nN) ? j0 + cpb : nN;
    int buf = 0;
    for (int64_t j = j0; j < j1; j += CF, buf ^= 1) {
        double acc[CF];
        const double2 *col[CF];
#pragma unroll
        for (int k = 0; k < CF; ++k) {
            acc[k] = 0.0;
            const int64_t jj = (j + k < j1) ? j + k : j1 - 1;
            col[k] = reinterpret_cast<const double2 *>(A + jj * ld);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t idx = tid + BT * t;
            if (idx < half) {
                double2 v[CF];
#pragma unroll
                for (int k = 0; k < CF; ++k) v[k] = ld2<NT>(col[k] + idx);
#pragma unroll
                for (int k = 0; k < CF; ++k) {
                    acc[k] = fma(v[k].x, ur[t].x, acc[k]);
                    acc[k] = fma(v[k].y, ur[t].y, acc[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CF; ++k) acc[k] = wave_sum(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < CF; ++k) s_part[buf][wave][k] = acc[k];
        }
        __syncthreads();
        if (tid < CF && j + tid < j1) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += s_part[buf][w][tid];
            r[j + tid] = s;
        }
    }
}

// wave-per-column: each wave owns whole columns (no block barrier), CF columns in flight
template <int CF, bool NT>
__global__ __launch_bounds__(256) void k_wave(const double *A, const double *u, double *r, int64_t ld, int64_t nN) {
    const int lane = threadIdx.x & 63;
    const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t half = ld >> 1;
    const double2 *u2 = reinterpret_cast<const double2 *>(u);
    for (int64_t j = wg * CF; j < nN; j += nw * CF) {
        double acc[CF];
        const double2 *col[CF];
#pragma unroll
        for (int k = 0; k < CF; ++k) {
            acc[k] = 0.0;
            col[k] = reinterpret_cast<const double2 *>(A + ((j + k < nN) ? j + k : nN - 1) * ld);
        }
        for (int64_t t0 = lane; t0 < half; t0 += 4 * 64) {
            double2 uu[4];
            double2 v[CF][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t t = t0 + 64 * q;
                const int64_t tc = t < half ? t : 0;
                uu[q] = u2[tc];
                if (t >= half) uu[q] = make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < CF; ++k) v[k][q] = ld2<NT>(col[k] + tc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < CF; ++k) {
                    acc[k] = fma(v[k][q].x, uu[q].x, acc[k]);
                    acc[k] = fma(v[k][q].y, uu[q].y, acc[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < CF; ++k) {
            const double s = wave_sum(acc[k]);
            if (lane == 0 && j + k < nN) r[j + k] = s;
        }
    }
}

struct Variant {
    const char *name;
    void (*launch)(const double *, const double *, double *, int64_t, int64_t, hipStream_t);
};

template <int T, int CF, int BT, bool NT, int GRIDK>
void launch_block(const double *A, const double *u, double *r, int64_t ld, int64_t nN, hipStream_t s) {
    int64_t cpb = (nN + GRIDK - 1) / GRIDK;
    if (cpb < 1) cpb = 1;
    const int g = (int)((nN + cpb - 1) / cpb);
    hipLaunchKernelGGL((k_block<T, CF, BT, NT>), dim3(g), dim3(BT), 0, s, A, u, r, ld, nN, (int)cpb);
}
template <int CF, bool NT, int GRID>
void launch_wave(const double *A, const double *u, double *r, int64_t ld, int64_t nN, hipStream_t s) {
    hipLaunchKernelGGL((k_wave<CF, NT>), dim3(GRID), dim3(256), 0, s, A, u, r, ld, nN);
}

template <int T256, int T512>
void run_shape(int64_t m, int64_t nN) {
    const int64_t ld = (m + 15) / 16 * 16;
    double *A, *u, *r;
    CHK(hipMalloc(&A, sizeof(double) * ld * nN));
    CHK(hipMalloc(&u, sizeof(double) * ld));
    CHK(hipMalloc(&r, sizeof(double) * nN));
    std::vector<double> h(ld * nN);
    unsigned long long s = 88172645463325252ull;
    for (auto &v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    CHK(hipMemcpy(A, h.data(), sizeof(double) * ld * nN, hipMemcpyHostToDevice));
    CHK(hipMemcpy(u, h.data(), sizeof(double) * ld, hipMemcpyHostToDevice));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    std::vector<Variant> vs = {
        {"block256 CF4 g1024 (engine)", launch_block<T256, 4, 256, false, 1024>},
        {"block256 CF4 g2048", launch_block<T256, 4, 256, false, 2048>},
        {"block256 CF8 g1024", launch_block<T256, 8, 256, false, 1024>},
        {"block256 CF2 g2048", launch_block<T256, 2, 256, false, 2048>},
        {"block256 CF4 g1024 nontemporal", launch_block<T256, 4, 256, true, 1024>},
        {"block512 CF4 g1024", launch_block<T512, 4, 512, false, 1024>},
        {"wave CF2 grid2048", launch_wave<2, false, 2048>},
        {"wave CF4 grid1024", launch_wave<4, false, 1024>},
        {"wave CF2 grid2048 nontemporal", launch_wave<2, true, 2048>},
    };
    const int rounds = 15, reps = 20;
    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int rd = 0; rd < rounds; ++rd) {
        for (size_t v = 0; v < vs.size(); ++v) {
            vs[v].launch(A, u, r, ld, nN, st);  // warm
            CHK(hipEventRecord(e0, st));
            for (int k = 0; k < reps; ++k) vs[v].launch(A, u, r, ld, nN, st);
            CHK(hipEventRecord(e1, st));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            t[v].push_back(ms * 1000.f / reps);
        }
    }
    CHK(hipGetLastError());
    printf("shape m=%lld |N|=%lld  (%.1f MB per pass)\n", (long long)m, (long long)nN, 8.0 * ld * nN / 1e6);
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(t[v].begin(), t[v].end());
        const double med = t[v][t[v].size() / 2], mn = t[v][0];
        printf("  %-34s median %8.2f us  min %8.2f us  -> %7.1f GB/s (median)\n", vs[v].name, med, mn,
               8.0 * ld * nN / med / 1e3);
    }
    CHK(hipFree(A)); CHK(hipFree(u)); CHK(hipFree(r));
}

int main() {
    run_shape<4, 2>(2000, 7000);
    run_shape<8, 4>(4000, 44000);
    return 0;
}
