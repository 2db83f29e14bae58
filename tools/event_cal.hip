// event_cal.hip — what does a hipEvent bracket around ONE kernel measure, relative to the kernel's
// own duration (rocprofv3 kernel-trace)?  Patterns, all on one non-blocking stream:
//   idle_pair    [a b] with a host sync between pairs            (queue idle)
//   steady_pair  spin kernel, then 32x [a b] enqueued at once    (queue busy, host ahead)
//   idle_K       [a K b] + sync
//   steady_K     spin, then 32x [a K b]
//   steady_null  spin, then 32x [a null b]
//   b2b_K        [a K x20 b]/20  = duration + back-to-back gap
// Run plain and under `rocprofv3 --kernel-trace --stats` and compare with the trace's k_dot avg.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_null() {}
__global__ void k_spin(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }

__global__ __launch_bounds__(256) void k_dot(const double *A, const double *u, double *r, int64_t ld, int64_t nN, int cpb) {
    __shared__ double s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * cpb, j1 = j0 + cpb < nN ? j0 + cpb : nN;
    for (int64_t j = j0; j < j1; ++j) {
        double acc = 0.0;
        for (int64_t i = tid; i < ld; i += 256) acc = fma(A[j * ld + i], u[i], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        if (tid == 0) r[j] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
    }
}

static double med(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1000.0; }

int main() {
    const int64_t m = 2000, ld = 2000, nN = 7000;
    double *A, *u, *r;
    CHK(hipMalloc(&A, 8 * ld * nN)); CHK(hipMalloc(&u, 8 * ld)); CHK(hipMalloc(&r, 8 * nN));
    CHK(hipMemset(A, 0, 8 * ld * nN)); CHK(hipMemset(u, 0, 8 * ld));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int P = 32;
    std::vector<hipEvent_t> ea(P), eb(P);
    for (int k = 0; k < P; ++k) { CHK(hipEventCreate(&ea[k])); CHK(hipEventCreate(&eb[k])); }
    auto K = [&] { hipLaunchKernelGGL(k_dot, dim3(1000), dim3(256), 0, st, A, u, r, ld, nN, 7); };
    auto collect = [&](int n) { std::vector<float> v; for (int k = 0; k < n; ++k) { float ms; CHK(hipEventElapsedTime(&ms, ea[k], eb[k])); v.push_back(ms); } return med(v); };
    for (int k = 0; k < 5; ++k) K();
    CHK(hipStreamSynchronize(st));
    (void)m;
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < P; ++k) { CHK(hipEventRecord(ea[k], st)); CHK(hipEventRecord(eb[k], st)); CHK(hipStreamSynchronize(st)); }
        const double idle_pair = collect(P);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 100);
        for (int k = 0; k < P; ++k) { CHK(hipEventRecord(ea[k], st)); CHK(hipEventRecord(eb[k], st)); }
        CHK(hipStreamSynchronize(st));
        const double steady_pair = collect(P);
        for (int k = 0; k < P; ++k) { CHK(hipEventRecord(ea[k], st)); K(); CHK(hipEventRecord(eb[k], st)); CHK(hipStreamSynchronize(st)); }
        const double idle_K = collect(P);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 100);
        for (int k = 0; k < P; ++k) { CHK(hipEventRecord(ea[k], st)); K(); CHK(hipEventRecord(eb[k], st)); }
        CHK(hipStreamSynchronize(st));
        const double steady_K = collect(P);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 100);
        for (int k = 0; k < P; ++k) { CHK(hipEventRecord(ea[k], st)); hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, st); CHK(hipEventRecord(eb[k], st)); }
        CHK(hipStreamSynchronize(st));
        const double steady_null = collect(P);
        // steady, kernels with NO brackets in between except every one (as the engine's profile run does): K K' pattern
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 100);
        CHK(hipEventRecord(ea[0], st));
        for (int k = 0; k < 20; ++k) K();
        CHK(hipEventRecord(eb[0], st));
        CHK(hipStreamSynchronize(st));
        float ms; CHK(hipEventElapsedTime(&ms, ea[0], eb[0]));
        const double b2b = ms * 1000.0 / 20;
        printf("rep %d: idle_pair %.2f  steady_pair %.2f  idle_K %.2f  steady_K %.2f  steady_null %.2f  b2b_K %.2f  (us)\n",
               rep, idle_pair, steady_pair, idle_K, steady_K, steady_null, b2b);
    }
    return 0;
}
