// tools/mfma_f64_check.hip — which lane holds what in v_mfma_f64_16x16x4_f64 (gfx950)?  Build & run:
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_check.hip -o /tmp/mfma_check && /tmp/mfma_check
// Hypothesis checked: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[i][j] in register v of lane l with
// j = l % 16, i = 4 (l / 16) + v.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + (l / 16)];   // A row-major 16 x 4: A[i][k], i = l % 16, k = l / 16
    const double b = B[(l / 16) * 16 + (l % 16)];  // B row-major 4 x 16: B[k][j], k = l / 16, j = l % 16
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i; hB[i] = 2.0 - 0.11 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int j = l % 16, i = 4 * (l / 16) + v;
        if (fabs(hD[l * 4 + v] - ref[i * 16 + j]) > 1e-9) ++bad;
    }
    printf("layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    if (bad) {  // print what each lane/register holds in terms of (i, j)
        for (int l = 0; l < 64; l += 5) for (int v = 0; v < 4; ++v) {
            for (int e = 0; e < 256; ++e) if (fabs(hD[l * 4 + v] - ref[e]) < 1e-9) printf("lane %d reg %d = D[%d][%d]\n", l, v, e / 16, e % 16);
        }
    }
    return bad != 0;
}
