// Times ldlt_bordered_lds (ba.hip) alone on one workgroup, with shader-clock stamps inside wave 0.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ucoslam-cv3_amd/csrc scripts/micro/ldlt_time.hip -o /tmp/ldlt_time
#include <hip/hip_runtime.h>
__device__ long long g_clk[64];
__device__ long long g_wait[2], g_t0[2], g_tile, g_tt, g_nm;
#define UH_LDLT_CLK(i) do { if ((i) >= 102) { if (threadIdx.x == 64) { if ((i) == 102) g_tt = clock64(); else if ((i) == 103) g_tile += clock64() - g_tt; else g_nm++; } } else if ((i) >= 100) { if ((threadIdx.x & 63) == 0 && threadIdx.x < 128) { const int w_ = threadIdx.x >> 6; if ((i) == 100) g_t0[w_] = clock64(); else g_wait[w_] += clock64() - g_t0[w_]; } } else if (threadIdx.x == 0) g_clk[(i) < 64 ? (i) : 0] = clock64(); } while (0)
#include "ba.hip"
#include <cstdio>
#include <vector>
#include <random>
__global__ __launch_bounds__(256) void k(const double* A, double* out, int n, int nfree, long long* clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ld = n + 1;
    double* M = lds;
    double (*s_w)[121][6] = reinterpret_cast<double (*)[121][6]>(M + ld * ld + 2);
    __shared__ short s_pair[64][2];
    const int npairs = nfree * (nfree + 1) / 2;
    for (int t = threadIdx.x; t < npairs; t += 256) { int s1 = 0, rem = t; while (rem >= nfree - s1) { rem -= nfree - s1; ++s1; } s_pair[t][0] = s1; s_pair[t][1] = s1 + rem; }
    long long best = 1ll << 60, bestb = 1ll << 60;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < ld * ld; i += 256) M[i] = A[i];
        __syncthreads();
        const long long t0 = clock64();
        ldlt_bordered_lds(M, n, ld, nfree, npairs, s_pair, s_w);
        __syncthreads();
        const long long t1 = clock64();
        if (t1 - t0 < best) best = t1 - t0;
        __syncthreads();
        __shared__ double s_x[128];
        __shared__ double s_zero[2];
        if (threadIdx.x == 0) s_zero[0] = 0.0;
        const long long t2 = clock64();
        if (n <= 63) backsolve_v2(M, n, ld, s_x, s_zero); else backsolve2_lds(M, n, ld, s_x);   // (two rows per lane from 64 rows on: that form keeps no D — only |S x - b| is meaningful there)
        __syncthreads();
        const long long t3 = clock64();
        if (t3 - t2 < bestb) bestb = t3 - t2;
        if (threadIdx.x < n) out[ld * ld + threadIdx.x] = s_x[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) { clk[65] = g_wait[0] / reps; clk[66] = g_wait[1] / reps; g_wait[0] = g_wait[1] = 0; }
    __syncthreads();
    if (threadIdx.x == 64) { clk[67] = g_tile / reps; clk[68] = g_nm / reps; g_tile = 0; g_nm = 0; }
    if (threadIdx.x == 0) { clk[0] = best; clk[64] = bestb; for (int i = 0; i < 63; i++) clk[1 + i] = g_clk[i]; }
    for (int i = threadIdx.x; i < ld * ld; i += 256) out[i] = M[i];
}
int run(int nfree) {
    const int n = 6 * nfree, ld = n + 1;
    std::mt19937 rng(nfree); std::normal_distribution<double> N(0, 1);
    std::vector<double> B(n * n), A(ld * ld, 0.0);
    for (auto& v : B) v = N(rng);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = i == j ? 50.0 : 0.0; for (int k = 0; k < n; k++) s += B[i * n + k] * B[j * n + k]; A[i * ld + j] = s; }
    for (int i = 0; i < n; i++) for (int j = i + 1; j < ld; j++) A[i * ld + j] = std::nan("");   // the upper triangle is never to be used
    for (int j = 0; j < n; j++) A[n * ld + j] = N(rng);
    double *dA, *dO; long long* dc;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dO, (A.size() + 256) * 8); hipMalloc(&dc, 69 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (ld * ld + 2 + 2 * 129 * 6) * 8;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), lds, 0, dA, dO, n, nfree, dc, 20);
    hipDeviceSynchronize();
    long long c[69]; hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    std::vector<double> O(A.size() + 256); hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost);
    auto worse = [](double a, double b) { return (b > a || b != b) ? b : a; };   // (a NaN is an error, not a smaller number)
    // from 64 rows on the two-rows-per-lane form runs: it keeps no D (zeros on and above the diagonal) — recover d_k = S_kk - sum_t L_kt^2 d_t
    std::vector<double> D(n);
    for (int k = 0; k < n; k++) { if (n + 1 <= 64) D[k] = O[k * ld + k]; else { double s = A[k * ld + k]; for (int t = 0; t < k; t++) s -= O[k * ld + t] * O[k * ld + t] * D[t]; D[k] = s; } }
    double err = 0, errb = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k <= j; k++) { const double li = k == i ? 1.0 : O[i * ld + k], lj = k == j ? 1.0 : O[j * ld + k]; s += li * D[k] * lj; } err = worse(err, fabs(s - A[i * ld + j])); }
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k <= i; k++) s += (k == i ? 1.0 : O[i * ld + k]) * D[k] * O[n * ld + k]; errb = worse(errb, fabs(s - A[n * ld + i])); }   // L D z = b
    double errx = 0;   // S x = b with the original matrix
    for (int i = 0; i < n; i++) { double sx = 0; for (int j = 0; j < n; j++) sx += (j <= i ? A[i * ld + j] : A[j * ld + i]) * O[ld * ld + j]; errx = worse(errx, fabs(sx - A[n * ld + i])); }
    printf("   backsolve %lld clocks (%.2f us), |S x - b| = %.3g\n", c[64], c[64] / 2390.0, errx);
    printf("   barrier waits per factorisation: wave 0 (panels) %lld, wave 1 (trailing update) %lld clocks\n", c[65], c[66]);
    printf("nfree %d: best %lld shader clocks (%.2f us at 2.39 GHz), |LDL^T - A| = %.3g, |L D z - b| = %.3g\n", nfree, c[0], c[0] / 2390.0, err, errb);
#ifdef LDLT_PRINT_CLK
    if (nfree == 8) {   // wave 0's stamps: 0 = start; per block column kb: 1+4kb after the panel application, 3+4kb after pivots + stores, 4+4kb behind the barrier
        for (int kb = 0; kb < nfree; kb++) {
            const long long b0 = kb == 0 ? c[1 + 0] : c[1 + 4 + 4 * (kb - 1)];
            printf("   kb %d: loads + panel application %lld | pivots + stores %lld | barrier %lld clocks\n", kb, c[1 + 1 + 4 * kb] - b0, c[1 + 3 + 4 * kb] - c[1 + 1 + 4 * kb], c[1 + 4 + 4 * kb] - c[1 + 3 + 4 * kb]);
        }
    }
#endif
    hipFree(dA); hipFree(dO); hipFree(dc);
    return !(err < 1e-9 && errb < 1e-9 && errx < 1e-9);
}
int main() {
    int bad = 0;
    for (int nf : {1, 2, 3, 5, 7, 8, 9, 10, 12, 16, 17, 20}) bad += run(nf);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
