// Times ldlt_solve_mfma_lds (ldlt_mfma.hpp) alone on one workgroup of 256 / 512 threads: packed triangle (the stand-alone solve of 22-32
// free keyframes) and the row-stride form; checks L D L^T = A and S x = b.  Thread 0's phase stamps cost ~250 clocks each.
// scripts/micro/run_ldlt_mfma_time.sh
#include <hip/hip_runtime.h>
__device__ long long g_acc[8];
__device__ long long g_last;
#define UH_LDLTM_CLK(i) do { if (threadIdx.x == 0) { const long long t_ = clock64(); if (i > 0) g_acc[i] += t_ - g_last; g_last = t_; } } while (0)
#include "ba.hip"
#include <cstdio>
#include <vector>
#include <random>
template <bool PACKED, int NT>
__global__ __launch_bounds__(NT) void k(const double* A, const double* b, double* out, double* xout, int n, long long* clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ld = n + 1;
    const int msize = PACKED ? (n + 1) * (n + 2) / 2 : ld * ld;
    double* M = lds;
    double* s_aux = M + ((msize + 9) & ~1);
    double* s_x = s_aux + kLdltAux;
    long long best = 1ll << 60, bestb = 1ll << 60;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < msize; i += NT) M[i] = A[i];
        for (int i = threadIdx.x; i < n; i += NT) s_x[i] = b[i];
        __syncthreads();
        const long long t0 = clock64();
        ldlt_solve_mfma_lds<PACKED>(M, n, ld, s_aux, s_x);
        __syncthreads();
        const long long t1 = clock64();
        if (t1 - t0 < best) best = t1 - t0;
        if (r == 0) for (int i = threadIdx.x; i < msize; i += NT) out[i] = M[i];
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += NT) xout[i] = s_x[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) { clk[0] = best; clk[1] = bestb; for (int i = 0; i < 8; i++) { clk[2 + i] = g_acc[i] / reps; g_acc[i] = 0; } }
}
template <bool PACKED, int NT>
int run(int nfree) {
    const int n = 6 * nfree, ld = n + 1;
    std::mt19937 rng(nfree); std::normal_distribution<double> N(0, 1);
    std::vector<double> B(n * n), S(n * n), rhs(n);
    for (auto& v : B) v = N(rng);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = i == j ? 50.0 : 0.0; for (int k = 0; k < n; k++) s += B[i * n + k] * B[j * n + k]; S[i * n + j] = S[j * n + i] = s; }
    for (auto& v : rhs) v = N(rng);
    const int msize = PACKED ? (n + 1) * (n + 2) / 2 : ld * ld;
    std::vector<double> A(msize, std::nan(""));
    auto IX = [&](int r, int c) { return PACKED ? r * (r + 1) / 2 + c : r * ld + c; };
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) A[IX(i, j)] = S[i * n + j];
    for (int j = 0; j < n; j++) A[IX(n, j)] = rhs[j];
    double *dA, *dO, *db, *dx; long long* dc;
    hipMalloc(&dA, msize * 8); hipMalloc(&dO, msize * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&dc, 128);
    hipMemcpy(dA, A.data(), msize * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, rhs.data(), n * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(msize + 10 + kLdltAux + n + 8) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PACKED, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<PACKED, NT>), dim3(1), dim3(NT), lds, 0, dA, db, dO, dx, n, dc, 10);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed (%zu bytes of LDS)\n", lds); return 1; }
    long long c[14]; hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    std::vector<double> O(msize), x(n); hipMemcpy(O.data(), dO, msize * 8, hipMemcpyDeviceToHost); hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
    // D: packed keeps it on the diagonal; the bordered form does not — recover d_k = S_kk - sum_t L_kt^2 d_t
    std::vector<double> D(n);
    for (int k2 = 0; k2 < n; k2++) { D[k2] = O[IX(k2, k2)]; }
    double err = 0, errx = 0, scale = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k2 = 0; k2 <= j; k2++) { const double li = k2 == i ? 1.0 : O[IX(i, k2)], lj = k2 == j ? 1.0 : O[IX(j, k2)]; s += li * D[k2] * lj; } err = fmax(err, fabs(s - S[i * n + j])); scale = fmax(scale, fabs(S[i * n + j])); }
    for (int i = 0; i < n; i++) { double sx = 0; for (int j = 0; j < n; j++) sx += S[i * n + j] * x[j]; errx = fmax(errx, fabs(sx - rhs[i])); }
    printf("%s %d threads nfree %2d (n %3d): solve %6lld clocks (%6.2f us), |LDL^T - A| / max|A| = %.3g, |S x - b| = %.3g\n", PACKED ? "packed" : "stride", NT, nfree, n,
           c[0], c[0] / 2390.0, err / scale, errx);
    printf("      thread 0: first panels %lld, second panel application %lld, second panels %lld, trailing %lld, its barrier %lld, substitution %lld clocks\n", c[3], c[4], c[5], c[6], c[2] + c[7], c[8]);
    hipFree(dA); hipFree(dO); hipFree(db); hipFree(dx); hipFree(dc);
    return !(err / scale < 1e-12 && errx < 1e-9);
}
int main() {
    int bad = 0;
    for (int nf : {11, 12, 16, 17, 19, 21}) bad += run<false, 256>(nf);
    for (int nf : {17, 21, 22, 23, 24, 28, 31, 32}) bad += run<true, 256>(nf);
    for (int nf : {17, 21}) bad += run<false, 512>(nf);
    for (int nf : {17, 21, 22, 23, 24, 28, 31, 32}) bad += run<true, 512>(nf);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
