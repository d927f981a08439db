// Times the persistent local BA's redundant solve (bordered LDL^T + back substitution of the 6 nfree + 1 row system, one workgroup of 256
// threads, one wave per SIMD) WITHOUT stamps inside: shader clocks (s_memtime) of the best of `reps` repetitions, and |S x - b| against the
// original matrix.  Round 5 on MI355X, eight free keyframes: rounds 3-4's form (ldlt_rowlane_lds + backsolve_lds, removed) 15 250 + 4 780 clocks,
// ldlt_rowlane_v2 + backsolve_v2 10 870 + 2 520 (profiles/r05_solve48.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ucoslam-cv3_amd/csrc -I include scripts/micro/solve48.hip ucoslam-cv3_amd/csrc/ctx.hip -o scripts/micro/solve48.bin
#include <hip/hip_runtime.h>
#include "ba.hip"
#include <cstdio>
#include <random>
#include <vector>

__device__ __forceinline__ long long now_clk() { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int FORM>
__global__ __launch_bounds__(256) void k(const double* A, double* out, int n, int nfree, long long* clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __builtin_amdgcn_s_setprio(3);
    const int ld = n + 1;
    double* M = lds;
    double* s_w = M + ld * ld + (ld * ld & 1);
    __shared__ short s_pair[160][2];
    __shared__ double s_x[128];
    __shared__ double s_zero[2];
    if (threadIdx.x == 0) s_zero[0] = 0.0;
    const int npairs = nfree * (nfree + 1) / 2;
    for (int t = threadIdx.x; t < npairs; t += 256) { int s1 = 0, rem = t; while (rem >= nfree - s1) { rem -= nfree - s1; ++s1; } s_pair[t][0] = s1; s_pair[t][1] = s1 + rem; }
    long long bf = 1ll << 60, bb = 1ll << 60, bt = 1ll << 60;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < ld * ld; i += 256) M[i] = A[i];
        __syncthreads();
        const long long t0 = now_clk();
        bool failed;
        failed = ldlt_rowlane_v2(M, n, ld, nfree, npairs, s_pair, s_w);
        __syncthreads();
        const long long t1 = now_clk();
        backsolve_v2(M, n, ld, s_x, s_zero);
        __syncthreads();
        const long long t2 = now_clk();
        if (t1 - t0 < bf) bf = t1 - t0;
        if (t2 - t1 < bb) bb = t2 - t1;
        if (t2 - t0 < bt) bt = t2 - t0;
        if (failed && threadIdx.x == 0) clk[3] = 1;
        __syncthreads();
    }
    if (threadIdx.x == 0) { clk[0] = bf; clk[1] = bb; clk[2] = bt; }
    if (threadIdx.x < n) out[threadIdx.x] = s_x[threadIdx.x];
}

template <int FORM>
int run(int nfree) {
    const int n = 6 * nfree, ld = n + 1;
    std::mt19937 rng(nfree); std::normal_distribution<double> N(0, 1);
    std::vector<double> B(n * n), A(ld * ld, 0.0);
    for (auto& v : B) v = N(rng);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = i == j ? 50.0 : 0.0; for (int kk = 0; kk < n; kk++) s += B[i * n + kk] * B[j * n + kk]; A[i * ld + j] = s; }
    for (int i = 0; i < n; i++) for (int j = i + 1; j < ld; j++) A[i * ld + j] = std::nan("");   // the upper triangle is never to be used
    for (int j = 0; j < n; j++) A[n * ld + j] = N(rng);
    double *dA, *dO; long long* dc;
    (void)hipMalloc(&dA, A.size() * 8); (void)hipMalloc(&dO, 256 * 8); (void)hipMalloc(&dc, 8 * 8);
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dc, 0, 64);
    const size_t lds = (ld * ld + 2 + 2 * 6 * 64 + 512) * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<FORM>, dim3(1), dim3(256), lds, 0, dA, dO, n, nfree, dc, 50);
    (void)hipDeviceSynchronize();
    long long c[8]; (void)hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    std::vector<double> x(256); (void)hipMemcpy(x.data(), dO, 256 * 8, hipMemcpyDeviceToHost);
    double errx = 0;
    for (int i = 0; i < n; i++) { double sx = 0; for (int j = 0; j < n; j++) sx += (j <= i ? A[i * ld + j] : A[j * ld + i]) * x[j]; const double e = fabs(sx - A[n * ld + i]); errx = (e > errx || e != e) ? e : errx; }
    printf("form %d nfree %2d: factor %6lld | backsolve %6lld | both %6lld clocks (%.2f us at 2.4 GHz)  |S x - b| = %.3g%s\n", FORM, nfree, c[0], c[1], c[2], c[2] / 2400.0, errx, c[3] ? "  FAILED PIVOT" : "");
    (void)hipFree(dA); (void)hipFree(dO); (void)hipFree(dc);
    return !(errx < 1e-9);
}
int main() {
    int bad = 0;
    for (int nf : {8, 7, 5, 3, 2, 1, 10}) bad += run<1>(nf);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
