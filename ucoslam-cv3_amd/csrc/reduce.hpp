// Deterministic block reductions shared by the fp64 optimisation kernels (ba.hip: 256 threads = the default NW = 4 waves;
// pnp.hip: 512 threads, NW = 8).
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr int kRedThreads = 256;

// deterministic block sum of one double per thread, result valid in EVERY thread: xor butterfly inside each wave (fixed
// pairing, all 64 lanes end with the wave total), then the wave totals are added in wave order.  Two barriers instead of the
// nine of an LDS tree — the fp64 kernels call this on their serial paths.
template <int NW = 4>
__device__ __forceinline__ double block_sum(double v, double* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();   // a previous call's readers are done with s_red
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r += s_red[w];
    return r;
}
// Deterministic block reduction of N values per thread.  Inside a wave the N x 64 values are reduced with a butterfly
// TRANSPOSE: at every step a lane keeps one half of its values, ships the other half to its partner (lane ^ bit) and adds
// what it receives, so the number of live values per lane halves while the number of lanes sharing a sum doubles —
// N-1 + (padding) cross-lane moves in total instead of 6*N for N independent butterflies.  After the six steps lane
// `l` owns the wave total of value index off(l) (at most one per lane).  The four wave totals are then added in wave order.
// Every addition has a fixed operand pairing, so the result is run-to-run deterministic.
template <int N, int BIT>
struct WaveTranspose {
    static constexpr int H = (N + 1) / 2;
    // `real` = how many of this lane's N slots carry real values (the rest is zero padding introduced by odd halvings)
    __device__ static __forceinline__ void run(double (&v)[N < 1 ? 1 : N], int lane, int& off, int& real) {
        const bool up = (lane & BIT) != 0;
        double nv[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            const double lo = v[i];
            const double hi = (i + H < N) ? v[(i + H < N) ? i + H : 0] : 0.0;
            const double keep = up ? hi : lo, send = up ? lo : hi;
            nv[i] = keep + __shfl_xor(send, BIT);
        }
        if (up) { off += H; real = real - H > 0 ? real - H : 0; }
        else real = real < H ? real : H;
#pragma unroll
        for (int i = 0; i < H; i++) v[i] = nv[i];
        if constexpr (BIT > 1) {
            double (&w)[H] = reinterpret_cast<double (&)[H]>(v);
            WaveTranspose<H, BIT / 2>::run(w, lane, off, real);
        }
    }
};

template <int N, int NW = 4>
__device__ __forceinline__ void block_sum_vec(double (&v)[N], double* s_part /* NW*N */, double* s_out /* N */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int off = 0, real = N;
    WaveTranspose<N, 32>::run(v, lane, off, real);
    if (real >= 1) s_part[wv * N + off] = v[0];   // exactly one lane per value index ends with a real slot; the others hold padding
    __syncthreads();
    if ((int)threadIdx.x < N) {
        double r = s_part[threadIdx.x];
#pragma unroll
        for (int w = 1; w < NW; w++) r += s_part[w * N + threadIdx.x];   // wave order (for NW = 4: ((p0 + p1) + p2) + p3, as before)
        s_out[threadIdx.x] = r;
    }
    __syncthreads();
}

__device__ __forceinline__ double block_max(double v, double* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = s_red[0];
#pragma unroll
    for (int w = 1; w < kRedThreads / 64; w++) r = fmax(r, s_red[w]);
    return r;
}

}  // namespace
