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

// ---- cross-lane sums without the LDS crossbar.  __shfl_xor is ds_bpermute_b32 (an address register, two LDS-pipe operations per double
// and their ~100-cycle round trip); on a lone wave that latency is not hidden.  gfx950 can do every level of a butterfly in the VALU:
//   partner 1, 2: DPP quad_perm;  4: row_half_mirror (i <-> 7-i);  8: row_mirror (i <-> 15-i) or row_ror:8 (i <-> i^8);
//   16, 32: v_permlane16_swap / v_permlane32_swap (rows of 16 / 32 lanes of TWO registers exchanged: odd rows of the first with even rows
//   of the second) — with both registers holding x, the two results are {own, partner} in one half of the wave and {partner, own} in
//   the other: their sum is x + x_partner in every lane, no select.
// The mirror levels pair lane i with another partner than i ^ 4 / i ^ 8, which is as good for an all-reduce: every level is a perfect
// matching whose two lanes both form a + b, so all lanes end with the same bits (taken in ascending order of the levels).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// a (from this lane) and b (from this lane), rows of ROW lanes: returns {x0, x1} with x0 + x1 = (own a + partner's a) in even rows and
// (partner's b + own b) in odd rows — the butterfly-transpose step; with a == b the plain pair sum
template <int ROW>
__device__ __forceinline__ void swap_rows(double& a, double& b) {
    static_assert(ROW == 16 || ROW == 32, "v_permlane16_swap / v_permlane32_swap");
    const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    unsigned alo = (unsigned)ba, ahi = (unsigned)(ba >> 32), blo = (unsigned)bb, bhi = (unsigned)(bb >> 32);
#if __has_builtin(__builtin_amdgcn_permlane32_swap)   // (the device pass; through the builtin, not inline asm: the compiler inserts the wait states a VALU write -> permlane read needs)
    if (ROW == 16) {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false), h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
    } else {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false), h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
    }
#endif
    a = __longlong_as_double(((long long)ahi << 32) | alo);
    b = __longlong_as_double(((long long)bhi << 32) | blo);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppRor8 = 0x128;

// The butterfly transpose of WaveTranspose with every level in the VALU (no ds_bpermute): rows of 32 / 16 lanes exchanged between two
// registers (v_permlane32_swap / v_permlane16_swap: no select either), then DPP row_ror:8, row_half_mirror (i <-> 7 - i), quad_perm
// xor 2 / xor 1.  The partner of a lane always differs from it in the level's lane bit, so the index range a lane ends up owning
// ([off, off + real), at most one value for N <= 64) follows from its lane bits exactly as in WaveTranspose.
template <int N, int BIT>
struct WaveTransposeValu {
    static constexpr int H = (N + 1) / 2;
    __device__ static __forceinline__ void run(double (&v)[N < 1 ? 1 : N], int lane, int& off, int& real) {
        const bool up = (lane & BIT) != 0;
        double nv[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            const double lo = v[i];
            const double hi = (i + H < N) ? v[(i + H < N) ? i + H : 0] : 0.0;
            if constexpr (BIT == 32 || BIT == 16) {
                double a = lo, b = hi;
                swap_rows<BIT>(a, b);
                nv[i] = a + b;
            } else {
                const double keep = up ? hi : lo, send = up ? lo : hi;
                nv[i] = keep + dpp_f64<(BIT == 8 ? kDppRor8 : BIT == 4 ? kDppHalfMirror : BIT == 2 ? kDppXor2 : kDppXor1)>(send);
            }
        }
        if (up) { off += H; real = real - H > 0 ? real - H : 0; }
        else real = real < H ? real : H;
#pragma unroll
        for (int i = 0; i < H; i++) v[i] = nv[i];
        if constexpr (BIT > 1) {
            double (&w)[H] = reinterpret_cast<double (&)[H]>(v);
            WaveTransposeValu<H, BIT / 2>::run(w, lane, off, real);
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
