// Latency of the dependent-chain building blocks the serial part of the local BA is made of, one wave per SIMD (256 threads, only wave 0
// measured; the others idle at a barrier or spin on VALU work, see MODE): shader clocks (s_memtime) per link of a chain of N links.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/chain_latency.hip -o /tmp/chain_latency && /tmp/chain_latency
#include <hip/hip_runtime.h>
#include "reduce.hpp"
#include <cstdio>
#include <vector>

#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__)
#define REP64(...) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__)

__device__ __forceinline__ double readlane_f64(double v, int lane_uniform) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readfirst_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double bperm_f64(double v, int byte_addr) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ long long now() { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ __launch_bounds__(256) void k(double* out, long long* clk, double seed) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    lds[threadIdx.x] = seed + threadIdx.x;
    __syncthreads();
    if (wv != 0) { out[threadIdx.x] = lds[threadIdx.x]; return; }
    double x = seed + lane * 1e-3, y = 1.0 + 1e-9 * lane, z = 0.5;
    long long t0, t1;
    int slot = 0;
    // 0: dependent v_fma_f64
    t0 = now();
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 1: dependent v_mul_f64
    t0 = now();
    REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 2: dependent v_rcp_f64
    t0 = now();
    REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(x));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 3: independent v_fma_f64 (4 accumulators)
    {
        double a = x, b = x + 1, c = x + 2, d = x + 3;
        t0 = now();
        REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));)
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
        x = a + b + c + d;
    }
    // 4: readlane (2 x b32) -> fma with the SGPR pair -> readlane ...
    t0 = now();
    REP64(x = fma(y, readlane_f64(x, 5), x); asm volatile("" : "+v"(x));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 5: readlane alone, dependent through a v_mov from the SGPR (v_readlane -> v_mov_b32 -> v_readlane)
    {
        int xi = lane;
        t0 = now();
        REP64(asm volatile("v_readlane_b32 s20, %0, 5\n\tv_mov_b32 %0, s20" : "+v"(xi) :: "s20");)
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
        x += xi;
    }
    // 6: DPP row_shr:1 move of a double (2 x v_mov_b32_dpp) -> fma
    t0 = now();
    REP64(x = fma(y, dpp_f64<0x111>(x), x); asm volatile("" : "+v"(x));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 7: LDS round trip: ds_write_b64 -> ds_read_b64 (same address per lane) -> fma
    {
        const unsigned addr = (unsigned)(size_t)(lds + lane) & 0xffff;
        (void)addr;
        double* pl = lds + lane;
        t0 = now();
        REP16(asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_fma_f64 %0, %2, %0, %0" : "+v"(x) : "v"((unsigned)(size_t)pl), "v"(y) : "memory");)
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    }
    // 8: LDS broadcast read chain: ds_read_b64 of a uniform address -> fma (address independent of the data: pure latency with a wait each)
    {
        double* pl = lds + 7;
        double r;
        t0 = now();
        REP16(asm volatile("ds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_fma_f64 %0, %3, %1, %0" : "+v"(x), "=&v"(r) : "v"((unsigned)(size_t)pl), "v"(y) : "memory");)
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    }
    // 9: v_readfirstlane pair -> fma
    t0 = now();
    REP64(x = fma(y, readfirst_f64(x), x); asm volatile("" : "+v"(x));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 10: six independent readlane pairs then six dependent fma (the block back substitution's step)
    {
        t0 = now();
        REP16({ const double r1 = readlane_f64(x, 1), r2 = readlane_f64(x, 2), r3 = readlane_f64(x, 3), r4 = readlane_f64(x, 4), r5 = readlane_f64(x, 5), r6 = readlane_f64(x, 6);
                x = fma(y, r1, x); x = fma(y, r2, x); x = fma(y, r3, x); x = fma(y, r4, x); x = fma(y, r5, x); x = fma(y, r6, x); asm volatile("" : "+v"(x)); })
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    }
    // 11: ds_bpermute_b32 pair -> fma (arbitrary cross-lane gather through the LDS crossbar)
    {
        const int src = ((lane + 1) & 63) * 4;
        double r;
        t0 = now();
        REP16(r = bperm_f64(x, src); x = fma(y, r, x); asm volatile("" : "+v"(x));)
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    }
    // 12: v_permlane32_swap chain (gfx950): swap halves then fma
    t0 = now();
    REP64({ double a = x, b = z; swap_rows<32>(a, b); x = fma(y, b, a); z = b; asm volatile("" : "+v"(x), "+v"(z)); })
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 13: dependent v_add_f64
    t0 = now();
    REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));)
    t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    // 14: dependent v_fma_f32 (for scale)
    {
        float xf = (float)x, yf = 1.0001f, zf = 0.5f;
        t0 = now();
        REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf) : "v"(yf), "v"(zf));)
        t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
        x += xf;
    }
    // 15: empty s_memtime pair
    t0 = now(); t1 = now(); if (lane == 0) clk[slot] = t1 - t0; slot++;
    out[threadIdx.x] = x + z;
}

// barrier cost with four waves: N barriers back to back (every wave), and barrier + LDS hand-off wave 0 -> wave 1 -> wave 0
__global__ __launch_bounds__(256) void kb(double* out, long long* clk) {
    __shared__ double lds[256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = now();
    REP64(asm volatile("s_barrier" ::: "memory");)
    long long t1 = now();
    if (threadIdx.x == 0) clk[16] = t1 - t0;
    t0 = now();
    REP16(__syncthreads();)
    t1 = now();
    if (threadIdx.x == 0) clk[17] = t1 - t0;
    double x = lds[lane];
    t0 = now();
    for (int i = 0; i < 16; i++) {   // ping-pong: wave 0 writes, barrier, wave 1 reads + writes, barrier, wave 0 reads
        if (wv == 0) lds[lane] = x;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wv == 1) lds[64 + lane] = lds[lane] + 1.0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wv == 0) x = lds[64 + lane];
    }
    t1 = now();
    if (threadIdx.x == 0) clk[18] = t1 - t0;
    out[threadIdx.x] = x;
}

int main() {
    double* d_out; long long* d_clk;
    hipMalloc(&d_out, 4096); hipMalloc(&d_clk, 64 * 8); hipMemset(d_clk, 0, 64 * 8);
    for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d_out, d_clk, 1.0 + r); hipLaunchKernelGGL(kb, dim3(1), dim3(256), 0, 0, d_out, d_clk); }
    hipDeviceSynchronize();
    long long c[64]; hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_fma_f64", "dependent v_mul_f64", "dependent v_rcp_f64", "independent v_fma_f64 (4 chains)", "readlane x2 -> fma(sgpr) chain",
                           "readlane -> v_mov chain (b32)", "dpp row_shr x2 -> fma", "ds_write -> ds_read -> fma", "ds_read broadcast -> fma", "readfirstlane x2 -> fma",
                           "6 readlane pairs + 6 dependent fma", "ds_bpermute x2 -> fma", "permlane32_swap -> fma", "dependent v_add_f64", "dependent v_fma_f32", "empty s_memtime pair"};
    const int links[] = {64, 64, 64, 64, 64, 64, 64, 16, 16, 64, 16, 16, 64, 64, 64, 1};
    for (int i = 0; i < 16; i++) printf("%-40s %8lld ticks / %d links = %.2f per link\n", names[i], c[i], links[i], (double)c[i] / links[i]);
    printf("64 x s_barrier (4 waves)                 %8lld ticks = %.2f each\n", c[16], c[16] / 64.0);
    printf("16 x __syncthreads (4 waves)             %8lld ticks = %.2f each\n", c[17], c[17] / 16.0);
    printf("16 x LDS ping-pong wave0 -> wave1 -> wave0 %6lld ticks = %.2f per round trip (2 barriers, 2 writes, 2 reads)\n", c[18], c[18] / 16.0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("(s_memtime ticks; device clockRate %d kHz)\n", p.clockRate);
    return 0;
}
