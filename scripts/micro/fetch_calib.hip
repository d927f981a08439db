// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this repository's kernels (MI355X_MICROARCH.md: the counter
// reports half the bytes of a 16-byte-per-lane coalesced stream; "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Four kernels over a 512 MiB buffer (past L2 and the 256 MiB Infinity Cache), each with a known byte count:
//   stream16    16 bytes per lane, coalesced                      -> every byte of the buffer once
//   stream4     4 bytes per lane, coalesced                       -> every byte once
//   gather1     one BYTE per lane at a pseudo-random 128-byte line (describe_kernel's pattern, but cold): N distinct lines
//   gather1_hot one byte per lane inside a 1 MiB window (L2-resident after the first touch: what describe_kernel does on a 0.5 MB pyramid)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calib.hip -o scripts/micro/fetch_calib.bin
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d out -- scripts/micro/fetch_calib.bin     (scripts/micro/run_fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void stream16(const uint4* __restrict__ p, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream4(const unsigned* __restrict__ p, size_t n4, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// lane t of the grid reads byte (t * 128 * stride) % bytes + (t & 127): one byte of its own 128-byte line, lines visited in a strided order
__global__ void gather1(const unsigned char* __restrict__ p, size_t bytes, size_t n_acc, size_t line_stride, unsigned* sink) {
    unsigned acc = 0;
    const size_t nlines = bytes / 128;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_acc; t += (size_t)gridDim.x * blockDim.x) acc += p[((t * line_stride) % nlines) * 128 + (t & 127)];
    if (acc == 0x12345678u) *sink = acc;
}
int main() {
    const size_t bytes = 512ull << 20;
    unsigned char* d; unsigned* sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 4);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8, block = 256;
    hipLaunchKernelGGL(stream16, dim3(grid), dim3(block), 0, 0, (const uint4*)d, bytes / 16, sink);
    hipLaunchKernelGGL(stream4, dim3(grid), dim3(block), 0, 0, (const unsigned*)d, bytes / 4, sink);
    const size_t n_acc = 2u << 20;   // 2 Mi byte reads, each in its own 128-byte line (4 Mi lines in the buffer; stride 2 -> every other line)
    hipLaunchKernelGGL(gather1, dim3(grid), dim3(block), 0, 0, d, bytes, n_acc, (size_t)2, sink);
    hipDeviceSynchronize();
    // hot window: 1 MiB = 8192 lines, 2 Mi byte reads -> 256 reads per line; first a warm-up pass (its own dispatch), then the measured one
    hipLaunchKernelGGL(gather1, dim3(grid), dim3(block), 0, 0, d, (size_t)1 << 20, n_acc, (size_t)1, sink);
    hipLaunchKernelGGL(gather1, dim3(grid), dim3(block), 0, 0, d, (size_t)1 << 20, n_acc, (size_t)1, sink);
    hipDeviceSynchronize();
    printf("stream16 %zu bytes; stream4 %zu bytes; gather1 cold: %zu byte reads in %zu distinct 128-byte lines (= %zu bytes at 64 B per request, %zu at 128); gather1 hot: %zu byte reads inside 1 MiB, twice\n",
           bytes, bytes, n_acc, n_acc, n_acc * 64, n_acc * 128, n_acc);
    return 0;
}
