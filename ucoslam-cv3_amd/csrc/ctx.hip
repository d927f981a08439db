// Context + error plumbing of the C ABI (include/ucoslam_hip.h).
#include <algorithm>
#include <string>
#include <vector>
#include "common.hpp"

namespace uh {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace uh

__global__ void uh_post_word_kernel(unsigned long long* host_done, unsigned long long word) {
    __hip_atomic_store(host_done, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// 16-byte-wide copies between pinned host memory and HBM as plain launches on the context stream (no copy engine, no runtime staging):
// every lane one 16-byte load + store; `publish` is the last launch of a host-pointer call — one workgroup copies the (small) result
// block into pinned memory, clears the device-side status word for the next call and posts the completion word behind the data.
__global__ __launch_bounds__(256) void uh_copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
__global__ __launch_bounds__(1024) void uh_publish_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, unsigned* reset_word,
                                                          unsigned long long* host_done, unsigned long long word) {
    for (size_t i = threadIdx.x; i < n16; i += 1024) dst[i] = src[i];
    // Every thread releases its own stores at system scope BEFORE the barrier (as pnp_solve_kernel does): a workgroup barrier alone does
    // not make the other waves drain their outstanding stores to pinned memory, and thread 0's release below only covers its own wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (reset_word) {   // the call's device-side status word travels in the header (behind the completion word) and is cleared for the next call
            reinterpret_cast<unsigned*>(host_done)[2] = *reset_word;
            *reset_word = 0u;
        }
        __hip_atomic_store(host_done, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

namespace uh {
int copy16(uh_ctx* ctx, void* dst, const void* src, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    if (!n16) return UH_OK;
    UH_LAUNCH(ctx, uh_copy16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}
int publish16(uh_ctx* ctx, void* dst_pinned, const void* src, size_t bytes, unsigned* d_reset_word, unsigned long long* d_word, unsigned long long word) {
    UH_LAUNCH(ctx, uh_publish_kernel, dim3(1), dim3(1024), 0, static_cast<const uint4*>(src), static_cast<uint4*>(dst_pinned), (bytes + 15) / 16, d_reset_word, d_word, word);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}
int post_host_word(uh_ctx* ctx, unsigned long long* d_word, unsigned long long word) {
    UH_LAUNCH(ctx, uh_post_word_kernel, dim3(1), dim3(1), 0, d_word, word);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}
}  // namespace uh

extern "C" {

const char* uh_last_error(void) { return uh::g_err; }
int uh_version(void) { return 100; }

static int ctx_create(int device, void* hip_stream, bool private_stream, uh_ctx** out) {
    UH_REQUIRE(out != nullptr, "uh_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        uh::set_error("uh_ctx_create: no HIP device available (%s); this library has no CPU path",
                      e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return UH_ENODEVICE;
    }
    UH_REQUIRE(device >= 0 && device < ndev, "uh_ctx_create: device %d out of range [0,%d)", device, ndev);
    UH_HIP_CHECK(hipSetDevice(device));
    uh_ctx* c = new uh_ctx();
    c->device = device;
    if (!private_stream) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = default stream
        c->owns_stream = false;
    } else {
        // HIP multiplexes the streams of one priority onto GPU_MAX_HW_QUEUES (4) hardware queues, and two streams that land
        // on the same queue serialise (measured: local BA next to tracking 0.97 -> 1.20 ms per step once RCCL's streams had
        // shifted the assignment).  A private stream exists to run BESIDE the caller's stream, so it is taken from the
        // high-priority queue pool, which the caller's default-priority streams never share.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
        if (e != hipSuccess) { uh::set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return UH_ENODEVICE; }
        c->owns_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    *out = c;
    return UH_OK;
}

int uh_ctx_create(int device, void* hip_stream, uh_ctx** out) { return ctx_create(device, hip_stream, false, out); }
int uh_ctx_create_private(int device, uh_ctx** out) { return ctx_create(device, nullptr, true, out); }

// A private stream confined to a share of the compute units (hipExtStreamCreateWithCUMask).  On MI355X bit p of the mask is CU p / 8 of
// XCD p % 8 (scripts/micro/cu_mask_probe.hip), so a contiguous bit range [first_bit, first_bit + n_bits) is the same number of CUs on
// every XCD; a mask that leaves an XCD empty is ignored by the runtime, so a range shorter than the XCD count is refused.  The
// context reports n_bits as its CU count (the persistent local BA sizes its co-resident grid from it).
// UNLIKE uh_ctx_create_private this stream has DEFAULT priority and is BLOCKING (hipExtStreamCreateWithCUMask offers neither flag): it
// synchronises implicitly with the NULL stream and shares the default hardware-queue pool, so keep NULL-stream work away from it.
// The bit layout (one bit per CU, XCD = bit % 8) was probed on gfx950 only: other devices are refused.
int uh_ctx_create_private_cus(int device, int first_bit, int n_bits, uh_ctx** out) {
    UH_REQUIRE(out != nullptr, "uh_ctx_create_private_cus: out is NULL");
    int rc = ctx_create(device, nullptr, false, out);   // (device checks; the stream is made below)
    if (rc) return rc;
    uh_ctx* c = *out;
    const int total = c->num_cus;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            uh::set_error("uh_ctx_create_private_cus: the CU-mask bit layout is only validated on gfx950 (device reports %s)", prop.gcnArchName);
            delete c; *out = nullptr; return UH_EINVAL;
        }
    }
    if (!(first_bit >= 0 && n_bits >= 8 && first_bit + n_bits <= total)) {
        uh::set_error("uh_ctx_create_private_cus: bits [%d, %d) outside the device's %d compute units (or fewer than 8)", first_bit, first_bit + n_bits, total);
        delete c; *out = nullptr; return UH_EINVAL;
    }
    std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
    for (int p = first_bit; p < first_bit + n_bits; p++) mask[p >> 5] |= 1u << (p & 31);
    const hipError_t e = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { uh::set_error("hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e)); delete c; *out = nullptr; return UH_ENODEVICE; }
    c->owns_stream = true;
    c->num_cus = n_bits;
    return UH_OK;
}

void uh_ctx_destroy(uh_ctx* ctx) {
    if (!ctx) return;
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int uh_ctx_synchronize(uh_ctx* ctx) {
    UH_REQUIRE(ctx != nullptr, "uh_ctx_synchronize: ctx is NULL");
    UH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return UH_OK;
}

int uh_prof_enable(uh_ctx* ctx, int on) {
    UH_REQUIRE(ctx != nullptr, "uh_prof_enable: ctx is NULL");
    ctx->prof.on = on != 0;
    return UH_OK;
}

static void prof_drain(uh_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->prof.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { ctx->prof.total_ms[r.id] += ms; ctx->prof.calls[r.id]++; }
        ctx->prof.pool.push_back(r.a);
        ctx->prof.pool.push_back(r.b);
    }
    ctx->prof.pending.clear();
}

int uh_prof_reset(uh_ctx* ctx) {
    UH_REQUIRE(ctx != nullptr, "uh_prof_reset: ctx is NULL");
    prof_drain(ctx);
    for (auto& v : ctx->prof.total_ms) v = 0.0;
    for (auto& v : ctx->prof.calls) v = 0;
    return UH_OK;
}

// text report, one line per kernel: "<name> <calls> <total_ms>\n"; returns bytes needed (incl. NUL)
int uh_prof_report(uh_ctx* ctx, char* buf, size_t cap) {
    UH_REQUIRE(ctx != nullptr, "uh_prof_report: ctx is NULL");
    prof_drain(ctx);
    std::string out;
    char line[256];
    for (size_t i = 0; i < ctx->prof.names.size(); i++) {
        snprintf(line, sizeof(line), "%s %ld %.6f\n", ctx->prof.names[i].c_str(), ctx->prof.calls[i], ctx->prof.total_ms[i]);
        out += line;
    }
    if (buf && cap) { size_t n = std::min(cap - 1, out.size()); memcpy(buf, out.data(), n); buf[n] = 0; }
    return (int)out.size() + 1;
}

// Measurement hook (scripts/time_interference.py): a background kernel of a chosen character, to see WHAT about a neighbouring
// launch slows the latency-bound BA chain down.  mode 0: waves that only sleep (occupy wave slots, nothing else); 1: dependent
// integer VALU work; 2: streaming loads over `buf` (L1/L2/TA traffic); 3: LDS traffic.
__global__ __launch_bounds__(256) void uh_debug_background_kernel(int mode, int iters, const unsigned int* buf, unsigned int words, unsigned int* sink) {
    __shared__ unsigned int s_l[1024];
    unsigned int acc = threadIdx.x;
    if (mode == 0) {
        for (int i = 0; i < iters; i++) __builtin_amdgcn_s_sleep(64);
    } else if (mode == 1) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) acc = acc * 1664525u + 1013904223u;
        }
    } else if (mode == 2) {
        unsigned int idx = (blockIdx.x * 256 + threadIdx.x) * 4;
        for (int i = 0; i < iters; i++) {
            const uint4 v = *reinterpret_cast<const uint4*>(buf + (idx % (words - 4)));
            acc += v.x ^ v.y ^ v.z ^ v.w;
            idx += 256 * 4 * 97;
        }
    } else {
        s_l[threadIdx.x] = acc; s_l[threadIdx.x + 256] = acc; s_l[threadIdx.x + 512] = acc; s_l[threadIdx.x + 768] = acc;
        __syncthreads();
        for (int i = 0; i < iters; i++) acc += s_l[(acc + i) & 1023];
    }
    if (acc == 0x12345678u) *sink = acc;
}

int uh_debug_background(uh_ctx* ctx, int mode, int blocks, int iters, const void* d_buf, size_t buf_bytes, void* d_sink) {
    UH_REQUIRE(ctx && blocks >= 1 && iters >= 0 && d_sink, "uh_debug_background: bad arguments");
    UH_HIP_CHECK(hipSetDevice(ctx->device));
    UH_LAUNCH(ctx, uh_debug_background_kernel, dim3(blocks), dim3(256), 0, mode, iters, (const unsigned int*)d_buf, (unsigned int)(buf_bytes / 4), (unsigned int*)d_sink);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

void* uh_ctx_stream(uh_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

// Pinned host memory for the buffers a host hands to the host-pointer entry points (frames in, keypoints / rows out): copies from / to it
// are true asynchronous DMA; pageable memory works everywhere too, through the runtime's staging
void* uh_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { uh::set_error("uh_host_alloc(%zu): hipHostMalloc failed", bytes); return nullptr; }
    return p;
}
void uh_host_free(void* p) { if (p) (void)hipHostFree(p); }

}  // extern "C"
