// Shared host-side plumbing for the HIP hot path (context, error reporting, scratch buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>
#include "../../include/ucoslam_hip.h"

namespace uh {

void set_error(const char* fmt, ...);

#define UH_HIP_CHECK(expr)                                                          \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            uh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                          __FILE__, __LINE__);                                      \
            return UH_ENODEVICE;                                                    \
        }                                                                           \
    } while (0)

#define UH_REQUIRE(cond, ...)                                                       \
    do {                                                                            \
        if (!(cond)) {                                                              \
            uh::set_error(__VA_ARGS__);                                             \
            return UH_EINVAL;                                                       \
        }                                                                           \
    } while (0)

// Growable device buffer (never shrinks). Not thread-safe; one owner.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    unsigned gen = 0;   // bumped by every (re)allocation: "same pointer" does not mean "same memory" (hipFree + hipMalloc may return the same base)
    int reserve(size_t bytes) {
        if (bytes <= cap) return UH_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return UH_ENOMEM; }
        cap = want;
        ++gen;
        return UH_OK;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Pinned host buffer for async D2H/H2D.
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return UH_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return UH_ENOMEM; }
        cap = want;
        return UH_OK;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
};


// Pinned host memory that the device addresses directly (hipHostMallocMapped): the latency-bound per-frame entry points hand their
// inputs to the kernel through such a block and receive the results in it — no copy engine, no stream synchronisation.
struct MappedBuf {
    void* h = nullptr;
    void* d = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return UH_OK;
        if (h) (void)hipHostFree(h);
        h = d = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&h, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) { set_error("hipHostMalloc(%zu, mapped) failed: %s", want, hipGetErrorString(e)); if (h) (void)hipHostFree(h); h = d = nullptr; return UH_ENOMEM; }
        std::memset(h, 0, want);
        cap = want;
        return UH_OK;
    }
    template <typename T> T* host() const { return reinterpret_cast<T*>(h); }
    template <typename T> T* dev() const { return reinterpret_cast<T*>(d); }
    ~MappedBuf() { if (h) (void)hipHostFree(h); }
    MappedBuf() = default;
    MappedBuf(const MappedBuf&) = delete;
    MappedBuf& operator=(const MappedBuf&) = delete;
};

// The device address of a host pointer the runtime knows as pinned (hipHostMalloc / hipHostRegister), NULL for pageable memory.
inline void* device_alias_of_host(const void* p) {
    if (!p) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // (pageable: "invalid value", not an error here)
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    return at.devicePointer;
}

// Wait for the completion word a kernel posts last (system-scope release store into pinned memory, behind its results): polling costs
// a few hundred nanoseconds after the store lands, a stream synchronisation 10-20 us.  A launch that disappears without posting
// (device fault) is noticed through hipStreamQuery; `what` names the caller in the error message.
inline int wait_host_word(volatile unsigned long long* word, unsigned long long expect, hipStream_t st, const char* what, int timeout_s = 30) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (*word == expect) break;
        __builtin_ia32_pause();
        if ((spin & 4095) == 4095) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady && *word != expect) {
                if (q == hipSuccess && *word == expect) break;
                set_error("%s: the launch ended without posting its completion word (%s)", what, hipGetErrorString(q == hipSuccess ? hipGetLastError() : q));
                return UH_ENODEVICE;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
                set_error("%s: no completion after %d s", what, timeout_s);
                return UH_ENODEVICE;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return UH_OK;
}

}  // namespace uh

// Optional per-kernel timing with HIP events recorded on the context stream (bench.py's roofline leg).
struct uh_prof {
    bool on = false;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<std::string> names;
    std::vector<double> total_ms;
    std::vector<long> calls;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    int id_of(const char* name) {
        for (size_t i = 0; i < names.size(); i++) if (names[i] == name) return (int)i;
        names.emplace_back(name); total_ms.push_back(0.0); calls.push_back(0);
        return (int)names.size() - 1;
    }
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

struct uh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 0;
    uh_prof prof;
};

namespace uh {
struct ProfScope {
    uh_ctx* c; int id = -1; hipEvent_t a = nullptr;
    ProfScope(uh_ctx* ctx, const char* name) : c(ctx) {
        if (c->prof.on) { id = c->prof.id_of(name); a = c->prof.get(); (void)hipEventRecord(a, c->stream); }
    }
    ~ProfScope() {
        if (id >= 0) { hipEvent_t b = c->prof.get(); (void)hipEventRecord(b, c->stream); c->prof.pending.push_back({id, a, b}); }
    }
};
}  // namespace uh

// launch `kernel` on the context stream; when profiling is on, bracket it with HIP events on that same stream
namespace uh {
// enqueue a one-thread launch that stores `word` (system-scope release) into pinned host memory behind everything already on the
// context stream: the host polls the word (wait_host_word) instead of synchronising the stream (ctx.hip)
int post_host_word(uh_ctx* ctx, unsigned long long* d_word_in_pinned_memory, unsigned long long word);
// 16-byte-wide copy launches on the context stream (sizes are rounded up to 16: the blocks involved are padded accordingly);
// publish16 = one workgroup: copy into pinned memory, then post the completion word; a device status word (optional) is copied to
// the 4 bytes at d_word + 8 and cleared
int copy16(uh_ctx* ctx, void* dst, const void* src, size_t bytes);
int publish16(uh_ctx* ctx, void* dst_pinned, const void* src, size_t bytes, unsigned* d_reset_word, unsigned long long* d_word_in_pinned_memory, unsigned long long word);
}  // namespace uh

namespace uh {
// uh_track_pose (track.hpp): the first solve's launch also takes the tracker's decision behind its last write — "enough inliers: the refined pose
// and the small disc, else the predicted pose and the wide radius" (system.cpp:6762-6881) — and leaves the local-map search its pose (rows +
// camera centre, 15 floats), radius (float 15) and a zero word (16) at dyn17, the chosen pose at pose_map (16 floats), the flag at tracked
struct PnpDecide { int min_inliers; float r_tracked, r_lost; float* dyn17; float* pose_map; int* tracked; };
}  // namespace uh

#define UH_LAUNCH(ctx, kernel, grid, block, shmem, ...)                                        \
    do {                                                                                        \
        uh::ProfScope _ps((ctx), #kernel);                                                      \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);             \
    } while (0)

static inline int uh_div_up(int a, int b) { return (a + b - 1) / b; }
