// Shared host-side plumbing for the HIP hot path (context, error reporting, scratch buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
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
    int reserve(size_t bytes) {
        if (bytes <= cap) return UH_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return UH_ENOMEM; }
        cap = want;
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

}  // namespace uh

struct uh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 0;
};

static inline int uh_div_up(int a, int b) { return (a + b - 1) / b; }
