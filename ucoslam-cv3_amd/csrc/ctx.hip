// Context + error plumbing of the C ABI (include/ucoslam_hip.h).
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
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
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

void* uh_ctx_stream(uh_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

}  // extern "C"
