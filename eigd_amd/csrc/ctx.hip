// Context, memory and transfer entry points of the C ABI (include/eigd_hip.h).
#include <cstring>

#include "common.h"

namespace eigd {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}
const char* get_error() { return g_error.c_str(); }

}  // namespace eigd

int eigd_ctx::ensure_scratch(size_t bytes) {
  if (bytes <= scratch_bytes) return EIGD_OK;
  size_t want = bytes + bytes / 4 + (1 << 20);
  if (scratch) {
    EIGD_HIP(hipStreamSynchronize(stream));
    EIGD_HIP(hipFree(scratch));
    scratch = nullptr;
    scratch_bytes = 0;
  }
  EIGD_HIP(hipMalloc(reinterpret_cast<void**>(&scratch), want));
  scratch_bytes = want;
  return EIGD_OK;
}

int eigd_ctx::ensure_coef(size_t bytes) {
  if (bytes <= coef_bytes) return EIGD_OK;
  size_t want = bytes + (1 << 16);
  if (coef) {
    EIGD_HIP(hipStreamSynchronize(stream));
    EIGD_HIP(hipFree(coef));
    coef = nullptr;
    coef_bytes = 0;
  }
  EIGD_HIP(hipMalloc(reinterpret_cast<void**>(&coef), want));
  coef_bytes = want;
  return EIGD_OK;
}

extern "C" {

const char* eigd_last_error(void) { return eigd::get_error(); }
int eigd_version(void) { return 100; }

int eigd_device_count(int* count) {
  EIGD_REQUIRE(count != nullptr, "count is null");
  *count = 0;
  EIGD_HIP(hipGetDeviceCount(count));
  return EIGD_OK;
}

int eigd_ctx_create(int device, eigd_ctx** out) {
  EIGD_REQUIRE(out != nullptr, "out is null");
  *out = nullptr;
  int count = 0;
  EIGD_HIP(hipGetDeviceCount(&count));
  EIGD_REQUIRE(device >= 0 && device < count, "device %d out of range (%d visible)", device, count);
  EIGD_HIP(hipSetDevice(device));
  eigd_ctx* c = new eigd_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  EIGD_HIP(hipGetDeviceProperties(&prop, device));
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  EIGD_HIP(hipStreamCreate(&c->stream));
  EIGD_HIP(hipEventCreate(&c->ev0));
  EIGD_HIP(hipEventCreate(&c->ev1));
  *out = c;
  return EIGD_OK;
}

int eigd_ctx_fork(eigd_ctx* parent, eigd_ctx** out) {
  EIGD_REQUIRE(parent && out, "null argument");
  return eigd_ctx_create(parent->device, out);
}

int eigd_ctx_destroy(eigd_ctx* ctx) {
  if (!ctx) return EIGD_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->coef) (void)hipFree(ctx->coef);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev_pinned) (void)hipEventDestroy(ctx->ev_pinned);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_h) (void)hipHostFree(ctx->pinned_h);
  if (ctx->bounce) (void)hipHostFree(ctx->bounce);
  if (ctx->proj_stats) (void)hipFree(ctx->proj_stats);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return EIGD_OK;
}

int eigd_ctx_make_current(eigd_ctx* ctx) {
  EIGD_REQUIRE(ctx, "ctx is null");
  EIGD_HIP(hipSetDevice(ctx->device));  // HIP's current device is per host thread: a new thread starts on device 0
  return EIGD_OK;
}

int eigd_sync(eigd_ctx* ctx) {
  EIGD_REQUIRE(ctx, "ctx is null");
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  return EIGD_OK;
}

int eigd_malloc(eigd_ctx* ctx, size_t bytes, void** dptr) {
  EIGD_REQUIRE(ctx && dptr, "null argument");
  *dptr = nullptr;
  if (bytes == 0) bytes = 8;
  EIGD_HIP(hipSetDevice(ctx->device));
  EIGD_HIP(hipMalloc(dptr, bytes));
  return EIGD_OK;
}

int eigd_free(eigd_ctx* ctx, void* dptr) {
  EIGD_REQUIRE(ctx, "ctx is null");
  if (!dptr) return EIGD_OK;
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  EIGD_HIP(hipFree(dptr));
  return EIGD_OK;
}

int eigd_memset(eigd_ctx* ctx, void* dptr, int value, size_t bytes) {
  EIGD_REQUIRE(ctx && dptr, "null argument");
  EIGD_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return EIGD_OK;
}

// Small copies (coefficient blocks, state rows, flags: the host reads and writes dozens of them per solve) go through
// 64 KB of page-locked memory owned by the context: a copy between the device and PAGEABLE memory is staged by the
// runtime with a wait of its own (30 to 200 us for a few hundred bytes); the DMA into page-locked memory and a memcpy
// take 10.  A context is driven by one host thread at a time (forked contexts for worker threads): no lock.
constexpr size_t kBounceBytes = 64 * 1024;
static int bounce_of(eigd_ctx* ctx) {
  if (ctx->bounce == nullptr) EIGD_HIP(hipHostMalloc(&ctx->bounce, kBounceBytes, hipHostMallocDefault));
  return EIGD_OK;
}

int eigd_h2d(eigd_ctx* ctx, void* dptr, const void* hsrc, size_t bytes) {
  EIGD_REQUIRE(ctx && dptr && hsrc, "null argument");
  if (bytes > 0 && bytes <= kBounceBytes && bounce_of(ctx) == EIGD_OK) {
    std::memcpy(ctx->bounce, hsrc, bytes);
    EIGD_HIP(hipMemcpyAsync(dptr, ctx->bounce, bytes, hipMemcpyHostToDevice, ctx->stream));
    EIGD_HIP(hipStreamSynchronize(ctx->stream));  // (the bounce buffer is free again)
    return EIGD_OK;
  }
  EIGD_HIP(hipMemcpyAsync(dptr, hsrc, bytes, hipMemcpyHostToDevice, ctx->stream));
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  return EIGD_OK;
}

int eigd_d2h(eigd_ctx* ctx, void* hdst, const void* dptr, size_t bytes) {
  EIGD_REQUIRE(ctx && dptr && hdst, "null argument");
  if (bytes > 0 && bytes <= kBounceBytes && bounce_of(ctx) == EIGD_OK) {
    EIGD_HIP(hipMemcpyAsync(ctx->bounce, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    EIGD_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(hdst, ctx->bounce, bytes);
    return EIGD_OK;
  }
  EIGD_HIP(hipMemcpyAsync(hdst, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  return EIGD_OK;
}

// Page-locked host memory for the numpy-facing call surface (reference callers hand numpy arrays to solve_adjoint /
// add_total_derivative and get numpy arrays back, eigenvector_derivatives.py:1988-2134, 2167-2207): a copy from or to
// pageable memory is staged by the runtime at roughly half the PCIe rate; eigd_h2d / eigd_d2h take the direct DMA path
// by themselves when the host pointer is page-locked.
int eigd_host_alloc(size_t bytes, void** hptr) {
  EIGD_REQUIRE(hptr != nullptr, "hptr is null");
  *hptr = nullptr;
  if (bytes == 0) bytes = 8;
  EIGD_HIP(hipHostMalloc(hptr, bytes, hipHostMallocPortable));
  return EIGD_OK;
}

int eigd_host_free(void* hptr) {
  if (!hptr) return EIGD_OK;
  EIGD_HIP(hipHostFree(hptr));
  return EIGD_OK;
}

int eigd_host_register(void* hptr, size_t bytes) {
  EIGD_REQUIRE(hptr != nullptr && bytes > 0, "null or empty host range");
  EIGD_HIP(hipHostRegister(hptr, bytes, hipHostRegisterPortable));
  return EIGD_OK;
}

int eigd_host_unregister(void* hptr) {
  EIGD_REQUIRE(hptr != nullptr, "hptr is null");
  EIGD_HIP(hipHostUnregister(hptr));
  return EIGD_OK;
}

int eigd_d2d(eigd_ctx* ctx, void* ddst, const void* dsrc, size_t bytes) {
  EIGD_REQUIRE(ctx && ddst && dsrc, "null argument");
  EIGD_HIP(hipMemcpyAsync(ddst, dsrc, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return EIGD_OK;
}

int eigd_mem_info(eigd_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  EIGD_REQUIRE(ctx && free_bytes && total_bytes, "null argument");
  EIGD_HIP(hipSetDevice(ctx->device));
  EIGD_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return EIGD_OK;
}

int eigd_timer_start(eigd_ctx* ctx) {
  EIGD_REQUIRE(ctx, "ctx is null");
  EIGD_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return EIGD_OK;
}

int eigd_timer_stop_ms(eigd_ctx* ctx, double* ms) {
  EIGD_REQUIRE(ctx && ms, "null argument");
  EIGD_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  EIGD_HIP(hipEventSynchronize(ctx->ev1));
  float f = 0.f;
  EIGD_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
  *ms = f;
  return EIGD_OK;
}

}  // extern "C"
