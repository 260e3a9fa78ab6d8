// Shared host-side plumbing of libeigd_hip.so: context object, error reporting,
// scratch management.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/eigd_hip.h"

namespace eigd {

void set_error(const char* fmt, ...);
const char* get_error();

#define EIGD_HIP(call)                                                                           \
  do {                                                                                           \
    hipError_t _e = (call);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      ::eigd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return EIGD_E_HIP;                                                                         \
    }                                                                                            \
  } while (0)

#define EIGD_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      ::eigd::set_error(__VA_ARGS__); \
      return EIGD_E_INVALID;         \
    }                                \
  } while (0)

// launch check: catches bad configurations at the launch site
#define EIGD_LAUNCH_CHECK() EIGD_HIP(hipGetLastError())

constexpr int kThreads = 256;
constexpr int kMaxK = 64;  // widest dense block a panel kernel handles in one call

inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace eigd

struct eigd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // device scratch: partial sums / small coefficient matrices
  double* scratch = nullptr;
  size_t scratch_bytes = 0;
  // second scratch (kept separate so a coefficient upload never aliases partial sums)
  double* coef = nullptr;
  size_t coef_bytes = 0;
  int n_cu = 256;
  // pinned host staging for small results whose D2H copy is issued in stream order and collected later
  // (eigd_colnorm2_dev / eigd_colnorm2_fetch): the copy completes behind the kernel that made the numbers,
  // not behind whatever the host enqueues afterwards
  double* pinned = nullptr;   // 2 * kMaxK doubles
  hipEvent_t ev_pinned = nullptr;
  int pinned_count = 0;
  void* bounce = nullptr;      // 64 KB of page-locked memory: small copies from / to pageable host memory go through it
  double* pinned_h = nullptr;  // pinned staging of coefficient blocks (eigd_stack_cgs2)
  size_t pinned_h_bytes = 0;
  // eigd_project_norm2: {projections measured, of those: updates applied} since the last eigd_project_stats
  int* proj_stats = nullptr;

  int ensure_scratch(size_t bytes);
  int ensure_coef(size_t bytes);
};

namespace eigd {
// small device results for the host: copy in stream order behind their kernel, collected by eigd_colnorm2_fetch (dense.hip)
int publish_norm2(eigd_ctx* ctx, const double* dOut, int k);
// krylov.hip: the coefficient kernel of the short-recurrence sibk over partial sums [nblocks][2 k] left by another kernel
int cg_coefficients_from_partials(eigd_ctx* ctx, const double* partial, int nblocks, int k, const double* dNorm2,
                                  double* dState, int step, int first, double* dLog);
}  // namespace eigd
