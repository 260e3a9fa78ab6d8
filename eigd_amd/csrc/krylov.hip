// Short-recurrence form of sibk for positive definite shifts (gfx950).
//
// The reference solves (A - lam_i B) psi_i = b_i by Arnoldi on OP = P K F (F = factor, K = B or A) with full
// Gram-Schmidt against the Krylov history and a Hessenberg least-squares problem per step
// (eigd/eigenvector_derivatives.py:1246-1277).  With F symmetric positive definite, OP is self-adjoint in the F inner
// product <u, v>_F = u^T F v and C_i = I - alpha_i OP (alpha_i = +-(lam_i - sigma), 1264-1269) is positive definite there
// once every eigenpair with lam_j <= lam_i is deflated by P -- the N requested pairs are.  The Krylov space is the same;
// conjugate gradients in that inner product needs no history:
//
//     r = b,  zr = F r,  p = r,  zp = zr,  rho = r.zr
//     y = K zp                                       (SpMM, 1250-1252)
//     a = rho / (zp.p - alpha zp.y)                  (= rho / <p, C p>_F)
//     psi += a zp;   r -= a (p - alpha y);   r <- P r   (1257; |r| is the true Euclidean residual of 1275)
//     zr = F r                                       (the sweep, 1248)
//     rho' = r.zr;  b = rho'/rho;  p = r + b p;  zp = zr + b zp
//
// All modes advance in lock step as columns of n x k row-major blocks; every scalar above is a per-column number that
// never leaves the device: state[row][column], rows below.  A column is frozen (a = b = 0) from the step on in which its
// residual norm meets the tolerance, so its psi does not depend on how long its block mates run.
// The kernels are single streaming passes (HBM bound); cross-workgroup sums go through partial slabs that ONE workgroup
// adds in a fixed order and turns into the coefficients in the same launch.  No atomics: bitwise reproducible.
#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "common.h"

namespace eigd {

// rows of the per-column state block (leading dimension kMaxK)
enum CgRow { kRho = 0, kA = 1, kB = 2, kDone = 3, kTol2 = 4, kAlpha = 5, kSteps = 6, kFlag = 7, kDen = 8, kCgRows = 9 };

constexpr int kCgMaxBlocks = 1024;

static inline int cg_grid(int n, int rows_per_block) {
  const int64_t need = (static_cast<int64_t>(n) + rows_per_block - 1) / rows_per_block;
  return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(need, kCgMaxBlocks)));
}

// sum over g of partial[g * nout + o] by one wave: lane l adds g = l, l + 64, ... in ascending order, then a fixed
// shuffle tree (the association of reduce_partials_kernel in dense.hip)
__device__ __forceinline__ double wave_sum_partials(const double* __restrict__ partial, int nblocks, int nout, int o) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  int g = lane;
  for (; g + 7 * 64 < nblocks; g += 8 * 64) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = partial[static_cast<int64_t>(g + 64 * q) * nout + o];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; g < nblocks; g += 64) s += partial[static_cast<int64_t>(g) * nout + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return __shfl(s, 0, 64);
}

// partial sums of up to two column-wise dot products with a shared left factor: X.Y1 and X.Y2
template <int KP, int NQ>
__global__ __launch_bounds__(kThreads) void cg_dots_kernel(int n, int k, const double* __restrict__ X, int ldx,
                                                          const double* __restrict__ Y1, int ld1,
                                                          const double* __restrict__ Y2, int ld2,
                                                          double* __restrict__ partial) {
  constexpr int RP = kThreads / KP;
  __shared__ double red[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double s1 = 0.0, s2 = 0.0;
  if (c < k)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
      const double x = X[r * ldx + c];
      s1 += x * Y1[r * ld1 + c];
      if (NQ > 1) s2 += x * Y2[r * ld2 + c];
    }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    red[threadIdx.x] = (q == 0) ? s1 : s2;
    __syncthreads();
    if (rr == 0 && c < k) {
      double t = 0.0;
      for (int z = 0; z < RP; ++z) t += red[z * KP + c];
      partial[(static_cast<int64_t>(blockIdx.x) * NQ + q) * k + c] = t;
    }
    __syncthreads();
  }
}

// one workgroup: the partial sums of zp.p and zp.y -> a = rho / (zp.p - alpha zp.y) per column
__global__ __launch_bounds__(kThreads) void cg_alpha_kernel(const double* __restrict__ partial, int nblocks, int k,
                                                           double* __restrict__ state) {
  __shared__ double sums[2 * kMaxK];
  const int wave = threadIdx.x >> 6;
  for (int o = wave; o < 2 * k; o += kThreads / 64) {
    const double s = wave_sum_partials(partial, nblocks, 2 * k, o);
    if ((threadIdx.x & 63) == 0) sums[o] = s;
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= k) return;
  const double den = sums[c] - state[kAlpha * kMaxK + c] * sums[k + c];
  const double rho = state[kRho * kMaxK + c];
  const bool done = state[kDone * kMaxK + c] != 0.0;
  double a = 0.0;
  if (!done) {
    // <p, C p>_F must be positive: the operator is positive definite in the deflated space.  Anything else (an
    // eigenvalue below lam_i that is not deflated, an indefinite factor) is reported; the caller falls back to the
    // Arnoldi form.  rho == 0: the residual vanished exactly, nothing left to do for this column.
    if (den > 0.0 && rho >= 0.0)
      a = rho / den;
    else if (rho != 0.0)
      state[kFlag * kMaxK + c] = 1.0;
  }
  state[kA * kMaxK + c] = a;
  state[kDen * kMaxK + c] = den;
}

// psi += a zp;  r -= a (p - alpha y)
template <int KP>
__global__ __launch_bounds__(kThreads) void cg_update_kernel(int n, int k, double* __restrict__ psi, int ldpsi,
                                                            double* __restrict__ r, int ldr,
                                                            const double* __restrict__ zp, int ldzp,
                                                            const double* __restrict__ p, int ldp,
                                                            const double* __restrict__ y, int ldy,
                                                            const double* __restrict__ state) {
  constexpr int RP = kThreads / KP;
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  if (c >= k) return;
  const double a = state[kA * kMaxK + c];
  if (a == 0.0) return;  // frozen column: nothing moves (and whatever its work blocks hold is never read)
  const double al = state[kAlpha * kMaxK + c];
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * RP + rr; row < n; row += static_cast<int64_t>(gridDim.x) * RP) {
    const double zv = zp[row * ldzp + c], pv = p[row * ldp + c], yv = y[row * ldy + c];
    const double rv = r[row * ldr + c], sv = psi[row * ldpsi + c];
    psi[row * ldpsi + c] = sv + a * zv;
    r[row * ldr + c] = rv - a * (pv - al * yv);
  }
}

// one workgroup: |r|^2 against the tolerance -> done, steps; the partial sums of r.zr -> rho' ; b = rho' / rho
__global__ __launch_bounds__(kThreads) void cg_beta_kernel(const double* __restrict__ partial, int nblocks, int k,
                                                          const double* __restrict__ norm2, double* __restrict__ state,
                                                          int step, int first) {
  __shared__ double sums[kMaxK];
  const int wave = threadIdx.x >> 6;
  for (int o = wave; o < k; o += kThreads / 64) {
    const double s = wave_sum_partials(partial, nblocks, k, o);
    if ((threadIdx.x & 63) == 0) sums[o] = s;
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= k) return;
  bool done = state[kDone * kMaxK + c] != 0.0;
  if (!done && norm2 != nullptr && norm2[c] < state[kTol2 * kMaxK + c]) {  // reference 1275 (1223-1225 for step 0)
    done = true;
    state[kDone * kMaxK + c] = 1.0;
    state[kSteps * kMaxK + c] = static_cast<double>(step);
  }
  const double rho_new = sums[c], rho_old = state[kRho * kMaxK + c];
  double b = 0.0;
  if (!done && !first && rho_old > 0.0) b = rho_new / rho_old;
  if (!done) {
    if (rho_new < 0.0) state[kFlag * kMaxK + c] = 1.0;  // r^T F r < 0: the factor is not positive definite
    state[kRho * kMaxK + c] = rho_new;
  }
  state[kB * kMaxK + c] = b;
}

// p = r + b p;  zp = zr + b zp   (first step: b = 0, p and zp need not be initialised)
template <int KP>
__global__ __launch_bounds__(kThreads) void cg_direction_kernel(int n, int k, double* __restrict__ p, int ldp,
                                                               double* __restrict__ zp, int ldzp,
                                                               const double* __restrict__ r, int ldr,
                                                               const double* __restrict__ zr, int ldzr,
                                                               const double* __restrict__ state, int first) {
  constexpr int RP = kThreads / KP;
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  if (c >= k) return;
  if (state[kDone * kMaxK + c] != 0.0 && !first) return;  // frozen column
  const double b = state[kB * kMaxK + c];
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * RP + rr; row < n; row += static_cast<int64_t>(gridDim.x) * RP) {
    const double rv = r[row * ldr + c], zv = zr[row * ldzr + c];
    if (first || b == 0.0) {
      p[row * ldp + c] = rv;
      zp[row * ldzp + c] = zv;
    } else {
      p[row * ldp + c] = rv + b * p[row * ldp + c];
      zp[row * ldzp + c] = zv + b * zp[row * ldzp + c];
    }
  }
}

template <typename F>
static int cg_dispatch_kp(int k, F&& f) {
  const int kp = next_pow2(std::max(1, k));
  switch (kp) {
    case 1: f(std::integral_constant<int, 1>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 8: f(std::integral_constant<int, 8>()); break;
    case 16: f(std::integral_constant<int, 16>()); break;
    case 32: f(std::integral_constant<int, 32>()); break;
    case 64: f(std::integral_constant<int, 64>()); break;
    default: set_error("block wider than %d columns", kMaxK); return EIGD_E_INVALID;
  }
  return EIGD_OK;
}

}  // namespace eigd

using namespace eigd;

extern "C" {

int eigd_cg_state_rows(void) { return kCgRows; }

int eigd_cg_alpha(eigd_ctx* ctx, int n, int k, const double* dZp, int ldzp, const double* dP, int ldp, const double* dY,
                  int ldy, double* dState) {
  EIGD_REQUIRE(ctx && dZp && dP && dY && dState, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldzp >= k && ldp >= k && ldy >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = cg_grid(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * static_cast<size_t>(nb) * 2 * k);
  if (rc) return rc;
  double* partial = ctx->scratch;
  rc = cg_dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL((cg_dots_kernel<decltype(KP)::value, 2>), dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dZp, ldzp, dP,
                       ldp, dY, ldy, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  hipLaunchKernelGGL(cg_alpha_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, partial, nb, k, dState);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_cg_update(eigd_ctx* ctx, int n, int k, double* dPsi, int ldpsi, double* dR, int ldr, const double* dZp, int ldzp,
                   const double* dP, int ldp, const double* dY, int ldy, const double* dState) {
  EIGD_REQUIRE(ctx && dPsi && dR && dZp && dP && dY && dState, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldpsi >= k && ldr >= k && ldzp >= k && ldp >= k && ldy >= k,
               "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = cg_grid(n, (kThreads / kp) * 4);
  int rc = cg_dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(cg_update_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dPsi, ldpsi, dR,
                       ldr, dZp, ldzp, dP, ldp, dY, ldy, dState);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_cg_beta(eigd_ctx* ctx, int n, int k, const double* dR, int ldr, const double* dZr, int ldzr, const double* dNorm2,
                 double* dState, int step, int first) {
  EIGD_REQUIRE(ctx && dR && dZr && dState, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldr >= k && ldzr >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = cg_grid(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * static_cast<size_t>(nb) * k);
  if (rc) return rc;
  double* partial = ctx->scratch;
  rc = cg_dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL((cg_dots_kernel<decltype(KP)::value, 1>), dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dR, ldr, dZr,
                       ldzr, dZr, ldzr, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  hipLaunchKernelGGL(cg_beta_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, partial, nb, k, dNorm2, dState, step, first);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_cg_direction(eigd_ctx* ctx, int n, int k, double* dP, int ldp, double* dZp, int ldzp, const double* dR, int ldr,
                      const double* dZr, int ldzr, const double* dState, int first) {
  EIGD_REQUIRE(ctx && dP && dZp && dR && dZr && dState, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldp >= k && ldzp >= k && ldr >= k && ldzr >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = cg_grid(n, (kThreads / kp) * 4);
  int rc = cg_dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(cg_direction_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dP, ldp, dZp,
                       ldzp, dR, ldr, dZr, ldzr, dState, first);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

}  // extern "C"
