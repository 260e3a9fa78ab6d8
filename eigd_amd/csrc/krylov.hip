// Short-recurrence form of sibk for positive definite shifts (gfx950).
//
// The reference solves (A - lam_i B) psi_i = b_i by Arnoldi on OP = P K F (F = factor, K = B or A) with full
// Gram-Schmidt against the Krylov history and a Hessenberg least-squares problem per step
// (eigd/eigenvector_derivatives.py:1246-1277).  With F symmetric positive definite, OP is self-adjoint in the F inner
// product <u, v>_F = u^T F v and C_i = I - alpha_i OP (alpha_i = +-(lam_i - sigma), 1264-1269) is positive definite there
// once every eigenpair with lam_j <= lam_i is deflated by P -- the N requested pairs are.  The Krylov space is the same;
// conjugate gradients in that inner product needs no history.  The iterates are formed by the three-term recurrences
// (Rutishauser's form of CG: residual and solution directly, no direction vectors -- 11 streaming passes over the work
// blocks per step where the two-term form with p, F p needs 18; same iterates, measured: same step counts and psi):
//
//     z = F r_k                                         (the sweep, 1248)
//     y = K z                                           (SpMM, 1250-1252)
//     rr = r_k.z,  gam = rr / (rr - alpha z.y)          (= <r,r>_F / <r, C r>_F)
//     rho = 1 / (1 - (gam/gam') (rr/rr') / rho')        (primes: previous step; rho = 1 in the first)
//     r_{k+1}   = rho (r_k - gam (r_k - alpha y)) + (1 - rho) r_{k-1};   r_{k+1} <- P r_{k+1}   (1257)
//     psi_{k+1} = rho (psi_k + gam z)             + (1 - rho) psi_{k-1}
//
// The solution recurrence need not run at all: with alpha_k = rho_k gam_k and beta_k = (rho_{k+1} - 1) alpha_k / alpha_{k+1}
// it is the two-term form psi_{k+1} = psi_k + alpha_k p_k, p_k = z_k + beta_{k-1} p_{k-1}, so the final psi is a combination
// of the z_k with POSITIVE coefficients s_k = alpha_k + beta_k s_{k+1}.  The caller keeps every step's z (the sweep writes
// it into a slab of a history stack instead of a work block: no extra traffic), passes no psi blocks to eigd_cg_update
// (four streaming passes per step instead of eight) and forms psi once at the end from the (gam, rho) log below.
//
// |r_{k+1}| is the true Euclidean residual norm of 1275.  All modes advance in lock step as columns of n x k row-major
// blocks; every scalar above is a per-column number that never leaves the device: state[row][column], rows below.  A
// column is frozen from the step on in which its residual norm meets the tolerance, so its psi does not depend on how
// long its block mates run.  The kernels are single streaming passes (HBM bound); cross-workgroup sums go through partial
// slabs that ONE workgroup adds in a fixed order and turns into the coefficients in the same launch.  No atomics:
// bitwise reproducible.
#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "common.h"

namespace eigd {

// rows of the per-column state block (leading dimension kMaxK)
enum CgRow { kRr = 0, kGam = 1, kRho = 2, kDone = 3, kTol2 = 4, kAlpha = 5, kSteps = 6, kFlag = 7, kGamNow = 8, kRhoNow = 9,
             kBadRr = 10, kBadDen = 11, kBadStep = 12,   // what a breakdown (flag 2) saw: r.z, r.z - alpha z.y, the step
             kCgRows = 13 };

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

// partial sums of z.r and z.y, one pass over the three blocks
template <int KP>
__global__ __launch_bounds__(kThreads) void cg_dots_kernel(int n, int k, const double* __restrict__ Z, int ldz,
                                                          const double* __restrict__ R, int ldr,
                                                          const double* __restrict__ Y, int ldy,
                                                          double* __restrict__ partial) {
  constexpr int RP = kThreads / KP;
  __shared__ double red[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double s1 = 0.0, s2 = 0.0;
  if (c < k)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
      const double z = Z[r * ldz + c];
      s1 += z * R[r * ldr + c];
      s2 += z * Y[r * ldy + c];
    }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    red[threadIdx.x] = (q == 0) ? s1 : s2;
    __syncthreads();
    if (rr == 0 && c < k) {
      double t = 0.0;
      for (int z = 0; z < RP; ++z) t += red[z * KP + c];
      partial[(static_cast<int64_t>(blockIdx.x) * 2 + q) * k + c] = t;
    }
    __syncthreads();
  }
}

// one workgroup of two waves per column: |r_k|^2 against the tolerance -> done, step; the partial sums -> gam, rho of
// this step (a single workgroup walking all 2 k sums one after the other took 32 us per step at 32 columns)
__global__ __launch_bounds__(128) void cg_coef_kernel(const double* __restrict__ partial, int nblocks, int k,
                                                     const double* __restrict__ norm2, double* __restrict__ state,
                                                     int step, int first, double* __restrict__ log) {
  __shared__ double sums[2 * kMaxK];
  const int c = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  {
    const int o = wave * k + c;   // wave 0: r.z of this column, wave 1: z.y
    const double s = wave_sum_partials(partial, nblocks, 2 * k, o);
    if ((threadIdx.x & 63) == 0) sums[o] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  bool done = state[kDone * kMaxK + c] != 0.0;
  if (!done && norm2 != nullptr && norm2[c] < state[kTol2 * kMaxK + c]) {  // reference 1275: the residual this step started from
    done = true;
    state[kDone * kMaxK + c] = 1.0;
    state[kSteps * kMaxK + c] = static_cast<double>(step - 1);
  }
  double gam = 0.0, rho = 1.0;
  if (!done) {
    const double rr = sums[c];
    const double den = rr - state[kAlpha * kMaxK + c] * sums[k + c];  // <r, C r>_F
    // rr = r^T F r and <r, C r>_F must be positive: F is positive definite and so is C in the deflated space.  Anything
    // else (an eigenvalue below lam_i that is not deflated, an indefinite factor) is a breakdown (flag 2): the column
    // stops moving and the caller falls back to the Arnoldi form.  rr == 0: the residual vanished exactly, nothing left to
    // do for this column.
    if (rr > 0.0 && den > 0.0) {
      gam = rr / den;
      if (!first) {
        // q <= 0 happens in finite precision when C_i is nearly singular in the deflated space (a pair just above lam_i
        // that is not deflated) and r^T F r has grown by orders between two steps: the step is then taken with rho = 1 -- a
        // restart of the recurrence from the current iterate; r and psi stay consistent for any (gam, rho), so the
        // residual test decides as before.  Counted (flag 1), not an error.
        const double q = 1.0 - (gam / state[kGam * kMaxK + c]) * (rr / state[kRr * kMaxK + c]) / state[kRho * kMaxK + c];
        if (q > 0.0)
          rho = 1.0 / q;
        else if (state[kFlag * kMaxK + c] == 0.0)
          state[kFlag * kMaxK + c] = 1.0;
      }
      state[kRr * kMaxK + c] = rr;
      state[kGam * kMaxK + c] = gam;
      state[kRho * kMaxK + c] = rho;
    } else if (rr != 0.0) {
      if (state[kFlag * kMaxK + c] != 2.0) {
        state[kBadRr * kMaxK + c] = rr;
        state[kBadDen * kMaxK + c] = den;
        state[kBadStep * kMaxK + c] = static_cast<double>(step);
      }
      state[kFlag * kMaxK + c] = 2.0;
    }
  }
  state[kGamNow * kMaxK + c] = gam;  // gam == 0: the column does not move in this step
  state[kRhoNow * kMaxK + c] = rho;
  if (log != nullptr) {              // (gam, rho) of every step, for the caller that forms psi from the z history
    log[(static_cast<int64_t>(step) - 1) * 2 * kMaxK + c] = gam;
    log[(static_cast<int64_t>(step) - 1) * 2 * kMaxK + kMaxK + c] = rho;
  }
}

// r_{k+1} = rho (r - gam (r - alpha y)) + (1 - rho) r_old  and  psi_{k+1} = rho (psi + gam z) + (1 - rho) psi_old, written
// over r_old / psi_old (the caller swaps the roles of the buffers); columns that do not move are copied
template <int KP, bool PSI>
__global__ __launch_bounds__(kThreads) void cg_update_kernel(int n, int k, const double* __restrict__ r, int ldr,
                                                            double* __restrict__ ro, int ldro,
                                                            const double* __restrict__ psi, int ldpsi,
                                                            double* __restrict__ pso, int ldpso,
                                                            const double* __restrict__ z, int ldz,
                                                            const double* __restrict__ y, int ldy,
                                                            const double* __restrict__ state, int first,
                                                            double* __restrict__ normpart) {
  constexpr int RP = kThreads / KP;
  __shared__ double red[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double nrm = 0.0;
  if (c < k) {
  const double gam = state[kGamNow * kMaxK + c], rho = state[kRhoNow * kMaxK + c];
  const double al = state[kAlpha * kMaxK + c];
  const bool moves = gam != 0.0;
  const bool three = moves && !first && rho != 1.0;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * RP + rr; row < n; row += static_cast<int64_t>(gridDim.x) * RP) {
    const double rv = r[row * ldr + c];
    double sv = 0.0;
    if constexpr (PSI) sv = psi[row * ldpsi + c];
    double rn = rv, sn = sv;
    if (moves) {
      const double yv = y[row * ldy + c];
      rn = rv - gam * (rv - al * yv);
      if constexpr (PSI) sn = sv + gam * z[row * ldz + c];
      if (three) {
        rn = rho * rn + (1.0 - rho) * ro[row * ldro + c];
        if constexpr (PSI) sn = rho * sn + (1.0 - rho) * pso[row * ldpso + c];
      }
    }
    ro[row * ldro + c] = rn;
    if constexpr (PSI) pso[row * ldpso + c] = sn;
    nrm += rn * rn;
  }
  }
  if (normpart == nullptr) return;  // (uniform)
  red[threadIdx.x] = nrm;
  __syncthreads();
  if (rr == 0 && c < k) {
    double t = 0.0;
    for (int q = 0; q < RP; ++q) t += red[q * KP + c];
    normpart[static_cast<int64_t>(blockIdx.x) * k + c] = t;
  }
}

// one wave per column: the partial squared norms of the new residual -> out[c]
__global__ __launch_bounds__(64) void cg_norm_kernel(const double* __restrict__ partial, int nblocks, int k,
                                                    double* __restrict__ out) {
  const double s = wave_sum_partials(partial, nblocks, k, blockIdx.x);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
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

namespace eigd {
// coefficients of psi = sum_j s_j z_j from the (gam, rho) log (see the file header): one lane per column walks its steps
// backwards, s_last = alpha_last, s_j = alpha_j + beta_j s_{j+1} over the steps in which the column moved (gam != 0);
// S[j][c], zero for the other steps.  The host twin is adjoint._cg_solution_coefficients (tests).
__global__ __launch_bounds__(kMaxK) void cg_solution_coef_kernel(const double* __restrict__ log, int nsteps, int k,
                                                                double* __restrict__ S) {
  const int c = threadIdx.x;
  if (c >= k) return;
  double s = 0.0, alpha_next = 0.0, rho_next = 1.0;
  bool have = false;
  for (int j = nsteps - 1; j >= 0; --j) {
    const double gam = log[static_cast<int64_t>(2 * j) * kMaxK + c], rho = log[static_cast<int64_t>(2 * j + 1) * kMaxK + c];
    double out = 0.0;
    if (gam != 0.0) {
      const double alpha = rho * gam;
      s = have ? alpha + ((rho_next - 1.0) * alpha / alpha_next) * s : alpha;
      have = true;
      alpha_next = alpha;
      rho_next = rho;
      out = s;
    }
    S[static_cast<int64_t>(j) * k + c] = out;
  }
}
}  // namespace eigd

// the coefficient kernel over partial sums some other kernel left ([nblocks][2 k]: r.z then z.y per column)
int eigd::cg_coefficients_from_partials(eigd_ctx* ctx, const double* partial, int nblocks, int k, const double* dNorm2,
                                        double* dState, int step, int first, double* dLog) {
  hipLaunchKernelGGL(cg_coef_kernel, dim3(k), dim3(128), 0, ctx->stream, partial, nblocks, k, dNorm2, dState, step, first,
                     dLog);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

using namespace eigd;

extern "C" {

int eigd_cg_state_rows(void) { return kCgRows; }

int eigd_cg_coefficients(eigd_ctx* ctx, int n, int k, const double* dZ, int ldz, const double* dR, int ldr, const double* dY,
                         int ldy, const double* dNorm2, double* dState, int step, int first, double* dLog) {
  EIGD_REQUIRE(ctx && dZ && dR && dY && dState, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldz >= k && ldr >= k && ldy >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = cg_grid(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * static_cast<size_t>(nb) * 2 * k);
  if (rc) return rc;
  double* partial = ctx->scratch;
  rc = cg_dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(cg_dots_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dZ, ldz, dR, ldr,
                       dY, ldy, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  hipLaunchKernelGGL(cg_coef_kernel, dim3(k), dim3(128), 0, ctx->stream, partial, nb, k, dNorm2, dState, step, first,
                     dLog);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_cg_solution_coefficients(eigd_ctx* ctx, int k, const double* dLog, int nsteps, double* dS) {
  EIGD_REQUIRE(ctx && dLog && dS && k >= 1 && k <= kMaxK && nsteps >= 1, "bad argument");
  hipLaunchKernelGGL(cg_solution_coef_kernel, dim3(1), dim3(kMaxK), 0, ctx->stream, dLog, nsteps, k, dS);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_cg_update(eigd_ctx* ctx, int n, int k, const double* dR, int ldr, double* dRold, int ldro, const double* dPsi,
                   int ldpsi, double* dPsiOld, int ldpso, const double* dZ, int ldz, const double* dY, int ldy,
                   const double* dState, int first, double* dNorm2) {
  EIGD_REQUIRE(ctx && dR && dRold && dY && dState, "null argument");
  const bool with_psi = dPsi != nullptr;   // without: the residual recurrence alone (psi from the z history, see the header)
  EIGD_REQUIRE(!with_psi || (dPsiOld && dZ), "the solution recurrence needs psi, psi_old and z");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldr >= k && ldro >= k && ldy >= k &&
                   (!with_psi || (ldpsi >= k && ldpso >= k && ldz >= k)),
               "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = cg_grid(n, (kThreads / kp) * 4);
  double* partial = nullptr;
  int rc = EIGD_OK;
  if (dNorm2 != nullptr) {
    rc = ctx->ensure_scratch(sizeof(double) * static_cast<size_t>(nb) * k);
    if (rc) return rc;
    partial = ctx->scratch;
  }
  rc = cg_dispatch_kp(k, [&](auto KP) {
    if (with_psi)
      hipLaunchKernelGGL((cg_update_kernel<decltype(KP)::value, true>), dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dR, ldr,
                         dRold, ldro, dPsi, ldpsi, dPsiOld, ldpso, dZ, ldz, dY, ldy, dState, first, partial);
    else
      hipLaunchKernelGGL((cg_update_kernel<decltype(KP)::value, false>), dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dR, ldr,
                         dRold, ldro, dPsi, ldpsi, dPsiOld, ldpso, dZ, ldz, dY, ldy, dState, first, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  if (dNorm2 == nullptr) return EIGD_OK;
  hipLaunchKernelGGL(cg_norm_kernel, dim3(k), dim3(64), 0, ctx->stream, partial, nb, k, dNorm2);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

}  // extern "C"
