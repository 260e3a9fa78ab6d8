// Element-wise bilinear forms for the total-derivative callbacks:
//     out[e] += alpha * scale[e] * sum_c  w_e(:, c)^T  M_e  v_e(:, c)
// with w_e, v_e the rows of two n x k blocks gathered through the element's dof list.  This is
// d/d(rho_e) of w^T K v (or w^T G v, w^T M v) for a SIMP-penalised Q4 mesh, i.e. what the
// reference's harness callbacks get_stiffness_matrix_deriv / get_mass_matrix_deriv /
// get_stress_stiffness_matrix_xderiv compute with numpy einsums (examples/buckling.py:178-218,
// 283-340; examples/natural_frequency.py:162-203, 238-284; examples/thermal.py:150-190, 216-246),
// so W, Phi and psi never leave HBM during add_total_derivative.
//
// KP lanes share one element (one lane per mode column): row gathers are k*8-byte contiguous
// segments, the per-element sum is a fixed shuffle tree (deterministic).
#include "common.h"

namespace eigd {

__device__ const double g_zero_fem = 0.0;
__device__ const int32_t g_minus_one = -1;

template <int KP>
__global__ __launch_bounds__(kThreads) void elem_bilinear_kernel(int nelem, int nd, const int32_t* __restrict__ edofs,
                                                                const double* __restrict__ Me, int per_elem,
                                                                const double* __restrict__ scale,
                                                                const double* __restrict__ W, int ldw,
                                                                const double* __restrict__ V, int ldv, int k,
                                                                double alpha, double* __restrict__ out) {
  constexpr int EPB = kThreads / KP;
  const int c = threadIdx.x % KP;
  const int el = blockIdx.x * EPB + threadIdx.x / KP;
  const bool valid = (el < nelem) && (c < k);
  // all element dofs first, then all sixteen gathers: two memory latencies per element instead of sixteen
  // (masked lanes read a zero word through address select; a branch around a load serialises the loads)
  double w[8], v[8];
  int dofs[8];
  const int32_t* ed = edofs + static_cast<int64_t>(valid ? el : 0) * nd;
#pragma unroll
  for (int a = 0; a < 8; ++a) dofs[a] = *((valid && a < nd) ? ed + a : &g_minus_one);
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    w[a] = *((dofs[a] >= 0) ? W + static_cast<int64_t>(dofs[a]) * ldw + c : &g_zero_fem);
    v[a] = *((dofs[a] >= 0) ? V + static_cast<int64_t>(dofs[a]) * ldv + c : &g_zero_fem);
  }
  double s = 0.0;
  if (valid) {
    const double* M = Me + (per_elem ? static_cast<int64_t>(el) * nd * nd : 0);
    for (int a = 0; a < nd; ++a) {
      double t = 0.0;
      for (int b = 0; b < nd; ++b) t += M[a * nd + b] * v[b];
      s += w[a] * t;
    }
  }
#pragma unroll
  for (int off = KP / 2; off > 0; off >>= 1) s += __shfl_down(s, off, KP);
  if (c == 0 && el < nelem) out[el] += alpha * (scale ? scale[el] : 1.0) * s;
}

}  // namespace eigd

using namespace eigd;

extern "C" int eigd_elem_bilinear(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, const double* dMe,
                                  int per_elem, const double* dscale, const double* dW, int ldw, const double* dV,
                                  int ldv, int k, double alpha, double* dOut) {
  EIGD_REQUIRE(ctx && d_edofs && dMe && dW && dV && dOut, "null argument");
  EIGD_REQUIRE(nelem > 0 && nd >= 1 && nd <= 8 && k >= 1 && k <= kMaxK && ldw >= k && ldv >= k,
               "bad shape nelem=%d nd=%d k=%d", nelem, nd, k);
  const int kp = next_pow2(k);
  const int epb = kThreads / kp;
  const dim3 grid((nelem + epb - 1) / epb);
#define EIGD_EB_CASE(KP)                                                                                             \
  case KP:                                                                                                           \
    hipLaunchKernelGGL(elem_bilinear_kernel<KP>, grid, dim3(kThreads), 0, ctx->stream, nelem, nd, d_edofs, dMe,      \
                       per_elem, dscale, dW, ldw, dV, ldv, k, alpha, dOut);                                          \
    break;
  switch (kp) {
    EIGD_EB_CASE(1)
    EIGD_EB_CASE(2)
    EIGD_EB_CASE(4)
    EIGD_EB_CASE(8)
    EIGD_EB_CASE(16)
    EIGD_EB_CASE(32)
    EIGD_EB_CASE(64)
    default:
      set_error("internal: unexpected kp=%d", kp);
      return EIGD_E_INTERNAL;
  }
#undef EIGD_EB_CASE
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}
