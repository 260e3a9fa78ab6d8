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
#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <vector>

#include "common.h"

namespace eigd {

__device__ const double g_zero_fem = 0.0;
__device__ const int32_t g_minus_one = -1;

template <int KP, int ND>  // ND: dofs per element compiled in (8: the Q4 plane / thermal elements, 24: 6-dof shell facets)
__global__ __launch_bounds__(kThreads) void elem_bilinear_kernel(int nelem, int nd, const int32_t* __restrict__ edofs,
                                                                const double* __restrict__ Me, int per_elem,
                                                                const int32_t* __restrict__ etype,
                                                                const double* __restrict__ scale,
                                                                const double* __restrict__ W, int ldw,
                                                                const double* __restrict__ V, int ldv, int k,
                                                                double alpha, double* __restrict__ out) {
  constexpr int EPB = kThreads / KP;
  const int c = threadIdx.x % KP;
  const int el = blockIdx.x * EPB + threadIdx.x / KP;
  const bool valid = (el < nelem) && (c < k);
  // all element dofs first, then all the gathers: two memory latencies per element instead of one per dof
  // (masked lanes read a zero word through address select; a branch around a load serialises the loads)
  double w[ND], v[ND];
  int dofs[ND];
  const int32_t* ed = edofs + static_cast<int64_t>(valid ? el : 0) * nd;
#pragma unroll
  for (int a = 0; a < ND; ++a) dofs[a] = *((valid && a < nd) ? ed + a : &g_minus_one);
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    w[a] = *((dofs[a] >= 0) ? W + static_cast<int64_t>(dofs[a]) * ldw + c : &g_zero_fem);
    v[a] = *((dofs[a] >= 0) ? V + static_cast<int64_t>(dofs[a]) * ldv + c : &g_zero_fem);
  }
  double s = 0.0;
  // per_elem 0: one shared matrix; 1: a matrix per element; 2: a matrix per element TYPE (etype[e]).
  // The shared matrix sits at wave-uniform addresses (kernel argument + constant offset when nd == ND): scalar loads.
  // Through a lane-dependent pointer its nd^2 entries were nd^2 vector loads per lane next to the 2 nd gathers that do
  // the work -- the address path of the CU, not HBM, set the pace (600 us per call at 1 M dofs and 32 columns).  Same
  // products in the same order: the same bits.
  auto form = [&](const double* __restrict__ M, auto full) {
    constexpr bool kFull = decltype(full)::value;  // nd == ND: no bounds inside the loops
#pragma unroll
    for (int a = 0; a < ND; ++a) {
      if (kFull || a < nd) {
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < ND; ++b)
          if (kFull || b < nd) t += M[a * (kFull ? ND : nd) + b] * v[b];
        s += w[a] * t;
      }
    }
  };
  if (per_elem == 0) {
    if (nd == ND)
      form(Me, std::true_type{});
    else
      form(Me, std::false_type{});
    if (!valid) s = 0.0;
  } else if (valid) {
    const int64_t which = (per_elem == 1) ? el : etype[el];
    form(Me + which * nd * nd, std::false_type{});
  }
#pragma unroll
  for (int off = KP / 2; off > 0; off >>= 1) s += __shfl_down(s, off, KP);
  if (c == 0 && el < nelem) out[el] += alpha * (scale ? scale[el] : 1.0) * s;
}

// Assembly of element matrices into CSR values, gather form: the entry of every stored (row, column) pair sums its
// contributions in a fixed order (ascending element, then position in the element matrix) -- no atomics, the same
// bits on every run.  vals[z] = sum_s scale[e_s] * Me[(per_elem ? e_s : 0)][ab_s], src = e * nd^2 + ab.
__global__ __launch_bounds__(kThreads) void assemble_gather_kernel(int64_t nnz, const int32_t* __restrict__ nz_ptr,
                                                                  const int32_t* __restrict__ nz_src, int nd2,
                                                                  const double* __restrict__ Me, int per_elem,
                                                                  const int32_t* __restrict__ etype,
                                                                  const double* __restrict__ scale,
                                                                  double* __restrict__ vals) {
  for (int64_t z = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; z < nnz;
       z += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    double sum = 0.0;
    for (int q = nz_ptr[z]; q < nz_ptr[z + 1]; ++q) {
      const int src = nz_src[q];
      const int e = src / nd2, ab = src - e * nd2;
      const int64_t which = (per_elem == 1) ? e : (per_elem == 2 ? etype[e] : 0);
      const double m = Me[which * nd2 + ab];
      sum += (scale ? scale[e] : 1.0) * m;
    }
    vals[z] = sum;
  }
}

// Element matrices that depend linearly on the element's own dof values (geometric / stress stiffness of a
// linear pre-buckling state, examples/buckling.py:220-255): Me[e] = sum_m (L[m] . u_e) * Q[m], u_e gathered from the
// FULL dof vector through edofs (no constraints here: prescribed dofs carry their values).  One thread per entry.
__global__ __launch_bounds__(kThreads) void elem_linear_matrices_kernel(int nelem, int nd, const int32_t* __restrict__ edofs,
                                                                       const double* __restrict__ u, int nterms,
                                                                       const double* __restrict__ L,
                                                                       const double* __restrict__ Q,
                                                                       double* __restrict__ out) {
  const int nd2 = nd * nd;
  const int64_t total = static_cast<int64_t>(nelem) * nd2;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int e = static_cast<int>(idx / nd2), ab = static_cast<int>(idx - static_cast<int64_t>(e) * nd2);
    double ue[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) ue[a] = (a < nd) ? u[edofs[static_cast<int64_t>(e) * nd + a]] : 0.0;
    double sum = 0.0;
    for (int m = 0; m < nterms; ++m) {
      double c = 0.0;
      for (int a = 0; a < nd; ++a) c += L[m * nd + a] * ue[a];
      sum += c * Q[m * nd2 + ab];
    }
    out[idx] = sum;
  }
}


// Transpose of elem_linear_matrices with respect to u: for Me[e](u) = sum_m (L[m] . u_e) Q[m],
//     g[e*nd + a] = alpha * scale[e] * sum_m L[m][a] * t_m,   t_m = sum_c w_e(:, c)^T Q[m] v_e(:, c),
// i.e. d/du_e of sum_c w_c^T G(u) v_c per element -- the tensor form of the reference's
// get_stress_stiffness_matrix_uderiv (examples/buckling.py:283-316: dfds, then C Be contraction).  The element
// vectors are summed into the dof vector by a CSR product with the (fixed-order) dof <- element-entry incidence
// matrix, so the path adjoint right-hand side is reproducible bit for bit.  KP lanes (one per column) share an element.
constexpr int kMaxTerms = 16;

template <int KP>
__global__ __launch_bounds__(kThreads) void elem_linear_adjoint_kernel(int nelem, int nd, const int32_t* __restrict__ edofs,
                                                                      int nterms, const double* __restrict__ L,
                                                                      const double* __restrict__ Q,
                                                                      const double* __restrict__ scale,
                                                                      const double* __restrict__ W, int ldw,
                                                                      const double* __restrict__ V, int ldv, int k,
                                                                      double alpha, double* __restrict__ out) {
  constexpr int EPB = kThreads / KP;
  const int c = threadIdx.x % KP;
  const int el = blockIdx.x * EPB + threadIdx.x / KP;
  const bool valid = (el < nelem) && (c < k);
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
  const int nd2 = nd * nd;
  double g[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) g[a] = 0.0;
  for (int m = 0; m < nterms; ++m) {
    const double* Qm = Q + static_cast<int64_t>(m) * nd2;
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      if (a < nd) {
        double r = 0.0;
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (b < nd) r += Qm[a * nd + b] * v[b];
        t += w[a] * r;
      }
    }
    // fixed butterfly: every lane of the element ends with the same sum over the columns
#pragma unroll
    for (int off = KP / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, KP);
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < nd) g[a] += L[m * nd + a] * t;
  }
  if (c == 0 && el < nelem) {
    const double f = alpha * (scale ? scale[el] : 1.0);
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < nd) out[static_cast<int64_t>(el) * nd + a] = f * g[a];
  }
}


// Elementwise maps of the design-variable chain, one thread per entry:
//   kind 0: out = x^p + c0                       SIMP penalisation rhoE^p + rho0 (examples/buckling.py:157-160)
//   kind 1: out = p * x^(p-1)  [* g]             its derivative (buckling.py:207-208), optionally times g
//   kind 2: out = (tanh(b e) + tanh(b (x - e))) / (tanh(b e) + tanh(b (1 - e)))      projection (node_filter.py:175-181)
//   kind 3: out = g * (b / denom) / cosh(b (x - e))^2                                its derivative (node_filter.py:196-203)
//   kind 4: out = c0 * x + c1                    linear interpolation (thermal.py:133, 198; mass densities)
// p, c0 double as (b, e) for the projection.
__global__ __launch_bounds__(kThreads) void design_map_kernel(int64_t n, int kind, double p, double c0, double c1,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ g, double* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double xi = x[i];
    double r;
    if (kind == 0) {
      r = pow(xi, p) + c0;
    } else if (kind == 1) {
      r = p * pow(xi, p - 1.0);
      if (g) r *= g[i];
    } else if (kind == 2) {
      r = (tanh(p * c0) + tanh(p * (xi - c0))) / (tanh(p * c0) + tanh(p * (1.0 - c0)));
    } else if (kind == 3) {
      const double ch = cosh(p * (xi - c0));
      r = (g ? g[i] : 1.0) * ((p / (tanh(p * c0) + tanh(p * (1.0 - c0)))) * 1.0 / (ch * ch));
    } else {
      r = c0 * xi + c1;
    }
    out[i] = r;
  }
}

}  // namespace eigd

using namespace eigd;

extern "C" int eigd_elem_linear_matrices(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, const double* du,
                                         int nterms, const double* dL, const double* dQ, double* dOut) {
  EIGD_REQUIRE(ctx && d_edofs && du && dL && dQ && dOut, "null argument");
  EIGD_REQUIRE(nelem > 0 && nd >= 1 && nd <= 8 && nterms >= 1, "bad shape nelem=%d nd=%d nterms=%d", nelem, nd, nterms);
  const int64_t total = static_cast<int64_t>(nelem) * nd * nd;
  const int nb = static_cast<int>(std::min<int64_t>((total + kThreads - 1) / kThreads, 65536));
  hipLaunchKernelGGL(elem_linear_matrices_kernel, dim3(nb), dim3(kThreads), 0, ctx->stream, nelem, nd, d_edofs, du, nterms,
                     dL, dQ, dOut);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

struct eigd_assembler {
  eigd_ctx* ctx = nullptr;
  int n = 0, nelem = 0, nd = 0;
  int64_t nnz = 0;
  std::vector<int32_t> indptr, indices;  // host copy of the pattern (rows sorted)
  int32_t *d_nz_ptr = nullptr, *d_nz_src = nullptr;
};

extern "C" int eigd_assembler_free(eigd_assembler* a) {
  if (!a) return EIGD_OK;
  if (a->ctx && a->ctx->stream) (void)hipStreamSynchronize(a->ctx->stream);
  if (a->d_nz_ptr) (void)hipFree(a->d_nz_ptr);
  if (a->d_nz_src) (void)hipFree(a->d_nz_src);
  delete a;
  return EIGD_OK;
}

extern "C" int eigd_assembler_create(eigd_ctx* ctx, int n, int nelem, int nd, const int32_t* elem_dofs,
                                     eigd_assembler** out) {
  EIGD_REQUIRE(ctx && elem_dofs && out, "null argument");
  EIGD_REQUIRE(n > 0 && nelem > 0 && nd >= 1 && nd <= 24, "bad shape n=%d nelem=%d nd=%d", n, nelem, nd);
  *out = nullptr;
  const int nd2 = nd * nd;
  EIGD_REQUIRE(static_cast<int64_t>(nelem) * nd2 < (int64_t(1) << 31), "too many element entries");
  // (row, col, source) triples of the unconstrained element entries, sorted by (row, col, source)
  struct Trip { int32_t r, c, src; };
  std::vector<Trip> tr;
  tr.reserve(static_cast<size_t>(nelem) * nd2);
  for (int e = 0; e < nelem; ++e)
    for (int a = 0; a < nd; ++a) {
      const int32_t r = elem_dofs[static_cast<int64_t>(e) * nd + a];
      if (r < 0) continue;
      EIGD_REQUIRE(r < n, "element %d: dof %d out of range", e, r);
      for (int b = 0; b < nd; ++b) {
        const int32_t c = elem_dofs[static_cast<int64_t>(e) * nd + b];
        if (c < 0) continue;
        tr.push_back(Trip{r, c, static_cast<int32_t>(e * nd2 + a * nd + b)});
      }
    }
  std::sort(tr.begin(), tr.end(), [](const Trip& x, const Trip& y) {
    if (x.r != y.r) return x.r < y.r;
    if (x.c != y.c) return x.c < y.c;
    return x.src < y.src;
  });
  eigd_assembler* a = new eigd_assembler();
  a->ctx = ctx;
  a->n = n;
  a->nelem = nelem;
  a->nd = nd;
  a->indptr.assign(static_cast<size_t>(n) + 1, 0);
  std::vector<int32_t> nz_ptr, nz_src(tr.size());
  nz_ptr.reserve(tr.size() / 2 + 2);
  for (size_t q = 0; q < tr.size(); ++q) {
    if (q == 0 || tr[q].r != tr[q - 1].r || tr[q].c != tr[q - 1].c) {
      nz_ptr.push_back(static_cast<int32_t>(q));
      a->indices.push_back(tr[q].c);
      a->indptr[tr[q].r + 1] += 1;
    }
    nz_src[q] = tr[q].src;
  }
  nz_ptr.push_back(static_cast<int32_t>(tr.size()));
  for (int i = 0; i < n; ++i) a->indptr[i + 1] += a->indptr[i];
  a->nnz = static_cast<int64_t>(a->indices.size());
  EIGD_HIP(hipSetDevice(ctx->device));
  hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&a->d_nz_ptr), sizeof(int32_t) * nz_ptr.size());
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&a->d_nz_src), sizeof(int32_t) * std::max<size_t>(nz_src.size(), 1));
  if (e1 != hipSuccess || e2 != hipSuccess) {
    eigd_assembler_free(a);
    set_error("hipMalloc failed for the assembly lists");
    return EIGD_E_HIP;
  }
  EIGD_HIP(hipMemcpy(a->d_nz_ptr, nz_ptr.data(), sizeof(int32_t) * nz_ptr.size(), hipMemcpyHostToDevice));
  if (!nz_src.empty())
    EIGD_HIP(hipMemcpy(a->d_nz_src, nz_src.data(), sizeof(int32_t) * nz_src.size(), hipMemcpyHostToDevice));
  *out = a;
  return EIGD_OK;
}

extern "C" int eigd_assembler_nnz(eigd_assembler* a, int64_t* nnz) {
  EIGD_REQUIRE(a && nnz, "null argument");
  *nnz = a->nnz;
  return EIGD_OK;
}

extern "C" int eigd_assembler_pattern(eigd_assembler* a, int32_t* hindptr, int32_t* hindices) {
  EIGD_REQUIRE(a && hindptr && hindices, "null argument");
  std::copy(a->indptr.begin(), a->indptr.end(), hindptr);
  std::copy(a->indices.begin(), a->indices.end(), hindices);
  return EIGD_OK;
}

extern "C" int eigd_assemble(eigd_assembler* a, const double* dMe, int per_elem, const int32_t* d_etype,
                             const double* dscale, double* dvals) {
  EIGD_REQUIRE(a && dMe && dvals, "null argument");
  EIGD_REQUIRE(per_elem >= 0 && per_elem <= 2 && (per_elem != 2 || d_etype), "per_elem = 2 needs the element types");
  if (a->nnz == 0) return EIGD_OK;
  const int nb = static_cast<int>(std::min<int64_t>((a->nnz + kThreads - 1) / kThreads, 65536));
  hipLaunchKernelGGL(assemble_gather_kernel, dim3(nb), dim3(kThreads), 0, a->ctx->stream, a->nnz, a->d_nz_ptr, a->d_nz_src,
                     a->nd * a->nd, dMe, per_elem, d_etype, dscale, dvals);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

extern "C" int eigd_elem_bilinear(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, const double* dMe,
                                  int per_elem, const int32_t* d_etype, const double* dscale, const double* dW, int ldw,
                                  const double* dV, int ldv, int k, double alpha, double* dOut) {
  EIGD_REQUIRE(ctx && d_edofs && dMe && dW && dV && dOut, "null argument");
  EIGD_REQUIRE(nelem > 0 && nd >= 1 && nd <= 24 && k >= 1 && k <= kMaxK && ldw >= k && ldv >= k,
               "bad shape nelem=%d nd=%d k=%d", nelem, nd, k);
  EIGD_REQUIRE(per_elem >= 0 && per_elem <= 2 && (per_elem != 2 || d_etype), "per_elem = 2 needs the element types");
  const int kp = next_pow2(k);
  const int epb = kThreads / kp;
  const dim3 grid((nelem + epb - 1) / epb);
#define EIGD_EB_CASE(KP)                                                                                             \
  case KP:                                                                                                           \
    if (nd <= 8)                                                                                                     \
      hipLaunchKernelGGL((elem_bilinear_kernel<KP, 8>), grid, dim3(kThreads), 0, ctx->stream, nelem, nd, d_edofs, dMe, \
                         per_elem, d_etype, dscale, dW, ldw, dV, ldv, k, alpha, dOut);                               \
    else                                                                                                             \
      hipLaunchKernelGGL((elem_bilinear_kernel<KP, 24>), grid, dim3(kThreads), 0, ctx->stream, nelem, nd, d_edofs, dMe, \
                         per_elem, d_etype, dscale, dW, ldw, dV, ldv, k, alpha, dOut);                               \
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

extern "C" int eigd_elem_linear_adjoint(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, int nterms,
                                        const double* dL, const double* dQ, const double* dscale, const double* dW,
                                        int ldw, const double* dV, int ldv, int k, double alpha, double* dOut) {
  EIGD_REQUIRE(ctx && d_edofs && dL && dQ && dW && dV && dOut, "null argument");
  EIGD_REQUIRE(nelem > 0 && nd >= 1 && nd <= 8 && nterms >= 1 && nterms <= kMaxTerms && k >= 1 && k <= kMaxK &&
                   ldw >= k && ldv >= k,
               "bad shape nelem=%d nd=%d nterms=%d k=%d", nelem, nd, nterms, k);
  const int kp = next_pow2(k);
  const int epb = kThreads / kp;
  const dim3 grid((nelem + epb - 1) / epb);
#define EIGD_ELA_CASE(KP)                                                                                            \
  case KP:                                                                                                           \
    hipLaunchKernelGGL(elem_linear_adjoint_kernel<KP>, grid, dim3(kThreads), 0, ctx->stream, nelem, nd, d_edofs,     \
                       nterms, dL, dQ, dscale, dW, ldw, dV, ldv, k, alpha, dOut);                                    \
    break;
  switch (kp) {
    EIGD_ELA_CASE(1)
    EIGD_ELA_CASE(2)
    EIGD_ELA_CASE(4)
    EIGD_ELA_CASE(8)
    EIGD_ELA_CASE(16)
    EIGD_ELA_CASE(32)
    EIGD_ELA_CASE(64)
    default:
      set_error("internal: unexpected kp=%d", kp);
      return EIGD_E_INTERNAL;
  }
#undef EIGD_ELA_CASE
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

extern "C" int eigd_design_map(eigd_ctx* ctx, int64_t n, int kind, double p, double c0, double c1, const double* dx,
                               const double* dg, double* dout) {
  EIGD_REQUIRE(ctx && dx && dout, "null argument");
  EIGD_REQUIRE(n > 0 && kind >= 0 && kind <= 4, "bad arguments n=%lld kind=%d", (long long)n, kind);
  const int nb = static_cast<int>(std::min<int64_t>((n + kThreads - 1) / kThreads, 65536));
  hipLaunchKernelGGL(design_map_kernel, dim3(nb), dim3(kThreads), 0, ctx->stream, n, kind, p, c0, c1, dx, dg, dout);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}
