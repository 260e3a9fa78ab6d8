// Sparse shift-invert factorisation and multi-RHS triangular sweeps for gfx950.
//
// Replaces SuperLU behind eigd's SpLuOperator (eigd/eigenvector_derivatives.py:11-23).
// The shifted matrix K - sigma M (or K + sigma G below the first buckling load) is
// symmetric positive definite in every example of the reference, so the factor is
// LL^T on a nested-dissection ordering:
//
//   numeric : multifrontal.  Every front is a dense d x d square (d = own columns +
//             border) in one big HBM buffer; levels of the assembly tree are processed
//             bottom-up, all fronts of a level batched in the same launches.  Per panel
//             step of W columns: potrf (+ explicit inverse of the W x W diagonal block),
//             trsm as a product with that inverse, trailing update tile by tile.
//   solve   : the same (level, step) schedule.  Each front owns a d x k slice of a vector
//             workspace; forward substitution passes border contributions child -> parent
//             (extend-add of vectors), backward substitution pulls them parent -> child.
//             No atomics anywhere: the sweeps are bitwise reproducible.
//
// Per solve the kernels stream nnz(L) doubles twice (forward + backward) plus
// O(sum of front dimensions) x k vector traffic: HBM bound for k <= 32.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "symbolic.h"

struct eigd_symbolic {
  eigd::Symbolic s;
};

namespace eigd {

constexpr int TW = 64;       // tile edge == max panel width == row chunk
constexpr int TLD = TW + 1;  // padded LDS leading dimension
constexpr int KBMAX = 32;    // right-hand sides per sweep

struct FrontArrays {
  const int* c0;
  const int* ns;
  const int* bs;
  const int* parent;
  const int64_t* foff;
  const int64_t* voff;
  const int64_t* ioff;
  const int64_t* bptr;
  const int* rel;
  int W;
};

__device__ __forceinline__ int find_slot(const int* __restrict__ pref, int na, int idx) {
  int lo = 0, hi = na;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pref[mid] <= idx)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------ assembly
__global__ void scatter_a_kernel(int64_t nlower, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                 const double* __restrict__ data, double* __restrict__ F) {
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < nlower;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x)
    F[dst[e]] = data[src[e]];
}

// update matrix of each child (its trailing bs x bs block) added into the parent front
__global__ __launch_bounds__(kThreads) void extend_add_kernel(FrontArrays fa, const int* __restrict__ children,
                                                             double* __restrict__ F) {
  const int c = children[blockIdx.x];
  const int p = fa.parent[c];
  const int ns = fa.ns[c], bs = fa.bs[c];
  const int64_t dc = ns + bs, dp = fa.ns[p] + fa.bs[p];
  const int* __restrict__ rel = fa.rel + fa.bptr[c];
  const double* __restrict__ Fc = F + fa.foff[c] + static_cast<int64_t>(ns) * dc + ns;
  double* Fp = F + fa.foff[p];
  const int total = bs * bs;
  for (int idx = blockIdx.y * kThreads + threadIdx.x; idx < total; idx += gridDim.y * kThreads) {
    const int j = idx / bs, i = idx - j * bs;
    if (i >= j) Fp[static_cast<int64_t>(rel[j]) * dp + rel[i]] += Fc[static_cast<int64_t>(j) * dc + i];
  }
}

// ------------------------------------------------------------------ panel factorisation
// one workgroup per active front: Cholesky of the W x W diagonal block + its inverse
__global__ __launch_bounds__(kThreads) void potrf_inv_kernel(FrontArrays fa, const int* __restrict__ fronts, int step,
                                                            double* __restrict__ F, double* __restrict__ Inv,
                                                            int* __restrict__ flag) {
  __shared__ double S[TW * TLD];
  __shared__ double Iv[TW * TLD];
  const int f = fronts[blockIdx.x];
  const int W = fa.W;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = step * W;
  const int w = min(W, ns - j0);
  double* Fd = F + fa.foff[f] + static_cast<int64_t>(j0) * d + j0;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < w * w; idx += kThreads) {
    const int j = idx / w, i = idx - j * w;
    S[j * TLD + i] = (i >= j) ? Fd[static_cast<int64_t>(j) * d + i] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < w; ++j) {
    if (tid == 0) {
      double p = S[j * TLD + j];
      if (!(p > 0.0) || !(p < 1.0e300)) {
        atomicCAS(flag, 0, f + 1);
        p = 1.0;
      }
      S[j * TLD + j] = sqrt(p);
    }
    __syncthreads();
    const double dj = S[j * TLD + j];
    for (int i = j + 1 + tid; i < w; i += kThreads) S[j * TLD + i] /= dj;
    __syncthreads();
    const int m = w - j - 1;
    for (int idx = tid; idx < m * m; idx += kThreads) {
      const int cc = idx / m, ii = idx - cc * m;
      if (ii >= cc) S[(j + 1 + cc) * TLD + j + 1 + ii] -= S[j * TLD + j + 1 + ii] * S[j * TLD + j + 1 + cc];
    }
    __syncthreads();
  }
  // inverse of the lower triangular block, one column per lane
  if (tid < w) {
    const int c = tid;
    for (int i = 0; i < c; ++i) Iv[c * TLD + i] = 0.0;
    for (int i = c; i < w; ++i) {
      double sum = (i == c) ? 1.0 : 0.0;
      for (int j = c; j < i; ++j) sum -= S[j * TLD + i] * Iv[c * TLD + j];
      Iv[c * TLD + i] = sum / S[i * TLD + i];
    }
  }
  __syncthreads();
  double* Ig = Inv + fa.ioff[f] + static_cast<int64_t>(step) * W * W;
  for (int idx = tid; idx < w * w; idx += kThreads) {
    const int j = idx / w, i = idx - j * w;
    if (i >= j) Fd[static_cast<int64_t>(j) * d + i] = S[j * TLD + i];
    Ig[j * W + i] = Iv[j * TLD + i];
  }
}

// rows below the diagonal block: L21 = A21 * inv(L11)^T, one 64-row chunk per workgroup
__global__ __launch_bounds__(kThreads) void trsm_kernel(FrontArrays fa, const int* __restrict__ fronts, int na, int step,
                                                       const int* __restrict__ pref_chunks, double* __restrict__ F,
                                                       const double* __restrict__ Inv) {
  __shared__ double As[TW * TLD];
  __shared__ double Is[TW * TLD];
  const int q = find_slot(pref_chunks, na, blockIdx.x);
  const int f = fronts[q];
  const int chunk = blockIdx.x - pref_chunks[q];
  const int W = fa.W;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = step * W;
  const int w = min(W, ns - j0);
  const int j1 = j0 + w;
  const int row0 = j1 + chunk * TW;
  const int rows = min(TW, static_cast<int>(d) - row0);
  double* Fp = F + fa.foff[f] + static_cast<int64_t>(j0) * d + row0;  // element (r, j) at j*d + r
  const double* Ig = Inv + fa.ioff[f] + static_cast<int64_t>(step) * W * W;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < TW * TW; idx += kThreads) {
    const int j = idx / TW, r = idx - j * TW;
    As[j * TLD + r] = (j < w && r < rows) ? Fp[static_cast<int64_t>(j) * d + r] : 0.0;
    Is[j * TLD + r] = (j < w && r < w) ? Ig[j * W + r] : 0.0;  // Is[j][c] = inv(c, j)
  }
  __syncthreads();
  const int r0 = (tid % 16) * 4, c0 = (tid / 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.0;
  for (int j = 0; j < w; ++j) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = As[j * TLD + r0 + i];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = Is[j * TLD + c0 + c];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] += a[i] * b[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (r0 + i < rows && c0 + c < w) Fp[static_cast<int64_t>(c0 + c) * d + r0 + i] = acc[i][c];
}

// trailing update C(ti, tj) -= L21(ti) * L21(tj)^T on 64 x 64 tiles of the lower triangle
__global__ __launch_bounds__(kThreads) void syrk_kernel(FrontArrays fa, const int* __restrict__ fronts, int na, int step,
                                                       const int* __restrict__ pref_chunks,
                                                       const int* __restrict__ pref_tiles, double* __restrict__ F) {
  const int q = find_slot(pref_tiles, na, blockIdx.x);
  const int nch = pref_chunks[q + 1] - pref_chunks[q];
  const int local = blockIdx.x - pref_tiles[q];
  const int ti = local / nch, tj = local - ti * nch;
  if (tj > ti) return;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * TLD];
  const int f = fronts[q];
  const int W = fa.W;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = step * W;
  const int w = min(W, ns - j0);
  const int j1 = j0 + w;
  const int Ri = j1 + ti * TW, Rj = j1 + tj * TW;
  const int rows_i = min(TW, static_cast<int>(d) - Ri), rows_j = min(TW, static_cast<int>(d) - Rj);
  double* Fb = F + fa.foff[f];
  const double* Lp = Fb + static_cast<int64_t>(j0) * d;  // panel columns
  const int tid = threadIdx.x;
  for (int idx = tid; idx < TW * TW; idx += kThreads) {
    const int k = idx / TW, r = idx - k * TW;
    As[k * TLD + r] = (k < w && r < rows_i) ? Lp[static_cast<int64_t>(k) * d + Ri + r] : 0.0;
    Bs[k * TLD + r] = (k < w && r < rows_j) ? Lp[static_cast<int64_t>(k) * d + Rj + r] : 0.0;
  }
  __syncthreads();
  const int i0 = (tid % 16) * 4, jj0 = (tid / 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int k = 0; k < w; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = As[k * TLD + i0 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = Bs[k * TLD + jj0 + j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gi = Ri + i0 + i, gj = Rj + jj0 + j;
      if (i0 + i < rows_i && jj0 + j < rows_j && gi >= gj) Fb[static_cast<int64_t>(gj) * d + gi] -= acc[i][j];
    }
}

// ------------------------------------------------------------------ triangular sweeps
__global__ void solve_init_kernel(int64_t sumd, int kb, const int* __restrict__ v_src, const double* __restrict__ X,
                                  int ldx, double alpha, double* __restrict__ V) {
  const int64_t total = sumd * kb;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t q = idx / kb;
    const int c = static_cast<int>(idx - q * kb);
    const int src = v_src[q];
    V[idx] = (src >= 0) ? alpha * X[static_cast<int64_t>(src) * ldx + c] : 0.0;
  }
}

__global__ void solve_out_kernel(int64_t sumd, int kb, const int* __restrict__ v_src, const double* __restrict__ V,
                                 double* __restrict__ X, int ldx) {
  const int64_t total = sumd * kb;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t q = idx / kb;
    const int c = static_cast<int>(idx - q * kb);
    const int src = v_src[q];
    if (src >= 0) X[static_cast<int64_t>(src) * ldx + c] = V[idx];
  }
}

// forward: border part of each child's vector added into its parent's vector
__global__ __launch_bounds__(kThreads) void vec_extend_add_kernel(FrontArrays fa, const int* __restrict__ children,
                                                                 int kb, double* __restrict__ V) {
  const int c = children[blockIdx.x];
  const int p = fa.parent[c];
  const int ns = fa.ns[c], bs = fa.bs[c];
  const int* __restrict__ rel = fa.rel + fa.bptr[c];
  const double* __restrict__ Vc = V + (fa.voff[c] + ns) * kb;
  double* Vp = V + fa.voff[p] * kb;
  const int total = bs * kb;
  for (int idx = blockIdx.y * kThreads + threadIdx.x; idx < total; idx += gridDim.y * kThreads) {
    const int i = idx / kb, col = idx - i * kb;
    Vp[static_cast<int64_t>(rel[i]) * kb + col] += Vc[idx];
  }
}

// backward: border part of each front's vector pulled from its parent's (already final) vector
__global__ __launch_bounds__(kThreads) void vec_gather_kernel(FrontArrays fa, const int* __restrict__ fronts, int kb,
                                                             double* __restrict__ V) {
  const int f = fronts[blockIdx.x];
  const int p = fa.parent[f];
  if (p < 0) return;
  const int ns = fa.ns[f], bs = fa.bs[f];
  const int* __restrict__ rel = fa.rel + fa.bptr[f];
  double* Vf = V + (fa.voff[f] + ns) * kb;
  const double* Vp = V + fa.voff[p] * kb;
  const int total = bs * kb;
  for (int idx = blockIdx.y * kThreads + threadIdx.x; idx < total; idx += gridDim.y * kThreads) {
    const int i = idx / kb, col = idx - i * kb;
    Vf[idx] = Vp[static_cast<int64_t>(rel[i]) * kb + col];
  }
}

// 64 x (4*KPT) output tile, thread (o = tid/4, cg = tid%4) owns KPT columns.
//   TRANS == false : acc[o][c] += sum_k As[k*TLD + o] * Bs[k*KB + c]
//   TRANS == true  : acc[o][c] += sum_k As[o*TLD + k] * Bs[k*KB + c]
template <int KPT, bool TRANS>
__device__ __forceinline__ void tile_mac(const double* __restrict__ As, const double* __restrict__ Bs, int kdim, int o,
                                         int cg, double (&acc)[KPT]) {
  constexpr int KB = 4 * KPT;
  for (int k = 0; k < kdim; ++k) {
    const double a = TRANS ? As[o * TLD + k] : As[k * TLD + o];
    const double* b = Bs + k * KB + cg * KPT;
#pragma unroll
    for (int q = 0; q < KPT; ++q) acc[q] += a * b[q];
  }
}

struct StepArgs {
  const int* fronts;       // active fronts of this (level, step), npanels descending
  const int* pref_chunks;  // na + 1
  int na;
  int step;
  int kb;
};

template <int KPT>
__device__ __forceinline__ void load_panel_chunk(double* As, const double* __restrict__ Lp, int64_t d, int w, int row0,
                                                 int rows) {
  // As[j*TLD + r] = L(row0 + r, j0 + j)
  for (int idx = threadIdx.x; idx < TW * TW; idx += kThreads) {
    const int j = idx / TW, r = idx - j * TW;
    As[j * TLD + r] = (j < w && r < rows) ? Lp[static_cast<int64_t>(j) * d + row0 + r] : 0.0;
  }
}

template <int KPT>
__device__ __forceinline__ void load_vec_rows(double* Bs, const double* __restrict__ Vrows, int kb, int rows) {
  constexpr int KB = 4 * KPT;
  for (int idx = threadIdx.x; idx < TW * KB; idx += kThreads) {
    const int r = idx / KB, c = idx - r * KB;
    Bs[idx] = (r < rows && c < kb) ? Vrows[static_cast<int64_t>(r) * kb + c] : 0.0;
  }
}

// forward, diagonal block: y1 = inv(L11) v1 ; fronts whose rows below fit one chunk are finished here
template <int KPT>
__global__ __launch_bounds__(kThreads) void fwd_diag_kernel(FrontArrays fa, StepArgs sa, const double* __restrict__ F,
                                                           const double* __restrict__ Inv, double* __restrict__ V) {
  constexpr int KB = 4 * KPT;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * KB];
  const int q = blockIdx.x;
  const int f = sa.fronts[q];
  const int nch = sa.pref_chunks[q + 1] - sa.pref_chunks[q];
  const int W = fa.W, kb = sa.kb;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = sa.step * W;
  const int w = min(W, ns - j0);
  const int j1 = j0 + w;
  const double* Ig = Inv + fa.ioff[f] + static_cast<int64_t>(sa.step) * W * W;
  double* Vf = V + fa.voff[f] * kb;
  const int tid = threadIdx.x, o = tid / 4, cg = tid % 4;
  for (int idx = tid; idx < TW * TW; idx += kThreads) {
    const int j = idx / TW, i = idx - j * TW;
    As[j * TLD + i] = (j < w && i < w) ? Ig[j * W + i] : 0.0;  // As[j][i] = inv(i, j)
  }
  load_vec_rows<KPT>(Bs, Vf + static_cast<int64_t>(j0) * kb, kb, w);
  __syncthreads();
  double acc[KPT];
#pragma unroll
  for (int t = 0; t < KPT; ++t) acc[t] = 0.0;
  tile_mac<KPT, false>(As, Bs, w, o, cg, acc);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < KPT; ++t) {
    const int c = cg * KPT + t;
    Bs[o * KB + c] = (o < w) ? acc[t] : 0.0;
    if (o < w && c < kb) Vf[static_cast<int64_t>(j0 + o) * kb + c] = acc[t];
  }
  if (nch != 1) return;
  const int rows = static_cast<int>(d) - j1;
  __syncthreads();
  load_panel_chunk<KPT>(As, F + fa.foff[f] + static_cast<int64_t>(j0) * d, d, w, j1, rows);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < KPT; ++t) acc[t] = 0.0;
  tile_mac<KPT, false>(As, Bs, w, o, cg, acc);
  if (o < rows) {
#pragma unroll
    for (int t = 0; t < KPT; ++t) {
      const int c = cg * KPT + t;
      if (c < kb) Vf[static_cast<int64_t>(j1 + o) * kb + c] -= acc[t];
    }
  }
}

// forward, rows below: v2 -= L21 y1, one 64-row chunk per workgroup (fronts with > 1 chunk)
template <int KPT>
__global__ __launch_bounds__(kThreads) void fwd_update_kernel(FrontArrays fa, StepArgs sa, const double* __restrict__ F,
                                                             double* __restrict__ V) {
  constexpr int KB = 4 * KPT;
  const int q = find_slot(sa.pref_chunks, sa.na, blockIdx.x);
  const int nch = sa.pref_chunks[q + 1] - sa.pref_chunks[q];
  if (nch <= 1) return;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * KB];
  const int f = sa.fronts[q];
  const int chunk = blockIdx.x - sa.pref_chunks[q];
  const int W = fa.W, kb = sa.kb;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = sa.step * W;
  const int w = min(W, ns - j0);
  const int row0 = j0 + w + chunk * TW;
  const int rows = min(TW, static_cast<int>(d) - row0);
  double* Vf = V + fa.voff[f] * kb;
  const int tid = threadIdx.x, o = tid / 4, cg = tid % 4;
  load_panel_chunk<KPT>(As, F + fa.foff[f] + static_cast<int64_t>(j0) * d, d, w, row0, rows);
  load_vec_rows<KPT>(Bs, Vf + static_cast<int64_t>(j0) * kb, kb, w);
  __syncthreads();
  double acc[KPT];
#pragma unroll
  for (int t = 0; t < KPT; ++t) acc[t] = 0.0;
  tile_mac<KPT, false>(As, Bs, w, o, cg, acc);
  if (o < rows) {
#pragma unroll
    for (int t = 0; t < KPT; ++t) {
      const int c = cg * KPT + t;
      if (c < kb) Vf[static_cast<int64_t>(row0 + o) * kb + c] -= acc[t];
    }
  }
}

// backward, rows below: partial t = L21(chunk)^T x2(chunk) into the partial slab (fronts with > 1 chunk)
template <int KPT>
__global__ __launch_bounds__(kThreads) void bwd_partial_kernel(FrontArrays fa, StepArgs sa, const double* __restrict__ F,
                                                              const double* __restrict__ V, double* __restrict__ P) {
  constexpr int KB = 4 * KPT;
  const int q = find_slot(sa.pref_chunks, sa.na, blockIdx.x);
  const int nch = sa.pref_chunks[q + 1] - sa.pref_chunks[q];
  if (nch <= 1) return;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * KB];
  const int f = sa.fronts[q];
  const int chunk = blockIdx.x - sa.pref_chunks[q];
  const int W = fa.W, kb = sa.kb;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = sa.step * W;
  const int w = min(W, ns - j0);
  const int row0 = j0 + w + chunk * TW;
  const int rows = min(TW, static_cast<int>(d) - row0);
  const double* Vf = V + fa.voff[f] * kb;
  const int tid = threadIdx.x, o = tid / 4, cg = tid % 4;
  load_panel_chunk<KPT>(As, F + fa.foff[f] + static_cast<int64_t>(j0) * d, d, w, row0, rows);
  load_vec_rows<KPT>(Bs, Vf + static_cast<int64_t>(row0) * kb, kb, rows);
  __syncthreads();
  double acc[KPT];
#pragma unroll
  for (int t = 0; t < KPT; ++t) acc[t] = 0.0;
  tile_mac<KPT, true>(As, Bs, rows, o, cg, acc);
  double* Pp = P + static_cast<int64_t>(blockIdx.x) * (TW * KBMAX);
  if (o < w) {
#pragma unroll
    for (int t = 0; t < KPT; ++t) Pp[o * KBMAX + cg * KPT + t] = acc[t];
  }
}

// backward, diagonal block: x1 = inv(L11)^T (y1 - sum of partials)
template <int KPT>
__global__ __launch_bounds__(kThreads) void bwd_diag_kernel(FrontArrays fa, StepArgs sa, const double* __restrict__ F,
                                                           const double* __restrict__ Inv, const double* __restrict__ P,
                                                           double* __restrict__ V) {
  constexpr int KB = 4 * KPT;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * KB];
  const int q = blockIdx.x;
  const int f = sa.fronts[q];
  const int pc0 = sa.pref_chunks[q];
  const int nch = sa.pref_chunks[q + 1] - pc0;
  const int W = fa.W, kb = sa.kb;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = sa.step * W;
  const int w = min(W, ns - j0);
  const int j1 = j0 + w;
  const double* Ig = Inv + fa.ioff[f] + static_cast<int64_t>(sa.step) * W * W;
  double* Vf = V + fa.voff[f] * kb;
  const int tid = threadIdx.x, o = tid / 4, cg = tid % 4;
  double acc[KPT];
#pragma unroll
  for (int t = 0; t < KPT; ++t) acc[t] = 0.0;
  if (nch == 1) {
    const int rows = static_cast<int>(d) - j1;
    load_panel_chunk<KPT>(As, F + fa.foff[f] + static_cast<int64_t>(j0) * d, d, w, j1, rows);
    load_vec_rows<KPT>(Bs, Vf + static_cast<int64_t>(j1) * kb, kb, rows);
    __syncthreads();
    tile_mac<KPT, true>(As, Bs, rows, o, cg, acc);
    __syncthreads();
  } else if (nch > 1 && o < w) {
    for (int ch = 0; ch < nch; ++ch) {
      const double* Pp = P + static_cast<int64_t>(pc0 + ch) * (TW * KBMAX) + o * KBMAX + cg * KPT;
#pragma unroll
      for (int t = 0; t < KPT; ++t) acc[t] += Pp[t];
    }
  }
  // z = y1 - t  -> Bs ; inv -> As
#pragma unroll
  for (int t = 0; t < KPT; ++t) {
    const int c = cg * KPT + t;
    Bs[o * KB + c] = (o < w && c < kb) ? Vf[static_cast<int64_t>(j0 + o) * kb + c] - acc[t] : 0.0;
  }
  for (int idx = tid; idx < TW * TW; idx += kThreads) {
    const int j = idx / TW, i = idx - j * TW;
    As[j * TLD + i] = (j < w && i < w) ? Ig[j * W + i] : 0.0;  // As[j][i] = inv(i, j)
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < KPT; ++t) acc[t] = 0.0;
  tile_mac<KPT, true>(As, Bs, w, o, cg, acc);  // x1[o] = sum_i inv(i, o) z[i]
  if (o < w) {
#pragma unroll
    for (int t = 0; t < KPT; ++t) {
      const int c = cg * KPT + t;
      if (c < kb) Vf[static_cast<int64_t>(j0 + o) * kb + c] = acc[t];
    }
  }
}

}  // namespace eigd

using namespace eigd;

// ---------------------------------------------------------------------------
struct eigd_factor {
  eigd_ctx* ctx = nullptr;
  const Symbolic* sym = nullptr;  // borrowed; the Python wrapper keeps the symbolic object alive
  // device copies of the symbolic arrays
  int *d_c0 = nullptr, *d_ns = nullptr, *d_bs = nullptr, *d_parent = nullptr, *d_rel = nullptr;
  int64_t *d_foff = nullptr, *d_voff = nullptr, *d_ioff = nullptr, *d_bptr = nullptr;
  int *d_lvl_fronts = nullptr, *d_pref_chunks = nullptr, *d_pref_tiles = nullptr, *d_cs_child = nullptr;
  int64_t *d_a_src = nullptr, *d_a_dst = nullptr;
  int* d_v_src = nullptr;
  double *d_data = nullptr, *d_F = nullptr, *d_Inv = nullptr, *d_V = nullptr, *d_P = nullptr;
  int* d_flag = nullptr;
  size_t bytes = 0;
  int64_t max_chunks = 0;
  std::vector<int> ea_split;  // per (level, slot): grid.y of the extend-add launches
  int64_t data_len = 0;

  FrontArrays fa() const {
    FrontArrays a;
    a.c0 = d_c0;
    a.ns = d_ns;
    a.bs = d_bs;
    a.parent = d_parent;
    a.foff = d_foff;
    a.voff = d_voff;
    a.ioff = d_ioff;
    a.bptr = d_bptr;
    a.rel = d_rel;
    a.W = sym->W;
    return a;
  }
};

namespace {

template <typename T>
int upload(eigd_factor* f, T** dptr, const std::vector<T>& h) {
  size_t bytes = sizeof(T) * std::max<size_t>(h.size(), 1);
  EIGD_HIP(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
  f->bytes += bytes;
  if (!h.empty()) EIGD_HIP(hipMemcpy(*dptr, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return EIGD_OK;
}

int numeric(eigd_factor* f, const double* hdata) {
  const Symbolic& s = *f->sym;
  hipStream_t st = f->ctx->stream;
  EIGD_HIP(hipMemcpyAsync(f->d_data, hdata, sizeof(double) * f->data_len, hipMemcpyHostToDevice, st));
  EIGD_HIP(hipMemsetAsync(f->d_F, 0, sizeof(double) * s.front_doubles, st));
  EIGD_HIP(hipMemsetAsync(f->d_flag, 0, sizeof(int), st));
  {
    const int nb = static_cast<int>(std::min<int64_t>((s.nlower + 255) / 256, 65536));
    hipLaunchKernelGGL(scatter_a_kernel, dim3(std::max(nb, 1)), dim3(256), 0, st, s.nlower, f->d_a_src, f->d_a_dst,
                       f->d_data, f->d_F);
    EIGD_LAUNCH_CHECK();
  }
  const FrontArrays fa = f->fa();
  for (int l = 0; l < s.nlevels; ++l) {
    for (int slot = 0; slot < s.maxslots; ++slot) {
      const size_t rec = static_cast<size_t>(l) * s.maxslots + slot;
      const int cnt = s.cs_ptr[rec + 1] - s.cs_ptr[rec];
      if (cnt == 0) continue;
      hipLaunchKernelGGL(extend_add_kernel, dim3(cnt, f->ea_split[rec]), dim3(kThreads), 0, st, fa,
                         f->d_cs_child + s.cs_ptr[rec], f->d_F);
      EIGD_LAUNCH_CHECK();
    }
    const int* fronts = f->d_lvl_fronts + s.lvl_ptr[l];
    for (int step = 0; step < s.lvl_nsteps[l]; ++step) {
      const int rec = s.ls_ptr[l] + step;
      const int na = s.ls_nactive[rec];
      const int64_t po = s.ls_pref_ptr[rec];
      const int nchunks = s.pref_chunks[po + na];
      const int ntiles = s.pref_tiles[po + na];
      hipLaunchKernelGGL(potrf_inv_kernel, dim3(na), dim3(kThreads), 0, st, fa, fronts, step, f->d_F, f->d_Inv,
                         f->d_flag);
      EIGD_LAUNCH_CHECK();
      if (nchunks > 0) {
        hipLaunchKernelGGL(trsm_kernel, dim3(nchunks), dim3(kThreads), 0, st, fa, fronts, na, step,
                           f->d_pref_chunks + po, f->d_F, f->d_Inv);
        EIGD_LAUNCH_CHECK();
        hipLaunchKernelGGL(syrk_kernel, dim3(ntiles), dim3(kThreads), 0, st, fa, fronts, na, step,
                           f->d_pref_chunks + po, f->d_pref_tiles + po, f->d_F);
        EIGD_LAUNCH_CHECK();
      }
    }
  }
  int flag = 0;
  EIGD_HIP(hipMemcpyAsync(&flag, f->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
  EIGD_HIP(hipStreamSynchronize(st));
  if (flag != 0) {
    set_error("matrix is not positive definite (non-positive pivot in front %d): the shift must lie below the "
              "lowest eigenvalue",
              flag - 1);
    return EIGD_E_NOTSPD;
  }
  return EIGD_OK;
}

template <int KPT>
int sweep(eigd_factor* f, double* dX, int ldx, int kb, double alpha) {
  const Symbolic& s = *f->sym;
  hipStream_t st = f->ctx->stream;
  const FrontArrays fa = f->fa();
  const int64_t total = s.sumd * kb;
  const int gb = static_cast<int>(std::min<int64_t>((total + 255) / 256, 16384));
  hipLaunchKernelGGL(solve_init_kernel, dim3(std::max(gb, 1)), dim3(256), 0, st, s.sumd, kb, f->d_v_src, dX, ldx, alpha,
                     f->d_V);
  EIGD_LAUNCH_CHECK();
  // ---- forward: leaves -> root
  for (int l = 0; l < s.nlevels; ++l) {
    for (int slot = 0; slot < s.maxslots; ++slot) {
      const size_t rec = static_cast<size_t>(l) * s.maxslots + slot;
      const int cnt = s.cs_ptr[rec + 1] - s.cs_ptr[rec];
      if (cnt == 0) continue;
      const int split = std::max(1, std::min(64, f->ea_split[rec] / 8));
      hipLaunchKernelGGL(vec_extend_add_kernel, dim3(cnt, split), dim3(kThreads), 0, st, fa,
                         f->d_cs_child + s.cs_ptr[rec], kb, f->d_V);
      EIGD_LAUNCH_CHECK();
    }
    for (int step = 0; step < s.lvl_nsteps[l]; ++step) {
      const int rec = s.ls_ptr[l] + step;
      const int64_t po = s.ls_pref_ptr[rec];
      StepArgs sa;
      sa.fronts = f->d_lvl_fronts + s.lvl_ptr[l];
      sa.pref_chunks = f->d_pref_chunks + po;
      sa.na = s.ls_nactive[rec];
      sa.step = step;
      sa.kb = kb;
      const int nchunks = s.pref_chunks[po + sa.na];
      hipLaunchKernelGGL(fwd_diag_kernel<KPT>, dim3(sa.na), dim3(kThreads), 0, st, fa, sa, f->d_F, f->d_Inv, f->d_V);
      EIGD_LAUNCH_CHECK();
      if (nchunks > sa.na || nchunks > 1) {  // some front has more than one chunk
        hipLaunchKernelGGL(fwd_update_kernel<KPT>, dim3(nchunks), dim3(kThreads), 0, st, fa, sa, f->d_F, f->d_V);
        EIGD_LAUNCH_CHECK();
      }
    }
  }
  // ---- backward: root -> leaves
  for (int l = s.nlevels - 1; l >= 0; --l) {
    const int nf = s.lvl_ptr[l + 1] - s.lvl_ptr[l];
    if (l < s.nlevels - 1 || true) {
      hipLaunchKernelGGL(vec_gather_kernel, dim3(nf, 4), dim3(kThreads), 0, st, fa, f->d_lvl_fronts + s.lvl_ptr[l], kb,
                         f->d_V);
      EIGD_LAUNCH_CHECK();
    }
    for (int step = s.lvl_nsteps[l] - 1; step >= 0; --step) {
      const int rec = s.ls_ptr[l] + step;
      const int64_t po = s.ls_pref_ptr[rec];
      StepArgs sa;
      sa.fronts = f->d_lvl_fronts + s.lvl_ptr[l];
      sa.pref_chunks = f->d_pref_chunks + po;
      sa.na = s.ls_nactive[rec];
      sa.step = step;
      sa.kb = kb;
      const int nchunks = s.pref_chunks[po + sa.na];
      if (nchunks > sa.na || nchunks > 1) {
        hipLaunchKernelGGL(bwd_partial_kernel<KPT>, dim3(nchunks), dim3(kThreads), 0, st, fa, sa, f->d_F, f->d_V,
                           f->d_P);
        EIGD_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL(bwd_diag_kernel<KPT>, dim3(sa.na), dim3(kThreads), 0, st, fa, sa, f->d_F, f->d_Inv, f->d_P,
                         f->d_V);
      EIGD_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(solve_out_kernel, dim3(std::max(gb, 1)), dim3(256), 0, st, s.sumd, kb, f->d_v_src, f->d_V, dX, ldx);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

}  // namespace

extern "C" {

int eigd_symbolic_create(int n, const int32_t* hindptr, const int32_t* hindices, int leaf_size, int panel_width,
                         eigd_symbolic** out) {
  EIGD_REQUIRE(hindptr && hindices && out, "null argument");
  *out = nullptr;
  eigd_symbolic* h = new eigd_symbolic();
  if (!analyze(n, hindptr, hindices, leaf_size, panel_width, h->s)) {
    set_error("symbolic analysis failed: %s", h->s.error.c_str());
    delete h;
    return EIGD_E_INVALID;
  }
  *out = h;
  return EIGD_OK;
}

int eigd_symbolic_free(eigd_symbolic* s) {
  delete s;
  return EIGD_OK;
}

int eigd_symbolic_sizes(eigd_symbolic* h, int64_t* out, int nout) {
  EIGD_REQUIRE(h && out && nout >= 1, "null argument");
  const Symbolic& s = h->s;
  const int64_t v[12] = {s.n,
                         s.nfronts,
                         s.nlevels,
                         s.nnzL,
                         s.front_doubles,
                         s.sumd,
                         s.maxd,
                         static_cast<int64_t>(s.border.size()),
                         s.nlower,
                         static_cast<int64_t>(s.ls_nactive.size()),
                         static_cast<int64_t>(s.flops),
                         s.maxns};
  for (int i = 0; i < nout && i < 12; ++i) out[i] = v[i];
  return EIGD_OK;
}

#define EIGD_GET(NAME, VEC)                                                               \
  if (std::strcmp(name, NAME) == 0) {                                                     \
    const auto& v = VEC;                                                                  \
    EIGD_REQUIRE(cap >= static_cast<int64_t>(v.size()), "buffer too small for %s", NAME); \
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];                                  \
    return static_cast<int>(EIGD_OK);                                                     \
  }

int eigd_symbolic_get_i32(eigd_symbolic* h, const char* name, int32_t* out, int64_t cap) {
  EIGD_REQUIRE(h && name && out, "null argument");
  const Symbolic& s = h->s;
  EIGD_GET("perm", s.perm)
  EIGD_GET("iperm", s.iperm)
  EIGD_GET("f_c0", s.f_c0)
  EIGD_GET("f_ns", s.f_ns)
  EIGD_GET("f_bs", s.f_bs)
  EIGD_GET("f_parent", s.f_parent)
  EIGD_GET("f_level", s.f_level)
  EIGD_GET("f_slot", s.f_slot)
  EIGD_GET("f_npanels", s.f_npanels)
  EIGD_GET("border", s.border)
  EIGD_GET("rel", s.rel)
  EIGD_GET("lvl_ptr", s.lvl_ptr)
  EIGD_GET("lvl_fronts", s.lvl_fronts)
  EIGD_GET("lvl_nsteps", s.lvl_nsteps)
  EIGD_GET("v_src", s.v_src)
  set_error("unknown int32 symbolic array '%s'", name);
  return EIGD_E_INVALID;
}

int eigd_symbolic_get_i64(eigd_symbolic* h, const char* name, int64_t* out, int64_t cap) {
  EIGD_REQUIRE(h && name && out, "null argument");
  const Symbolic& s = h->s;
  EIGD_GET("f_bptr", s.f_bptr)
  EIGD_GET("f_foff", s.f_foff)
  EIGD_GET("f_voff", s.f_voff)
  EIGD_GET("f_ioff", s.f_ioff)
  EIGD_GET("a_src", s.a_src)
  EIGD_GET("a_dst", s.a_dst)
  set_error("unknown int64 symbolic array '%s'", name);
  return EIGD_E_INVALID;
}
#undef EIGD_GET

int eigd_factor_free(eigd_factor* f) {
  if (!f) return EIGD_OK;
  if (f->ctx && f->ctx->stream) (void)hipStreamSynchronize(f->ctx->stream);
  void* ptrs[] = {f->d_c0,        f->d_ns,          f->d_bs,         f->d_parent,   f->d_rel,   f->d_foff,
                  f->d_voff,      f->d_ioff,        f->d_bptr,       f->d_lvl_fronts, f->d_pref_chunks,
                  f->d_pref_tiles, f->d_cs_child,   f->d_a_src,      f->d_a_dst,    f->d_v_src, f->d_data,
                  f->d_F,         f->d_Inv,         f->d_V,          f->d_P,        f->d_flag};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete f;
  return EIGD_OK;
}

int eigd_factor_create(eigd_ctx* ctx, eigd_symbolic* h, const double* hdata, eigd_factor** out) {
  EIGD_REQUIRE(ctx && h && hdata && out, "null argument");
  *out = nullptr;
  const Symbolic& s = h->s;
  EIGD_HIP(hipSetDevice(ctx->device));
  size_t free_b = 0, total_b = 0;
  EIGD_HIP(hipMemGetInfo(&free_b, &total_b));
  int64_t max_chunks = 1;
  for (size_t rec = 0; rec < s.ls_nactive.size(); ++rec)
    max_chunks = std::max<int64_t>(max_chunks, s.pref_chunks[s.ls_pref_ptr[rec] + s.ls_nactive[rec]]);
  const size_t need = sizeof(double) * (static_cast<size_t>(s.front_doubles) + s.inv_doubles + s.sumd * KBMAX +
                                        max_chunks * TW * KBMAX) +
                      16 * s.a_src.size() + (size_t(64) << 20);
  if (need > free_b) {
    set_error("factor needs %.2f GiB of device memory, %.2f GiB free", need / 1073741824.0, free_b / 1073741824.0);
    return EIGD_E_HIP;
  }
  eigd_factor* f = new eigd_factor();
  f->ctx = ctx;
  f->sym = &h->s;
  f->max_chunks = max_chunks;
  int rc = EIGD_OK;
#define UP(dst, vec)                          \
  if (rc == EIGD_OK) rc = upload(f, &f->dst, vec);
  UP(d_c0, s.f_c0)
  UP(d_ns, s.f_ns)
  UP(d_bs, s.f_bs)
  UP(d_parent, s.f_parent)
  UP(d_rel, s.rel)
  UP(d_foff, s.f_foff)
  UP(d_voff, s.f_voff)
  UP(d_ioff, s.f_ioff)
  UP(d_bptr, s.f_bptr)
  UP(d_lvl_fronts, s.lvl_fronts)
  UP(d_pref_chunks, s.pref_chunks)
  UP(d_pref_tiles, s.pref_tiles)
  UP(d_cs_child, s.cs_child)
  UP(d_a_src, s.a_src)
  UP(d_a_dst, s.a_dst)
  UP(d_v_src, s.v_src)
#undef UP
  int64_t maxsrc = 0;
  for (int64_t e : s.a_src) maxsrc = std::max(maxsrc, e);
  f->data_len = maxsrc + 1;
  // the caller's CSR data array may be longer (upper-triangle entries after the last lower one): we copy a prefix
  auto dmalloc = [&](double** p, size_t count) -> int {
    if (rc != EIGD_OK) return rc;
    size_t b = sizeof(double) * std::max<size_t>(count, 1);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), b);
    if (e != hipSuccess) {
      set_error("hipMalloc of %.2f GiB failed: %s", b / 1073741824.0, hipGetErrorString(e));
      return EIGD_E_HIP;
    }
    f->bytes += b;
    return EIGD_OK;
  };
  rc = dmalloc(&f->d_data, f->data_len);
  rc = dmalloc(&f->d_F, s.front_doubles);
  rc = dmalloc(&f->d_Inv, s.inv_doubles);
  rc = dmalloc(&f->d_V, static_cast<size_t>(s.sumd) * KBMAX);
  rc = dmalloc(&f->d_P, static_cast<size_t>(max_chunks) * TW * KBMAX);
  if (rc == EIGD_OK) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->d_flag), sizeof(int));
    if (e != hipSuccess) {
      set_error("hipMalloc failed: %s", hipGetErrorString(e));
      rc = EIGD_E_HIP;
    }
  }
  if (rc != EIGD_OK) {
    eigd_factor_free(f);
    return rc;
  }
  // extend-add launch shapes
  f->ea_split.assign(static_cast<size_t>(s.nlevels) * s.maxslots, 1);
  for (size_t rec = 0; rec + 1 < s.cs_ptr.size(); ++rec) {
    int64_t maxbs = 1;
    for (int q = s.cs_ptr[rec]; q < s.cs_ptr[rec + 1]; ++q) maxbs = std::max<int64_t>(maxbs, s.f_bs[s.cs_child[q]]);
    int64_t split = (maxbs * maxbs + kThreads * 32 - 1) / (kThreads * 32);
    f->ea_split[rec] = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(split, 512)));
  }
  EIGD_HIP(hipMemsetAsync(f->d_Inv, 0, sizeof(double) * std::max<int64_t>(s.inv_doubles, 1), ctx->stream));
  rc = numeric(f, hdata);
  if (rc != EIGD_OK) {
    eigd_factor_free(f);
    return rc;
  }
  *out = f;
  return EIGD_OK;
}

int eigd_factor_refactor(eigd_factor* f, const double* hdata) {
  EIGD_REQUIRE(f && hdata, "null argument");
  return numeric(f, hdata);
}

int eigd_factor_solve(eigd_factor* f, double* dX, int ldx, int k, double alpha) {
  EIGD_REQUIRE(f && dX, "null argument");
  EIGD_REQUIRE(k >= 1 && ldx >= k, "bad block shape k=%d ldx=%d", k, ldx);
  for (int c0 = 0; c0 < k; c0 += KBMAX) {
    const int kb = std::min(KBMAX, k - c0);
    int rc;
    if (kb <= 4)
      rc = sweep<1>(f, dX + c0, ldx, kb, alpha);
    else if (kb <= 8)
      rc = sweep<2>(f, dX + c0, ldx, kb, alpha);
    else if (kb <= 16)
      rc = sweep<4>(f, dX + c0, ldx, kb, alpha);
    else
      rc = sweep<8>(f, dX + c0, ldx, kb, alpha);
    if (rc != EIGD_OK) return rc;
  }
  return EIGD_OK;
}

int eigd_factor_stats(eigd_factor* f, double* out, int nout) {
  EIGD_REQUIRE(f && out && nout >= 1, "null argument");
  const double v[4] = {static_cast<double>(f->sym->nnzL), static_cast<double>(f->bytes), f->sym->flops,
                       static_cast<double>(f->sym->nfronts)};
  for (int i = 0; i < nout && i < 4; ++i) out[i] = v[i];
  return EIGD_OK;
}

int eigd_factor_solve_bytes(eigd_factor* f, int k, double* bytes) {
  EIGD_REQUIRE(f && bytes && k >= 1, "bad argument");
  // forward + backward each stream L once (8 B per entry) and the inverse diagonal blocks once;
  // the right-hand side block is read and written once (16 n k).
  const Symbolic& s = *f->sym;
  const int passes = (k + KBMAX - 1) / KBMAX;
  *bytes = passes * 2.0 * 8.0 * static_cast<double>(s.nnzL) + 16.0 * static_cast<double>(s.n) * k;
  return EIGD_OK;
}

}  // extern "C"
