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
  const int* poff;  // first partial slab of the front's border products, -1 if it has a single tile group
  double* sgn;      // +-1 per (permuted) column: A = L S L^T with S = diag(sgn); all +1 for a positive definite matrix
  int W;
  int BG;           // border tiles per group
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
  __shared__ double sg[TW];
  for (int j = 0; j < w; ++j) {
    if (tid == 0) {
      double p = S[j * TLD + j];
      const double ap = fabs(p);
      if (!(ap > 0.0) || !(ap < 1.0e300)) {  // zero / NaN / inf pivot: the matrix is singular to working precision
        atomicCAS(flag, 0, f + 1);
        p = 1.0;
      }
      if (p < 0.0) atomicAdd(flag + 1, 1);   // inertia: number of negative pivots
      sg[j] = (p < 0.0) ? -1.0 : 1.0;
      S[j * TLD + j] = sqrt(fabs(p));
    }
    __syncthreads();
    const double dj = S[j * TLD + j] * sg[j];   // a_ij = l_ij * s_j * l_jj
    for (int i = j + 1 + tid; i < w; i += kThreads) S[j * TLD + i] /= dj;
    __syncthreads();
    const int m = w - j - 1;
    const double sj = sg[j];
    for (int idx = tid; idx < m * m; idx += kThreads) {
      const int cc = idx / m, ii = idx - cc * m;
      if (ii >= cc) S[(j + 1 + cc) * TLD + j + 1 + ii] -= sj * S[j * TLD + j + 1 + ii] * S[j * TLD + j + 1 + cc];
    }
    __syncthreads();
  }
  if (tid < w) fa.sgn[fa.c0[f] + j0 + tid] = sg[tid];
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
  const double* sgp = fa.sgn + fa.c0[f] + j0;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (r0 + i < rows && c0 + c < w) Fp[static_cast<int64_t>(c0 + c) * d + r0 + i] = acc[i][c] * sgp[c0 + c];
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
  const double* sgp = fa.sgn + fa.c0[f] + j0;
  for (int idx = tid; idx < TW * TW; idx += kThreads) {
    const int k = idx / TW, r = idx - k * TW;
    As[k * TLD + r] = (k < w && r < rows_i) ? Lp[static_cast<int64_t>(k) * d + Ri + r] : 0.0;
    Bs[k * TLD + r] = (k < w && r < rows_j) ? sgp[k] * Lp[static_cast<int64_t>(k) * d + Rj + r] : 0.0;  // S folded in
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

// ---------------------------------------------------------------------------
// Tile products of the sweeps.  A workgroup produces a 64 x KB block of outputs (KB = 4*KPT >= k)
// from a 64 x 64 tile As of the factor and a 64 x KB block Bs of right-hand sides, both in LDS:
//   TRANS == false : out[o][c] += sum_k As[k*TLD + o] * Bs[k*BLD + c]
//   TRANS == true  : out[o][c] += sum_k As[o*TLD + k] * Bs[k*BLD + c]
//
//  * k <= 8  (KPT 1, 2): vector FMAs, lane (o = tid/4, cg = tid%4) owns KPT columns of row o.
//  * k > 8   (KPT 4, 8): v_mfma_f64_16x16x4_f64.  Wave w owns output rows 16w..16w+15 and all KB/16
//    column tiles; per K-step of 4 a lane feeds ONE double of A and one of B per tile, so the LDS
//    traffic per flop is ~9x lower than the vector form (which is LDS-bound at KB = 32).  Result map of
//    the f64 MFMA: acc[reg] <-> (row = (lane>>4) + 4*reg, col = lane&15) inside the 16 x 16 tile.
// ---------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int KPT>
struct Tile {
  static constexpr bool kMfma = (KPT >= 4);
  static constexpr int KB = 4 * KPT;
  static constexpr int NT = kMfma ? KB / 16 : 1;            // 16-wide column tiles per wave
  static constexpr int BLD = (KB == 32) ? 48 : KB;          // LDS row stride of Bs (48: conflict-free b-operand reads)
  static constexpr int NOUT = kMfma ? 4 * NT : KPT;         // outputs per lane

  __device__ static __forceinline__ void coords(int t, int& row, int& col) {
    if (kMfma) {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      row = 16 * wave + (lane >> 4) + 4 * (t & 3);
      col = 16 * (t >> 2) + (lane & 15);
    } else {
      row = threadIdx.x >> 2;
      col = (threadIdx.x & 3) * KPT + t;
    }
  }

  template <bool TRANS>
  __device__ static __forceinline__ void mac(const double* __restrict__ As, const double* __restrict__ Bs, int kdim,
                                             double (&acc)[NOUT]) {
    if constexpr (kMfma) {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int li = lane & 15, lk = lane >> 4;
      const int o = 16 * wave + li;
      double4_t c[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) c[n] = double4_t{acc[4 * n], acc[4 * n + 1], acc[4 * n + 2], acc[4 * n + 3]};
      const int kd = (kdim + 7) & ~7;
      for (int k0 = 0; k0 < kd; k0 += 8) {
#pragma unroll
        for (int kk = 0; kk < 8; kk += 4) {
          const int k = k0 + kk + lk;
          const double a = TRANS ? As[o * TLD + k] : As[k * TLD + o];
#pragma unroll
          for (int n = 0; n < NT; ++n)
            c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[k * BLD + 16 * n + li], c[n], 0, 0, 0);
        }
      }
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[4 * n + r] = c[n][r];
    } else {
      const int o = threadIdx.x >> 2, cg = threadIdx.x & 3;
      const int kd = (kdim + 7) & ~7;
      for (int k0 = 0; k0 < kd; k0 += 8) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = k0 + kk;
          const double a = TRANS ? As[o * TLD + k] : As[k * TLD + o];
          const double* b = Bs + k * BLD + cg * KPT;
#pragma unroll
          for (int q = 0; q < KPT; ++q) acc[q] += a * b[q];
        }
      }
    }
  }
};

struct StepArgs {
  const int* fronts;       // active fronts of this (level, step), npanels descending
  const int* pref_chunks;  // na + 1 : row chunks below the panel
  const int* pref_work;    // na + 1 : max(1, chunks)
  int na;
  int step;
  int kb;
};

// Tile loaders.  All global loads of a tile are issued before the first LDS store (register
// staging, fully unrolled): a 64 x 64 tile costs one memory latency, not sixteen.
constexpr int TILE_IT = TW * TW / kThreads;  // 16 elements per lane

// As[j*TLD + r] = L(row0 + r, j0 + j), zero padded to 64 x 64
__device__ __forceinline__ void load_panel_chunk(double* As, const double* __restrict__ Lp, int64_t d, int w, int row0,
                                                 int rows) {
  double tmp[TILE_IT];
  const int r = threadIdx.x & (TW - 1), jb = threadIdx.x >> 6;  // lane -> (row, first column); columns step by 4
#pragma unroll
  for (int it = 0; it < TILE_IT; ++it) {
    const int j = jb + it * (kThreads / TW);
    tmp[it] = (j < w && r < rows) ? Lp[static_cast<int64_t>(j) * d + row0 + r] : 0.0;
  }
#pragma unroll
  for (int it = 0; it < TILE_IT; ++it) As[(jb + it * (kThreads / TW)) * TLD + r] = tmp[it];
}

// As[r*TLD + c] = M(r, c) for a tile stored with element (r, c) at c*ld + r (rows contiguous), zero padded
__device__ __forceinline__ void load_tile_transposed(double* As, const double* __restrict__ M, int64_t ld, int nrows,
                                                     int ncols) {
  double tmp[TILE_IT];
  const int r = threadIdx.x & (TW - 1), cb = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < TILE_IT; ++it) {
    const int c = cb + it * (kThreads / TW);
    tmp[it] = (c < ncols && r < nrows) ? M[static_cast<int64_t>(c) * ld + r] : 0.0;
  }
#pragma unroll
  for (int it = 0; it < TILE_IT; ++it) As[r * TLD + cb + it * (kThreads / TW)] = tmp[it];
}

// As[j*TLD + i] = inv(i, j) of the step's diagonal block, zero padded
__device__ __forceinline__ void load_inverse(double* As, const double* __restrict__ Ig, int W, int w) {
  double tmp[TILE_IT];
  const int i = threadIdx.x & (TW - 1), jb = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < TILE_IT; ++it) {
    const int j = jb + it * (kThreads / TW);
    tmp[it] = (j < w && i < w && i >= j) ? Ig[j * W + i] : 0.0;  // lower triangular: the zero half is not fetched
  }
#pragma unroll
  for (int it = 0; it < TILE_IT; ++it) As[(jb + it * (kThreads / TW)) * TLD + i] = tmp[it];
}

// Bs[r*BLD + c] = Vrows[rowmap(r)*kb + c] (r < rows), zero padded to 64 x KB; rel == nullptr: rowmap(r) = r.
// Optionally mirrored into Vdst[r*kb + c].
template <int KPT>
__device__ __forceinline__ void load_vec_rows(double* Bs, const double* Vsrc, const int* __restrict__ rel, int kb,
                                              int rows, double* Vdst) {
  using T = Tile<KPT>;
  constexpr int IT = TW * T::KB / kThreads;  // KPT elements per lane
  double tmp[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = threadIdx.x + it * kThreads;
    const int r = idx / T::KB, c = idx - r * T::KB;
    const int64_t src = (rel != nullptr && r < rows) ? rel[r] : r;
    tmp[it] = (r < rows && c < kb) ? Vsrc[src * kb + c] : 0.0;
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = threadIdx.x + it * kThreads;
    const int r = idx / T::KB, c = idx - r * T::KB;
    Bs[r * T::BLD + c] = tmp[it];
    if (Vdst != nullptr && r < rows && c < kb) Vdst[static_cast<int64_t>(r) * kb + c] = tmp[it];
  }
}

// Forward step of one panel, fused: every workgroup recomputes y1 = inv(L11) v1 (64 x 64 x k
// flops, the inverse block comes from L2) and then updates ITS chunk of the rows below,
// v2 -= L21 y1.  y1 goes to the Y workspace (chunk 0 writes it) so v1 stays read-only during the
// launch: one launch per (level, step) instead of a diagonal launch plus an update launch.
template <int KPT>
__global__ __launch_bounds__(kThreads) void fwd_step_kernel(FrontArrays fa, StepArgs sa, const double* __restrict__ F,
                                                           const double* __restrict__ Inv, double* __restrict__ V,
                                                           double* __restrict__ Y) {
  using T = Tile<KPT>;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * T::BLD];
  const int q = find_slot(sa.pref_work, sa.na, blockIdx.x);
  const int chunk = blockIdx.x - sa.pref_work[q];
  const int f = sa.fronts[q];
  const int nch = sa.pref_chunks[q + 1] - sa.pref_chunks[q];
  const int W = fa.W, kb = sa.kb;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = sa.step * W;
  const int w = min(W, ns - j0);
  const int j1 = j0 + w;
  double* Vf = V + fa.voff[f] * kb;
  load_inverse(As, Inv + fa.ioff[f] + static_cast<int64_t>(sa.step) * W * W, W, w);
  load_vec_rows<KPT>(Bs, Vf + static_cast<int64_t>(j0) * kb, nullptr, kb, w, nullptr);
  __syncthreads();
  double acc[T::NOUT];
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  T::template mac<false>(As, Bs, w, acc);
  __syncthreads();
  double* Yf = Y + (fa.voff[f] + j0) * kb;
  const double* sgp = fa.sgn + fa.c0[f] + j0;
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    Bs[o * T::BLD + c] = (o < w) ? acc[t] : 0.0;
    if (chunk == 0 && o < w && c < kb) Yf[static_cast<int64_t>(o) * kb + c] = sgp[o] * acc[t];  // L^T x = S z
  }
  if (nch == 0) return;
  const int row0 = j1 + chunk * TW;
  const int rows = min(TW, static_cast<int>(d) - row0);
  load_panel_chunk(As, F + fa.foff[f] + static_cast<int64_t>(j0) * d, d, w, row0, rows);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  T::template mac<false>(As, Bs, w, acc);
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    if (o < rows && c < kb) Vf[static_cast<int64_t>(row0 + o) * kb + c] -= acc[t];
  }
}

// Backward, border rows: y(own columns) -= L21(border)^T x(border).  A workgroup owns one
// 64-column chunk of a front and one group of BG border tiles, which it walks in order; the border
// values of x are read straight from the parent's slice of V through the relative index list (group 0
// of chunk 0 also mirrors them into the front's own slice for its children).  Fronts whose border
// fits one group subtract directly (and single-panel ones are finished here); the others write a
// partial slab that bwd_fold_kernel adds in fixed order -- no atomics.
template <int KPT>
__global__ __launch_bounds__(kThreads) void bwd_border_kernel(FrontArrays fa, const int* __restrict__ fronts, int nf,
                                                             const int* __restrict__ pref_bwork, int kb,
                                                             const double* __restrict__ F,
                                                             const double* __restrict__ Inv, double* V,
                                                             double* __restrict__ Y, double* __restrict__ P) {
  using T = Tile<KPT>;
  const int q = find_slot(pref_bwork, nf, blockIdx.x);
  const int f = fronts[q];
  const int p = fa.parent[f];
  const int bs = fa.bs[f];
  const int W = fa.W;
  const int ns = fa.ns[f];
  if ((p < 0 || bs == 0) && ns > W) return;  // nothing to fold in; the step kernels do the rest
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * T::BLD];
  const int ntiles = (bs + TW - 1) / TW;
  const int ngroups = max(1, (ntiles + fa.BG - 1) / fa.BG);
  const int local = blockIdx.x - pref_bwork[q];
  const int cc = local / ngroups, rg = local - cc * ngroups;
  const int64_t d = ns + bs;
  const int c0 = cc * W;
  const int wc = min(W, ns - c0);
  const int* __restrict__ rel = fa.rel + fa.bptr[f];
  const double* Vp = V + fa.voff[p >= 0 ? p : f] * kb;
  double* Vb = V + (fa.voff[f] + ns) * kb;
  const double* Fc = F + fa.foff[f] + static_cast<int64_t>(c0) * d + ns;  // element (border r, col c) at c*d + r
  double acc[T::NOUT];
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  const int rbeg = rg * fa.BG * TW, rend = min(bs, (rg + 1) * fa.BG * TW);
  for (int r0 = rbeg; r0 < rend; r0 += TW) {
    const int rows = min(TW, bs - r0);
    load_tile_transposed(As, Fc + r0, d, rows, wc);  // As[r*TLD + c] = L(ns + r0 + r, c0 + c)
    load_vec_rows<KPT>(Bs, Vp, rel + r0, kb, rows, cc == 0 ? Vb + static_cast<int64_t>(r0) * kb : nullptr);
    __syncthreads();
    T::template mac<false>(As, Bs, rows, acc);
    __syncthreads();
  }
  double* Yf = Y + (fa.voff[f] + c0) * kb;
  if (ngroups > 1) {  // partial slab [cc][rg], 64 x KBMAX
    double* Pp = P + (static_cast<int64_t>(fa.poff[f]) + local) * (TW * KBMAX);
#pragma unroll
    for (int t = 0; t < T::NOUT; ++t) {
      int o, c;
      T::coords(t, o, c);
      Pp[o * KBMAX + c] = acc[t];
    }
    return;
  }
  if (ns > W) {  // multi-panel front, single group: fold into the right-hand side directly
#pragma unroll
    for (int t = 0; t < T::NOUT; ++t) {
      int o, c;
      T::coords(t, o, c);
      if (o < wc && c < kb) Yf[static_cast<int64_t>(o) * kb + c] -= acc[t];
    }
    return;
  }
  // single-panel front with a single group (the leaves and small separators): finish here
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    Bs[o * T::BLD + c] = (o < wc && c < kb) ? Yf[static_cast<int64_t>(o) * kb + c] - acc[t] : 0.0;
  }
  load_inverse(As, Inv + fa.ioff[f], W, wc);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  T::template mac<true>(As, Bs, wc, acc);
  double* Vf = V + fa.voff[f] * kb;
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    if (o < wc && c < kb) Vf[static_cast<int64_t>(o) * kb + c] = acc[t];
  }
}


// y(chunk) -= sum over groups of the partial slabs, groups in ascending order
__global__ __launch_bounds__(kThreads) void bwd_fold_kernel(FrontArrays fa, const int* __restrict__ fronts, int nf,
                                                           const int* __restrict__ pref_panels, int kb,
                                                           const double* __restrict__ P, double* __restrict__ Y) {
  const int q = find_slot(pref_panels, nf, blockIdx.x);
  const int f = fronts[q];
  const int po = fa.poff[f];
  if (po < 0) return;
  const int cc = blockIdx.x - pref_panels[q];
  const int W = fa.W;
  const int ns = fa.ns[f], bs = fa.bs[f];
  const int ntiles = (bs + TW - 1) / TW;
  const int ngroups = (ntiles + fa.BG - 1) / fa.BG;
  const int wc = min(W, ns - cc * W);
  double* Yf = Y + (fa.voff[f] + cc * W) * kb;
  const double* Pp = P + (static_cast<int64_t>(po) + cc * ngroups) * (TW * KBMAX);
  for (int idx = threadIdx.x; idx < wc * kb; idx += kThreads) {
    const int o = idx / kb, c = idx - o * kb;
    double s = 0.0;
    for (int g = 0; g < ngroups; ++g) s += Pp[static_cast<int64_t>(g) * (TW * KBMAX) + o * KBMAX + c];
    Yf[idx] -= s;
  }
}

// Backward step of one panel, fused and right-looking: every workgroup recomputes
// x1 = inv(L11)^T y1, workgroup 0 stores it in V, and workgroup cc folds it into the 64 earlier
// columns it owns: y(cc) -= L(panel rows, cc columns)^T x1.  No partial sums, one launch per step.
template <int KPT>
__global__ __launch_bounds__(kThreads) void bwd_step_kernel(FrontArrays fa, StepArgs sa, const double* __restrict__ F,
                                                           const double* __restrict__ Inv, double* __restrict__ V,
                                                           double* __restrict__ Y) {
  using T = Tile<KPT>;
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * T::BLD];
  const int wpf = max(1, sa.step);  // workgroups per front: one per earlier 64-column chunk
  const int q = blockIdx.x / wpf;
  const int cc = blockIdx.x - q * wpf;
  const int f = sa.fronts[q];
  const int W = fa.W, kb = sa.kb;
  const int ns = fa.ns[f];
  if (ns <= W && fa.poff[f] < 0) return;  // single-panel, single-group fronts are finished by bwd_border_kernel
  const int64_t d = ns + fa.bs[f];
  const int j0 = sa.step * W;
  const int w = min(W, ns - j0);
  double* Yf = Y + fa.voff[f] * kb;
  load_inverse(As, Inv + fa.ioff[f] + static_cast<int64_t>(sa.step) * W * W, W, w);
  load_vec_rows<KPT>(Bs, Yf + static_cast<int64_t>(j0) * kb, nullptr, kb, w, nullptr);
  __syncthreads();
  double acc[T::NOUT];
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  T::template mac<true>(As, Bs, w, acc);  // x1[o] = sum_i inv(i, o) y1[i]
  __syncthreads();
  double* Vf = V + (fa.voff[f] + j0) * kb;
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    Bs[o * T::BLD + c] = (o < w) ? acc[t] : 0.0;
    if (cc == 0 && o < w && c < kb) Vf[static_cast<int64_t>(o) * kb + c] = acc[t];
  }
  if (sa.step == 0) return;
  // As[r*TLD + c] = L(j0 + r, cc*W + c): the panel's rows in the columns of chunk cc (full width W)
  load_tile_transposed(As, F + fa.foff[f] + static_cast<int64_t>(cc) * W * d + j0, d, w, W);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  T::template mac<false>(As, Bs, w, acc);
  double* Yc = Yf + static_cast<int64_t>(cc) * W * kb;
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    if (o < W && c < kb) Yc[static_cast<int64_t>(o) * kb + c] -= acc[t];
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
  int *d_pref_work = nullptr, *d_pref_panels = nullptr, *d_pref_bwork = nullptr, *d_poff = nullptr;
  double* d_P = nullptr;
  int64_t *d_a_src = nullptr, *d_a_dst = nullptr;
  int* d_v_src = nullptr;
  double *d_data = nullptr, *d_F = nullptr, *d_Inv = nullptr, *d_V = nullptr, *d_Y = nullptr, *d_sgn = nullptr;
  int n_negative = 0;
  int* d_flag = nullptr;
  size_t bytes = 0;
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
    a.poff = d_poff;
    a.sgn = d_sgn;
    a.W = sym->W;
    a.BG = sym->BG;
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
  EIGD_HIP(hipMemsetAsync(f->d_flag, 0, 2 * sizeof(int), st));
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
  int flag[2] = {0, 0};
  EIGD_HIP(hipMemcpyAsync(flag, f->d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  EIGD_HIP(hipStreamSynchronize(st));
  f->n_negative = flag[1];
  if (flag[0] != 0) {
    set_error("zero or non-finite pivot in front %d: the (shifted) matrix is singular to working precision -- move the "
              "shift away from an eigenvalue",
              flag[0] - 1);
    return EIGD_E_NOTSPD;
  }
  return EIGD_OK;
}

template <int KPT>
int sweep(eigd_factor* f, hipStream_t st, double* wV, double* wY, double* wP, const double* dIn, int ldin, double* dX,
          int ldx, int kb, double alpha) {
  const Symbolic& s = *f->sym;
  const FrontArrays fa = f->fa();
  const int64_t total = s.sumd * kb;
  const int gb = static_cast<int>(std::min<int64_t>((total + 255) / 256, 16384));
  hipLaunchKernelGGL(solve_init_kernel, dim3(std::max(gb, 1)), dim3(256), 0, st, s.sumd, kb, f->d_v_src, dIn, ldin,
                     alpha, wV);
  EIGD_LAUNCH_CHECK();
  auto step_args = [&](int l, int step) {
    const int rec = s.ls_ptr[l] + step;
    const int64_t po = s.ls_pref_ptr[rec];
    StepArgs sa;
    sa.fronts = f->d_lvl_fronts + s.lvl_ptr[l];
    sa.pref_chunks = f->d_pref_chunks + po;
    sa.pref_work = f->d_pref_work + po;
    sa.na = s.ls_nactive[rec];
    sa.step = step;
    sa.kb = kb;
    return sa;
  };
  // ---- forward: leaves -> root
  for (int l = 0; l < s.nlevels; ++l) {
    for (int slot = 0; slot < s.maxslots; ++slot) {
      const size_t rec = static_cast<size_t>(l) * s.maxslots + slot;
      const int cnt = s.cs_ptr[rec + 1] - s.cs_ptr[rec];
      if (cnt == 0) continue;
      const int split = std::max(1, std::min(64, f->ea_split[rec] / 8));
      hipLaunchKernelGGL(vec_extend_add_kernel, dim3(cnt, split), dim3(kThreads), 0, st, fa,
                         f->d_cs_child + s.cs_ptr[rec], kb, wV);
      EIGD_LAUNCH_CHECK();
    }
    for (int step = 0; step < s.lvl_nsteps[l]; ++step) {
      const StepArgs sa = step_args(l, step);
      const int nwork = s.pref_work[s.ls_pref_ptr[s.ls_ptr[l] + step] + sa.na];
      hipLaunchKernelGGL(fwd_step_kernel<KPT>, dim3(nwork), dim3(kThreads), 0, st, fa, sa, f->d_F, f->d_Inv, wV,
                         wY);
      EIGD_LAUNCH_CHECK();
    }
  }
  // ---- backward: root -> leaves.  Y holds the right-hand sides, V receives the solution.
  for (int l = s.nlevels - 1; l >= 0; --l) {
    const int nf = s.lvl_ptr[l + 1] - s.lvl_ptr[l];
    const int64_t pp = s.lvl_pp_ptr[l];
    const int npan = s.pref_panels[pp + nf];
    const int nbw = s.pref_bwork[pp + nf];
    hipLaunchKernelGGL(bwd_border_kernel<KPT>, dim3(nbw), dim3(kThreads), 0, st, fa, f->d_lvl_fronts + s.lvl_ptr[l],
                       nf, f->d_pref_bwork + pp, kb, f->d_F, f->d_Inv, wV, wY, wP);
    EIGD_LAUNCH_CHECK();
    if (nbw > npan) {  // some front of the level has more than one tile group
      hipLaunchKernelGGL(bwd_fold_kernel, dim3(npan), dim3(kThreads), 0, st, fa, f->d_lvl_fronts + s.lvl_ptr[l], nf,
                         f->d_pref_panels + pp, kb, wP, wY);
      EIGD_LAUNCH_CHECK();
    }
    for (int step = s.lvl_nsteps[l] - 1; step >= 0; --step) {
      const StepArgs sa = step_args(l, step);
      hipLaunchKernelGGL(bwd_step_kernel<KPT>, dim3(sa.na * std::max(1, step)), dim3(kThreads), 0, st, fa, sa, f->d_F,
                         f->d_Inv, wV, wY);
      EIGD_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(solve_out_kernel, dim3(std::max(gb, 1)), dim3(256), 0, st, s.sumd, kb, f->d_v_src, wV, dX, ldx);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

}  // namespace

extern "C" {

int eigd_symbolic_create(int n, const int32_t* hindptr, const int32_t* hindices, int leaf_size, int panel_width,
                         eigd_symbolic** out) {
  return eigd_symbolic_create_geom(n, hindptr, hindices, leaf_size, panel_width, 0, nullptr, out);
}

int eigd_symbolic_create_geom(int n, const int32_t* hindptr, const int32_t* hindices, int leaf_size, int panel_width,
                              int dim, const double* hcoords, eigd_symbolic** out) {
  EIGD_REQUIRE(hindptr && hindices && out, "null argument");
  EIGD_REQUIRE(hcoords == nullptr || (dim >= 1 && dim <= 3), "coordinates need 1 <= dim <= 3");
  *out = nullptr;
  eigd_symbolic* h = new eigd_symbolic();
  if (!analyze(n, hindptr, hindices, leaf_size, panel_width, h->s, dim, hcoords)) {
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
                  f->d_F,         f->d_Inv,         f->d_V,          f->d_Y,        f->d_flag,  f->d_pref_work,
                  f->d_pref_panels, f->d_pref_bwork, f->d_poff,     f->d_P,      f->d_sgn};
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
  const size_t need = sizeof(double) * (static_cast<size_t>(s.front_doubles) + s.inv_doubles + 2 * s.sumd * KBMAX + s.nslabs * TW * KBMAX) +
                      16 * s.a_src.size() + (size_t(64) << 20);
  if (need > free_b) {
    set_error("factor needs %.2f GiB of device memory, %.2f GiB free", need / 1073741824.0, free_b / 1073741824.0);
    return EIGD_E_HIP;
  }
  eigd_factor* f = new eigd_factor();
  f->ctx = ctx;
  f->sym = &h->s;
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
  UP(d_pref_work, s.pref_work)
  UP(d_pref_panels, s.pref_panels)
  UP(d_pref_bwork, s.pref_bwork)
  UP(d_poff, s.f_poff)
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
  rc = dmalloc(&f->d_Y, static_cast<size_t>(s.sumd) * KBMAX);
  rc = dmalloc(&f->d_sgn, static_cast<size_t>(s.n));
  rc = dmalloc(&f->d_P, static_cast<size_t>(std::max<int64_t>(s.nslabs, 1)) * TW * KBMAX);
  if (rc == EIGD_OK) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->d_flag), 2 * sizeof(int));
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
  return eigd_factor_solve_to(f, dX, ldx, dX, ldx, k, alpha);
}

static int solve_blocks(eigd_factor* f, hipStream_t st, double* wV, double* wY, double* wP, const double* dIn, int ldin,
                        double* dOut, int ldout, int k, double alpha) {
  EIGD_REQUIRE(f && dIn && dOut, "null argument");
  EIGD_REQUIRE(k >= 1 && ldin >= k && ldout >= k, "bad block shape k=%d ldin=%d ldout=%d", k, ldin, ldout);
  for (int c0 = 0; c0 < k; c0 += KBMAX) {
    const int kb = std::min(KBMAX, k - c0);
    int rc;
    if (kb <= 4)
      rc = sweep<1>(f, st, wV, wY, wP, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    else if (kb <= 8)
      rc = sweep<2>(f, st, wV, wY, wP, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    else if (kb <= 16)
      rc = sweep<4>(f, st, wV, wY, wP, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    else
      rc = sweep<8>(f, st, wV, wY, wP, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    if (rc != EIGD_OK) return rc;
  }
  return EIGD_OK;
}

int eigd_factor_solve_to(eigd_factor* f, const double* dIn, int ldin, double* dOut, int ldout, int k, double alpha) {
  EIGD_REQUIRE(f, "null argument");
  return solve_blocks(f, f->ctx->stream, f->d_V, f->d_Y, f->d_P, dIn, ldin, dOut, ldout, k, alpha);
}

// A lane = a second set of sweep workspaces bound to another context (stream) of the same device: sweeps of
// different lanes run concurrently on the one factor (independent mode groups overlap each other's latency).
struct eigd_lane {
  eigd_factor* f = nullptr;
  eigd_ctx* ctx = nullptr;
  double *V = nullptr, *Y = nullptr, *P = nullptr;
};

int eigd_factor_lane_create(eigd_factor* f, eigd_ctx* ctx, eigd_lane** out) {
  EIGD_REQUIRE(f && ctx && out, "null argument");
  EIGD_REQUIRE(ctx->device == f->ctx->device, "lane context must live on the factor's device");
  *out = nullptr;
  const Symbolic& s = *f->sym;
  eigd_lane* l = new eigd_lane();
  l->f = f;
  l->ctx = ctx;
  const size_t vb = sizeof(double) * std::max<size_t>(static_cast<size_t>(s.sumd) * KBMAX, 1);
  const size_t pb = sizeof(double) * static_cast<size_t>(std::max<int64_t>(s.nslabs, 1)) * TW * KBMAX;
  hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&l->V), vb);
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&l->Y), vb);
  hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&l->P), pb);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    eigd_factor_lane_free(l);
    set_error("hipMalloc failed for a sweep lane");
    return EIGD_E_HIP;
  }
  *out = l;
  return EIGD_OK;
}

int eigd_factor_lane_free(eigd_lane* l) {
  if (!l) return EIGD_OK;
  if (l->ctx && l->ctx->stream) (void)hipStreamSynchronize(l->ctx->stream);
  if (l->V) (void)hipFree(l->V);
  if (l->Y) (void)hipFree(l->Y);
  if (l->P) (void)hipFree(l->P);
  delete l;
  return EIGD_OK;
}

int eigd_factor_lane_solve_to(eigd_lane* l, const double* dIn, int ldin, double* dOut, int ldout, int k, double alpha) {
  EIGD_REQUIRE(l, "null argument");
  return solve_blocks(l->f, l->ctx->stream, l->V, l->Y, l->P, dIn, ldin, dOut, ldout, k, alpha);
}

int eigd_factor_stats(eigd_factor* f, double* out, int nout) {
  EIGD_REQUIRE(f && out && nout >= 1, "null argument");
  const double v[5] = {static_cast<double>(f->sym->nnzL), static_cast<double>(f->bytes), f->sym->flops,
                       static_cast<double>(f->sym->nfronts), static_cast<double>(f->n_negative)};
  for (int i = 0; i < nout && i < 5; ++i) out[i] = v[i];
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
