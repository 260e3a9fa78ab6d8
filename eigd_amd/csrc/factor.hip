// Sparse shift-invert factorisation and multi-RHS triangular sweeps for gfx950.
//
// Replaces SuperLU behind eigd's SpLuOperator (eigd/eigenvector_derivatives.py:11-23).
// The shifted matrix K - sigma M (or K + sigma G below the first buckling load) is
// symmetric positive definite in every example of the reference; the factor is
// L S L^T (S = diag(+-1), all +1 then: plain Cholesky; negative entries = inertia of an
// interior shift, no pivoting) on a nested-dissection ordering:
//
//   numeric : multifrontal.  Every front is a dense d x d square (d = own columns +
//             border) in one big HBM buffer; levels of the assembly tree are processed
//             bottom-up, all fronts of a level batched in the same launches.  Per panel
//             step of W columns: potrf (+ explicit inverse of the W x W diagonal block),
//             trsm as a product with that inverse, trailing update tile by tile.
//   inverses: T = inv(L11) and M21 = L21 inv(L11) per front (two launches over all fronts).
//   solve   : ONE launch per level of the assembly tree and direction.  The forward sweep
//             multiplies the fronts' right-hand sides by [T; M21]; the carry of child number s
//             of a front is written into carry plane s at the parent's rows, so the parent adds
//             its rows of the planes in plane order (no index hop, no two writers per entry).
//             The backward sweep multiplies by the transpose and gathers x of the border rows
//             straight from the caller's block.  Long chains of big fronts are cut over several
//             workgroups that hand partial blocks over in-launch (write-through stores + ticket).
//             No floating-point atomics anywhere: the sweeps are bitwise reproducible.
//
// Per solve the kernels stream nnz(L) doubles twice (forward + backward) plus
// O(sum of front dimensions) x k vector traffic.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "symbolic.h"

struct eigd_symbolic {
  eigd::Symbolic s;
};

namespace eigd {

constexpr int TW = 64;       // tile edge == max panel width == row chunk
constexpr int TLD = TW + 1;  // padded LDS leading dimension
// waves per SIMD the kernels are compiled for (measured, docs/LOG.md): the 32-column forward fragment kernel would spill at
// 168 registers (2), the other fragment kernels run three; the 32-column backward one loses 3-6 % at three (four spilled
// registers) to two without spills; the 32-column buffer-access thin forward kernels: leaf fronts 3, fronts with two
// carry planes 1
constexpr int kFragWavesFwd = 2, kFragWaves = 3, kFragWavesBwd32 = 2, kThinWavesLeaf = 3, kThinWavesKids = 1;
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
  const int64_t* toff;  // offset of the front's inverted triangle T = inv(L11) (ns x ns, column-major)
  int nslot;            // carry planes a parent reads (forward sweep): V is nslot (+1 scratch) planes of vrows rows
  int64_t vrows;
  const int* v_src;     // per row of V: the row of the caller's block it holds (own rows), -1 for border rows
  const int* cmask;     // per row of V: bit s set = carry plane s holds a child's contribution on this row
  const int* bout;      // per border entry (bptr): its row in the caller's block
  double* sgn;          // +-1 per (permuted) column: A = L S L^T with S = diag(sgn); all +1 for a positive definite matrix
  const double* zero;   // one 0.0 and ...
  const int* neg1;      // ... one -1 in device memory: masked-off lanes of the sweeps load from here (address select)
  int W;                // instead of branching around the load, which would serialise the loads of a tile
  int tri;              // 1: the diagonal blocks of T are lower triangular (Cholesky / sign-tracked path); 0: dense 64 x 64
                        // diagonal blocks (Bunch-Kaufman path: pivoting inside the panel), the sweeps skip no part of them
  double pivtol;        // Bunch-Kaufman path: a pivot column whose largest entry inside the panel block is below this
                        // (sqrt(eps) * max |a_ij|) is singular to the panel: its diagonal is set to +-pivtol (static pivot)
};

constexpr int kPlaneCols = 4 + 8 + 16 + 32;  // one set of carry planes per sweep width
constexpr int kMaxS = 4;  // children per front with a carry plane of their own; further children share an extra plane

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

// largest |x_i| per workgroup (the threshold of the static pivots is sqrt(eps) times the largest matrix entry)
__global__ __launch_bounds__(kThreads) void absmax_kernel(int64_t n, const double* __restrict__ x, double* __restrict__ partial) {
  __shared__ double red[kThreads];
  double m = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads)
    m = fmax(m, fabs(x[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
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


// Indefinite shifts (sigma inside the spectrum; examples/crm.py:26, 221 -- SuperLU pivots there, reference 13): the same
// panel step with Bunch-Kaufman pivoting INSIDE the W x W diagonal block.  P A_pp P^T = L D L^T with 1 x 1 and 2 x 2
// pivots (alpha = (1 + sqrt 17) / 8, the bounded-growth rule of LAPACK's dsytf2); every 2 x 2 block is diagonalised,
// D = Q diag(e) Q^T, so that A_pp = M S M^T with M = P^T L Q |e|^(1/2) and S = diag(sign e).  M is a dense block, but
// the rest of the factorisation and the sweeps only ever use inv(M) (trsm as a product, T = inv(M11) explicit), so
// nothing else changes: the permutation never leaves this kernel and the symbolic structure stays what it was.
// Pivots are not delayed to the parent front.  A column that is singular INSIDE its panel block (largest candidate below
// sqrt(eps) |A|: a front whose own block is singular by itself, which SuperLU's partial pivoting over the whole column
// survives) gets a STATIC pivot instead, as SuperLU_DIST / PARDISO do: its diagonal is replaced by +-sqrt(eps) |A|, the
// replacements are counted (flag[2]) and the caller refines every solve against the true matrix a few times.
__global__ __launch_bounds__(kThreads) void ldlt_bk_inv_kernel(FrontArrays fa, const int* __restrict__ fronts, int step,
                                                              double* __restrict__ F, double* __restrict__ Inv,
                                                              int* __restrict__ flag) {
  __shared__ double A[TW * TLD];   // full symmetric working block, column-major: A[j * TLD + i]; columns < k hold L
  __shared__ double X[TW * TLD];   // inv(M), column c at X[c * TLD + .]
  __shared__ double wk[2 * TW];    // the two multiplier columns of a 2 x 2 step
  __shared__ int perm[TW], kind[TW];  // kind[k]: 1 = 1 x 1 pivot, 2 = first row of a 2 x 2 pivot, 0 = its second row
  __shared__ int s_kp, s_kstep, s_bad;
  const int f = fronts[blockIdx.x];
  const int W = fa.W;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int j0 = step * W;
  const int w = min(W, ns - j0);
  double* Fd = F + fa.foff[f] + static_cast<int64_t>(j0) * d + j0;
  const int tid = threadIdx.x;
  const double alpha = 0.6403882032022076;  // (1 + sqrt(17)) / 8
  for (int idx = tid; idx < w * w; idx += kThreads) {
    const int j = idx / w, i = idx - j * w;
    A[j * TLD + i] = (i >= j) ? Fd[static_cast<int64_t>(j) * d + i] : Fd[static_cast<int64_t>(i) * d + j];
  }
  if (tid < w) perm[tid] = tid;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  int k = 0;
  while (k < w) {
    if (tid == 0) {  // pivot choice (dsytf2, lower): at most two scans of w entries
      const double absakk = fabs(A[k * TLD + k]);
      int imax = k;
      double colmax = 0.0;
      for (int i = k + 1; i < w; ++i) {
        const double v = fabs(A[k * TLD + i]);
        if (v > colmax) { colmax = v; imax = i; }
      }
      int kp = k, kstep = 1;
      const double big = fmax(absakk, colmax);
      if (!(big >= 0.0) || !(big < 1.0e300)) {
        s_bad = 1;  // not finite
      } else if (big <= fa.pivtol) {
        // no usable pivot in this column of the panel block: static pivot
        A[k * TLD + k] = (A[k * TLD + k] < 0.0) ? -fa.pivtol : fa.pivtol;
        atomicAdd(flag + 2, 1);
        if (!(fa.pivtol > 0.0)) s_bad = 1;  // (all-zero matrix)
      } else if (absakk < alpha * colmax) {
        double rowmax = 0.0;
        for (int j = k; j < w; ++j)
          if (j != imax) rowmax = fmax(rowmax, fabs(A[j * TLD + imax]));
        if (absakk >= alpha * colmax * (colmax / rowmax)) {
          kp = k;
        } else if (fabs(A[imax * TLD + imax]) >= alpha * rowmax) {
          kp = imax;
        } else {
          kp = imax;
          kstep = 2;
        }
      }
      s_kp = kp;
      s_kstep = kstep;
    }
    __syncthreads();
    if (s_bad) break;
    const int kstep = s_kstep, kp = s_kp;
    const int kk = k + kstep - 1;  // the position the chosen row moves to
    if (kp != kk) {  // symmetric interchange kk <-> kp: rows (all columns, L included), then columns
      for (int j = tid; j < w; j += kThreads) {
        const double t = A[j * TLD + kk];
        A[j * TLD + kk] = A[j * TLD + kp];
        A[j * TLD + kp] = t;
      }
      __syncthreads();
      for (int i = tid; i < w; i += kThreads) {
        const double t = A[kk * TLD + i];
        A[kk * TLD + i] = A[kp * TLD + i];
        A[kp * TLD + i] = t;
      }
      if (tid == 0) {
        const int t = perm[kk];
        perm[kk] = perm[kp];
        perm[kp] = t;
      }
      __syncthreads();
    }
    if (kstep == 1) {
      const double dk = A[k * TLD + k];
      const int m = w - k - 1;
      for (int idx = tid; idx < m * m; idx += kThreads) {  // trailing block, both triangles
        const int cc = idx / m, ii = idx - cc * m;
        A[(k + 1 + cc) * TLD + k + 1 + ii] -= A[k * TLD + k + 1 + ii] * A[k * TLD + k + 1 + cc] / dk;
      }
      __syncthreads();
      for (int i = k + 1 + tid; i < w; i += kThreads) A[k * TLD + i] /= dk;  // the multipliers
      if (tid == 0) kind[k] = 1;
    } else {
      const double a = A[k * TLD + k], b = A[k * TLD + k + 1], c = A[(k + 1) * TLD + k + 1];
      const double det = a * c - b * b;
      const int m = w - k - 2;
      for (int i = tid; i < m; i += kThreads) {  // [w1 w2] = [a_i,k  a_i,k+1] inv(D)
        const double x = A[k * TLD + k + 2 + i], y = A[(k + 1) * TLD + k + 2 + i];
        wk[i] = (x * c - y * b) / det;
        wk[TW + i] = (y * a - x * b) / det;
      }
      __syncthreads();
      for (int idx = tid; idx < m * m; idx += kThreads) {
        const int cc = idx / m, ii = idx - cc * m;
        A[(k + 2 + cc) * TLD + k + 2 + ii] -= wk[ii] * A[k * TLD + k + 2 + cc] + wk[TW + ii] * A[(k + 1) * TLD + k + 2 + cc];
      }
      __syncthreads();
      for (int i = tid; i < m; i += kThreads) {
        A[k * TLD + k + 2 + i] = wk[i];
        A[(k + 1) * TLD + k + 2 + i] = wk[TW + i];
      }
      if (tid == 0) {
        kind[k] = 2;
        kind[k + 1] = 0;
      }
    }
    __syncthreads();
    k += kstep;
  }
  if (s_bad) {
    if (tid == 0) atomicCAS(flag, 0, f + 1);
    // leave something finite behind: the caller reports the failure, later launches of this numeric phase must not fault
    for (int idx = tid; idx < w * w; idx += kThreads) {
      const int j = idx / w, i = idx - j * w;
      A[j * TLD + i] = (i == j) ? 1.0 : 0.0;
    }
    if (tid < w) {
      perm[tid] = tid;
      kind[tid] = 1;
    }
    __syncthreads();
  }
  // inv(M) = |e|^(-1/2) Q^T inv(L) P, one column per lane: x = P e_c, forward substitution with the unit lower L
  // (the sub-diagonal entry inside a 2 x 2 block belongs to D, not to L), then the block-diagonal scaling
  if (tid < w) {
    const int c = tid;
    double* x = X + c * TLD;
    for (int i = 0; i < w; ++i) {
      double sum = (perm[i] == c) ? 1.0 : 0.0;
      const int jend = (kind[i] == 0) ? i - 1 : i;
      for (int j = 0; j < jend; ++j) sum -= A[j * TLD + i] * x[j];
      x[i] = sum;
    }
    for (int i = 0; i < w; ++i) {
      if (kind[i] == 1) {
        x[i] /= sqrt(fabs(A[i * TLD + i]));
      } else if (kind[i] == 2) {
        const double a = A[i * TLD + i], b = A[i * TLD + i + 1], cc = A[(i + 1) * TLD + i + 1];
        // eigen-decomposition of [[a, b], [b, cc]]: rotation by theta with tan(2 theta) = 2 b / (a - cc)
        const double th = 0.5 * atan2(2.0 * b, a - cc);
        const double cs = cos(th), sn = sin(th);
        double e1 = a * cs * cs + 2.0 * b * cs * sn + cc * sn * sn;
        double e2 = a * sn * sn - 2.0 * b * cs * sn + cc * cs * cs;
        // a 2 x 2 pivot block that is singular itself (the matrix' own null direction can hide in one): static pivot
        if (fabs(e1) <= fa.pivtol) {
          e1 = (e1 < 0.0) ? -fa.pivtol : fa.pivtol;
          if (c == 0) atomicAdd(flag + 2, 1);
        }
        if (fabs(e2) <= fa.pivtol) {
          e2 = (e2 < 0.0) ? -fa.pivtol : fa.pivtol;
          if (c == 0) atomicAdd(flag + 2, 1);
        }
        const double x0 = x[i], x1 = x[i + 1];
        x[i] = (cs * x0 + sn * x1) / sqrt(fabs(e1));
        x[i + 1] = (-sn * x0 + cs * x1) / sqrt(fabs(e2));
      }
    }
  }
  // signs of S, inertia
  if (tid < w) {
    double sgv = 1.0;
    if (kind[tid] == 1) {
      sgv = (A[tid * TLD + tid] < 0.0) ? -1.0 : 1.0;
    } else {
      const int i = (kind[tid] == 2) ? tid : tid - 1;
      const double a = A[i * TLD + i], b = A[i * TLD + i + 1], cc = A[(i + 1) * TLD + i + 1];
      const double th = 0.5 * atan2(2.0 * b, a - cc);
      const double cs = cos(th), sn = sin(th);
      const double e = (kind[tid] == 2) ? a * cs * cs + 2.0 * b * cs * sn + cc * sn * sn
                                        : a * sn * sn - 2.0 * b * cs * sn + cc * cs * cs;
      sgv = (e < 0.0) ? -1.0 : 1.0;
    }
    fa.sgn[fa.c0[f] + j0 + tid] = sgv;
    if (sgv < 0.0) atomicAdd(flag + 1, 1);
  }
  __syncthreads();
  double* Ig = Inv + fa.ioff[f] + static_cast<int64_t>(step) * W * W;
  for (int idx = tid; idx < w * w; idx += kThreads) {
    const int j = idx / w, i = idx - j * w;
    Ig[j * W + i] = X[j * TLD + i];
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


// ------------------------------------------------------------------ explicit inverses
// The sweeps never substitute: after the numeric factorisation every front's triangle is inverted,
//   T   = inv(L11)            (dense lower triangular ns x ns, column-major, own buffer)
//   M21 = L21 * inv(L11)      (bs x ns, overwrites L21 inside the front)
// so that a front's share of a sweep is ONE product with [T; M21] (forward) or its transpose (backward):
// one launch per level of the assembly tree and direction instead of one per 64-column panel.  The
// diagonal blocks of an FE separator are well conditioned (cond(L11) ~ 1e1..1e2 at the 1M-dof benchmark),
// the residuals match the substitution form to the last digits.

// T, column block j of one front per workgroup (blocks of the panel width W); rows top-down:
//   X_jj = inv(L_jj) (from the factorisation),  X_ij = -inv(L_ii) * sum_{k=j}^{i-1} L_ik X_kj
__global__ __launch_bounds__(kThreads) void trinv_kernel(FrontArrays fa, const int* __restrict__ tri_pref, int nfronts,
                                                        const double* __restrict__ F, const double* __restrict__ Inv,
                                                        double* Tb) {
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * TLD];
  const int f = find_slot(tri_pref, nfronts, blockIdx.x);
  const int j = blockIdx.x - tri_pref[f];
  const int W = fa.W;
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int nb = (ns + W - 1) / W;
  const int j0 = j * W, wj = min(W, ns - j0);
  const double* Ff = F + fa.foff[f];
  const double* If = Inv + fa.ioff[f];
  double* Tf = Tb + fa.toff[f];
  const int tid = threadIdx.x;
  const int r0 = (tid % 16) * 4, c0 = (tid / 16) * 4;
  for (int idx = tid; idx < W * W; idx += kThreads) {
    const int c = idx / W, i = idx - c * W;
    if (i < wj && c < wj)
      Tf[static_cast<int64_t>(j0 + c) * ns + j0 + i] = If[static_cast<int64_t>(j) * W * W + c * W + i];  // (zeros above the
    // diagonal on the Cholesky path, a dense block on the Bunch-Kaufman path)
  }
  __threadfence();
  __syncthreads();
  for (int i = j + 1; i < nb; ++i) {
    const int i0 = i * W, wi = min(W, ns - i0);
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k = j; k < i; ++k) {
      const int k0 = k * W;  // full block: k < nb - 1
      for (int idx = tid; idx < W * TW; idx += kThreads) {
        const int kk = idx / TW, r = idx - kk * TW;
        As[kk * TLD + r] = (r < wi) ? Ff[static_cast<int64_t>(k0 + kk) * d + i0 + r] : 0.0;        // L(i0 + r, k0 + kk)
      }
      for (int idx = tid; idx < W * TW; idx += kThreads) {
        const int c = idx / W, kk = idx - c * W;
        Bs[kk * TLD + c] = (c < wj) ? Tf[static_cast<int64_t>(j0 + c) * ns + k0 + kk] : 0.0;       // X(k0 + kk, j0 + c)
      }
      __syncthreads();
      for (int kk = 0; kk < W; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = As[kk * TLD + r0 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = Bs[kk * TLD + c0 + q];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p][q] += a[p] * b[q];
      }
      __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) Bs[(r0 + p) * TLD + c0 + q] = acc[p][q];
    for (int idx = tid; idx < W * TW; idx += kThreads) {
      const int kk = idx / TW, r = idx - kk * TW;
      As[kk * TLD + r] = (kk < wi && r < wi) ? If[static_cast<int64_t>(i) * W * W + kk * W + r] : 0.0;  // inv(L_ii)(r, kk)
    }
    __syncthreads();
    double out[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) out[p][q] = 0.0;
    for (int kk = 0; kk < wi; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = As[kk * TLD + r0 + q];
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = Bs[kk * TLD + c0 + q];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[p][q] += a[p] * b[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (r0 + p < wi && c0 + q < wj) Tf[static_cast<int64_t>(j0 + c0 + q) * ns + i0 + r0 + p] = -out[p][q];
    __threadfence();  // the next block row reads these entries back through global memory
    __syncthreads();
  }
}

// M21 = L21 * T in place, one 64-row tile of the border per workgroup, 64-column blocks left to right
// (block jt of the result needs blocks kt >= jt of L21 only, so overwriting block jt is safe).
__global__ __launch_bounds__(kThreads) void m21_kernel(FrontArrays fa, const int* __restrict__ m_pref, int nfronts,
                                                      double* F, const double* __restrict__ Tb) {
  __shared__ double As[TW * TLD];
  __shared__ double Bs[TW * TLD];
  const int f = find_slot(m_pref, nfronts, blockIdx.x);
  const int bt = blockIdx.x - m_pref[f];
  const int ns = fa.ns[f];
  const int64_t d = ns + fa.bs[f];
  const int rb0 = ns + bt * TW;
  const int rows = min(TW, static_cast<int>(d) - rb0);
  const int nct = (ns + TW - 1) / TW;
  double* Ff = F + fa.foff[f];
  const double* Tf = Tb + fa.toff[f];
  const int tid = threadIdx.x;
  const int r0 = (tid % 16) * 4, c0 = (tid / 16) * 4;
  for (int jt = 0; jt < nct; ++jt) {
    const int j0 = jt * TW, wj = min(TW, ns - j0);
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int kt = jt; kt < nct; ++kt) {
      const int k0 = kt * TW, wk = min(TW, ns - k0);
      for (int idx = tid; idx < TW * TW; idx += kThreads) {
        const int kk = idx / TW, r = idx - kk * TW;
        As[kk * TLD + r] = (kk < wk && r < rows) ? Ff[static_cast<int64_t>(k0 + kk) * d + rb0 + r] : 0.0;   // L21(r, k0 + kk)
      }
      for (int idx = tid; idx < TW * TW; idx += kThreads) {
        const int c = idx / TW, kk = idx - c * TW;
        Bs[kk * TLD + c] = (kk < wk && c < wj) ? Tf[static_cast<int64_t>(j0 + c) * ns + k0 + kk] : 0.0;      // T(k0 + kk, j0 + c)
      }
      __syncthreads();
      for (int kk = 0; kk < wk; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = As[kk * TLD + r0 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = Bs[kk * TLD + c0 + q];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p][q] += a[p] * b[q];
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (r0 + p < rows && c0 + q < wj) Ff[static_cast<int64_t>(j0 + c0 + q) * d + rb0 + r0 + p] = acc[p][q];
  }
}

// ------------------------------------------------------------------ triangular sweeps
// Tile products.  A workgroup produces a 64 x KB block of outputs (KB = 4*KPT >= k) from a 64 x 64 tile As
// of [T; M21] and a 64 x KB block Bs of right-hand sides, both in LDS:
//   out[o][c] += sum_k As[k*TLD + o] * Bs[k*BLD + c]
//
//  * k <= 8  (KPT 1, 2): vector FMAs, lane (o = tid/4, cg = tid%4) owns KPT columns of row o.
//  * k > 8   (KPT 4, 8): v_mfma_f64_16x16x4_f64.  Wave w owns output rows 16w..16w+15 and all KB/16
//    column tiles; per K-step of 4 a lane feeds ONE double of A and one of B per tile, so the LDS
//    traffic per flop is ~9x lower than the vector form (which is LDS-bound at KB = 32).  Result map of
//    the f64 MFMA: acc[reg] <-> (row = (lane>>4) + 4*reg, col = lane&15) inside the 16 x 16 tile.
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
          const double a = As[k * TLD + o];
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
          const double a = As[k * TLD + o];
          const double* b = Bs + k * BLD + cg * KPT;
#pragma unroll
          for (int q = 0; q < KPT; ++q) acc[q] += a * b[q];
        }
      }
    }
  }
};

// One workgroup of a level launch.
//   forward, mode 0 : output row tile `tile` of front f, steps [s0, s1) of the chain over column tiles ct
//   forward, mode 1 : front with a single column tile: row tiles [s0, s1), the right-hand side block stays in LDS
//   backward        : output column tile `tile`, steps [s0, s1) of the chain (own row tiles, then border tiles)
// Long chains of the big fronts near the root are cut into G groups, each its own workgroup: a group stores its
// partial 64 x KB block in slab `slab + g`, and the group that arrives LAST (atomic ticket `cnt`, nobody waits)
// adds the G slabs in fixed order and finishes the tile -- bitwise reproducible, no extra launch.
struct WgRec {
  int f, tile, s0, s1;
  int slab, cnt, G, flags;  // flags: bit 0 = mode 1, bit 1 = the front has children (carries to gather), bits 8.. = group
  // the front's own numbers ride along: one scalar load instead of a second dependent round of them
  int ns, bs, c0, slot;  // slot: the carry plane this front writes into (its index among its parent's children)
  int64_t voff, foff, toff, bptr;
  int64_t pvoff;         // the parent's first row in V, -1 for a root
  int64_t scratch;       // 1: surplus child (slot >= kMaxS): its carry goes to the scratch plane, at its own border rows
  int64_t ftoff;         // the front's block in the transposed copy Ft
  int64_t ldt;           // leading dimension of T as the sweeps read it (ns)
  int64_t moff;          // fronts with several column tiles: first tile of this record's chain in the fragment-major copy
};

struct FragFront {
  int f, nst, nbt, nko;  // nko: K-steps of the last own tile (columns in Fm, rows in Bm), nkb: of the last border tile
  int nkb, pad;
  int64_t fm, bm;        // the front's blocks in Fm / Bm (doubles)
};

__device__ __host__ inline int64_t frag_fwd_steps(int nst, int nko, int rt) {  // K-steps before row tile rt
  const int64_t ks = 16 * (nst - 1) + nko;
  return (rt <= nst - 1) ? 8LL * rt * (rt + 1) : 8LL * (nst - 1) * nst + ks * (rt - (nst - 1));
}
__device__ __host__ inline int64_t frag_bwd_chain(int nst, int nbt, int nko, int nkb, int ct) {  // K-steps of chain ct
  return 16LL * (nst - 1 - ct) + nko + (nbt > 0 ? 16LL * (nbt - 1) + nkb : 0);
}
__device__ __host__ inline int64_t frag_bwd_steps(int nst, int nbt, int nko, int nkb, int ct) {  // ... before chain ct
  int64_t sum = 0;
  for (int c = 0; c < ct; ++c) sum += frag_bwd_chain(nst, nbt, nko, nkb, c);
  return sum;
}
__device__ __host__ inline int64_t frag_bwd_in_chain(int nown, int nko, int t) {  // K-steps of chain steps before t
  return (t < nown) ? 16LL * t : 16LL * (nown - 1) + nko + 16LL * (t - nown);
}

struct LevelArgs {
  const WgRec* wg;
  double* P;      // partial slabs, 64 x KBMAX each
  int* tickets;   // one per split tile, zero between sweeps
  int kb;
  int kd;         // rows of the LDS tiles this launch was given (multiple of 8, <= 64): the longest product of the level
  int nwg;        // records of this launch
  int per_xcd;    // multi-tile launches: records per XCD (grid = 8 * per_xcd, see xcd_record); 0 = blockIdx is the record
  double* V1;     // pre-assembled levels (PRE): the plane that holds v1 at the fronts' own rows (rows of KB columns)
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one; observed, used for speed only):
// give every XCD a contiguous range of the level's records.  The records of a front are neighbours, so the workgroups
// that read the same right-hand side blocks of a front (all its row tiles in the forward sweep, all its column tiles
// in the backward sweep) sit behind one L2 and fetch them from HBM once instead of once per workgroup.
__device__ __forceinline__ int xcd_record(const LevelArgs& la) {
  if (la.per_xcd == 0) return blockIdx.x;
  return (blockIdx.x & 7) * la.per_xcd + (blockIdx.x >> 3);
}

constexpr int TILE_IT = TW * TW / kThreads;  // 16 matrix elements per lane and tile

// fronts with more than kMaxS children: the carries of the surplus children (written to the scratch plane) are
// summed, in fixed order, into the extra plane their parent reads
__global__ void overflow_sum_kernel(int nrows, const int* __restrict__ ov_dst, const int* __restrict__ ov_ptr,
                                    const int* __restrict__ ov_src, int kb, int ld, double* __restrict__ extra,
                                    const double* __restrict__ scratch) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * kb) return;
  const int x = idx / kb, c = idx - x * kb;
  double sum = 0.0;
  for (int e = ov_ptr[x]; e < ov_ptr[x + 1]; ++e) sum += scratch[static_cast<int64_t>(ov_src[e]) * ld + c];
  extra[static_cast<int64_t>(ov_dst[x]) * ld + c] = sum;
}

// Joins the G partial blocks of a split tile.  Returns false for all but the last group to arrive; for that one
// acc holds the sum of the slabs 0..G-1 (fixed order) on return.  Cross-workgroup hand-off inside a launch on
// gfx950 (per-XCD L2s are not coherent): the slab is stored write-through (agent-scope relaxed atomic stores =
// sc1), every wave drains its stores, ONE lane draws the ticket (relaxed, agent scope); the last arriver does ONE
// agent-scope acquire before the workgroup reads the slabs with plain loads.
template <int KPT>
__device__ __forceinline__ bool fold_groups(const WgRec& w, const LevelArgs& la, double (&acc)[Tile<KPT>::NOUT],
                                            int* sflag) {
  using T = Tile<KPT>;
  typedef unsigned long long u64;
  double* Pp = la.P + static_cast<int64_t>(w.slab) * (TW * KBMAX);
  u64* mine = reinterpret_cast<u64*>(Pp + static_cast<int64_t>(w.flags >> 8) * (TW * KBMAX));
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t)  // lane-major slab layout: element t of lane l at t*256 + l
    __hip_atomic_store(mine + t * kThreads + threadIdx.x, static_cast<u64>(__double_as_longlong(acc[t])),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    *sflag = (__hip_atomic_fetch_add(la.tickets + w.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == w.G - 1) ? 1 : 0;
  __syncthreads();
  if (*sflag == 0) return false;
  if (threadIdx.x == 0) {
    __hip_atomic_store(la.tickets + w.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next sweep
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  for (int gg = 0; gg < w.G; gg += 2) {  // two slabs per trip, all their loads in flight; sums in slab order
    const double* p0 = Pp + static_cast<int64_t>(gg) * (TW * KBMAX) + threadIdx.x;
    const bool two = gg + 1 < w.G;
    double a[T::NOUT], b[T::NOUT];
#pragma unroll
    for (int t = 0; t < T::NOUT; ++t) {
      a[t] = p0[t * kThreads];
      b[t] = two ? p0[TW * KBMAX + t * kThreads] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < T::NOUT; ++t) {
      acc[t] += a[t];
      if (two) acc[t] += b[t];
    }
  }
  return true;
}

// Forward sweep, one level.  The carries travel through V: nslot planes of (rows of all fronts) x KBMAX; child
// number s of a front writes its carry into plane s AT THE PARENT'S ROWS (scatter through rel, no two writers per
// plane and row), so the parent reads its rows of every plane with plain contiguous loads and adds them in plane
// order -- no index hop, no atomics.  Entries no child ever writes stay zero from the allocation on.
// For front f with v1 = alpha * X[own rows] + carries on the own rows:
//   own row tile rt    : z(rt)  =  sum_{ct <= rt} T(rt, ct) v1(ct)                    -> Y = S z
//   border row tile rt : carry  =  carries on those rows - sum_ct M21(rt, ct) v1(ct)  -> the parent's rows of plane w.slot
// Matrix tiles and vector blocks are register staged one step ahead (the gather indices two), so the loads of
// step t+1 are in flight while step t multiplies.  Masked lanes load from a zero word (address select): a branch
// around the load would serialise the loads of a tile.
// NSL: carry planes compiled in (2: binary trees, else kMaxS + 1); FRAG (fronts with several column tiles, MFMA widths):
// matrix operands straight from the fragment-major copy instead of staged through LDS
// PRE (FRAG only; levels with thousands of workgroups, see v1_assemble_kernel): v1 has been written to the plane la.V1 at
// the fronts' own rows by a launch of its own -- every row-tile workgroup of a front reads ONE contiguous 16 KB block per
// step instead of an index round, gathered rows of the caller's block and NSL planes
template <int KPT, bool SINGLE, int NSL, bool FRAG = false, bool PRE = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(((SINGLE && KPT >= 4) || (FRAG && KPT >= 4)) ? ((FRAG && KPT >= 8) ? kFragWavesFwd : 3) : 1)))
void fwd_level_kernel(FrontArrays fa, LevelArgs la, const double* __restrict__ F,
                                                            const double* __restrict__ Tb, const double* X, int ldx,
                                                            double alpha, double* V, double* __restrict__ Y) {
  using T = Tile<KPT>;
  constexpr int IT = KPT;  // vector elements per lane: (64 x KB) / 256
  extern __shared__ double lds_tiles[];  // la.kd (+ 1 spare) rows of the matrix tile, as many of the vector block
  double* const As = lds_tiles;
  const int kdl = SINGLE ? la.kd : TW;  // (a constant where the fronts have several column tiles: folded LDS offsets)
  double* const Bs = lds_tiles + (((SINGLE && KPT == 4) || FRAG) ? 0 : (kdl + 1) * TLD);  // (the direct-fragment paths have no matrix tile)
  const int rec = SINGLE ? static_cast<int>(blockIdx.x) : xcd_record(la);
  if (!SINGLE && rec >= la.nwg) return;
  const WgRec w = la.wg[rec];
  const int kb = la.kb;
  const int ns = w.ns;
  const int d = ns + w.bs;
  const int nst = (ns + TW - 1) / TW;
  const bool kids = (w.flags & 2) != 0;
  const int64_t vbase = w.voff;
  const double* Tf = Tb + w.toff;
  const double* Ff = F + w.foff;
  const int ar = threadIdx.x & (TW - 1), ajb = threadIdx.x >> 6;  // lane -> (row, first column) of a matrix tile

  int xi[IT];
  double bv[IT], av[TILE_IT];
  const int nslot = kids ? fa.nslot : 0;
  const int64_t vslot = fa.vrows * T::KB;  // plane size: rows of all fronts x KB (the planes of this sweep width)
  double* Vout = V + static_cast<int64_t>(w.slot) * vslot;

  static_assert(!PRE || FRAG, "pre-assembled right-hand sides: the fragment path only");
  auto fetch_idx = [&](int ct) {
    if constexpr (PRE) return;
    const int wd = min(TW, ns - ct * TW);
#pragma unroll
    for (int e = 0; e < IT; ++e) {
      const int r = (threadIdx.x + e * kThreads) / T::KB;
      const int64_t vrow = vbase + ct * TW + r;
      const bool ok = r < wd;
      xi[e] = *(ok ? fa.v_src + vrow : fa.neg1);
    }
  };
  auto fetch_b = [&](int ct) {
    if constexpr (PRE) {
      const int wd = min(TW, ns - ct * TW);
      const double* V1 = la.V1 + (vbase + ct * TW) * T::KB;
#pragma unroll
      for (int e = 0; e < IT; ++e) {
        const int idx = threadIdx.x + e * kThreads;
        bv[e] = *(((idx & (T::KB - 1)) < kb && idx / T::KB < wd) ? V1 + idx : fa.zero);
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < IT; ++e) {
      const int idx = threadIdx.x + e * kThreads;
      const int c = idx & (T::KB - 1);
      const bool ok = c < kb && xi[e] >= 0;
      const double xv = alpha * *(ok ? X + static_cast<int64_t>(xi[e]) * ldx + c : fa.zero);
      const double* cp = V + (vbase + ct * TW + idx / T::KB) * T::KB + c;
      double v = 0.0;  // the children's carries first, in plane order, then the right-hand side: alpha x + (p0 + p1 + ..)
#pragma unroll         // -- the association every forward kernel (and the in-LDS sums of the subtree kernels) uses
      for (int s = 0; s < NSL; ++s) v += *((ok && s < nslot) ? cp + s * vslot : fa.zero);
      bv[e] = xv + v;
    }
  };
  // tile (row tile rt, column tile ct) of [T; M21]: element (r, j) at base + j*ld + r
  auto fetch_a = [&](int rt, int ct, double (&av)[TILE_IT]) {
    const bool own = rt < nst;
    const int row0 = own ? rt * TW : ns + (rt - nst) * TW;
    const int rows = min(TW, (own ? ns : d) - row0);
    const int64_t ld = own ? w.ldt : d;
    const int wd = min(TW, ns - ct * TW);
    const double* Ap = (own ? Tf : Ff) + static_cast<int64_t>(ct) * TW * ld + row0 + ar;
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
      const int j = ajb + it * (kThreads / TW);
      av[it] = *((j < wd && ar < rows) ? Ap + static_cast<int64_t>(j) * ld : fa.zero);
    }
  };
  auto commit_a = [&](const double (&av)[TILE_IT]) {
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it)  // As[k][o] = R(row0 + o, ct*64 + k); rows past the allocation -> spare row kdl
      As[min(ajb + it * (kThreads / TW), kdl) * TLD + ar] = av[it];
  };
  auto commit_b = [&]() {
#pragma unroll
    for (int e = 0; e < IT; ++e) {
      const int idx = threadIdx.x + e * kThreads;
      Bs[min(idx / T::KB, kdl) * T::BLD + (idx & (T::KB - 1))] = bv[e];
    }
  };
  // carries on the rows of border tile rt and where its results go (rows of the parent), in the lane's output layout
  auto fetch_carry = [&](int rt, double (&cg)[T::NOUT], int (&di)[T::NOUT]) {
    const int row0 = ns + (rt - nst) * TW;
    const int rows = min(TW, d - row0);
#pragma unroll
    for (int t = 0; t < T::NOUT; ++t) {
      int o, c;
      T::coords(t, o, c);
      const bool ok = o < rows && c < kb;
      di[t] = *(ok ? fa.rel + w.bptr + (row0 - ns) + o : fa.neg1);
      const double* cp = V + (vbase + row0 + o) * T::KB + c;
      double v = 0.0;
#pragma unroll
      for (int s = 0; s < NSL; ++s) v += *((ok && s < nslot) ? cp + s * vslot : fa.zero);
      cg[t] = v;
    }
  };
  auto store_tile = [&](int rt, const double (&acc)[T::NOUT], const double (&cg)[T::NOUT], const int (&di)[T::NOUT]) {
    const bool own = rt < nst;
    const int row0 = own ? rt * TW : ns + (rt - nst) * TW;
    const int rows = min(TW, (own ? ns : d) - row0);
    // values first, ONE explicit wait, then the stores (a load inside each masked store block would make every
    // store wait for the one before it)
    double ov[T::NOUT];
    if (own) {
      const double* sgp = fa.sgn + w.c0 + row0;
      double* Yf = Y + (vbase + row0) * kb;
#pragma unroll
      for (int t = 0; t < T::NOUT; ++t) {
        int o, c;
        T::coords(t, o, c);
        ov[t] = *((o < rows && c < kb) ? sgp + o : fa.zero) * acc[t];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
      for (int t = 0; t < T::NOUT; ++t) {
        int o, c;
        T::coords(t, o, c);
        if (o < rows && c < kb) Yf[static_cast<int64_t>(o) * kb + c] = ov[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < T::NOUT; ++t) ov[t] = cg[t] - acc[t];
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
      for (int t = 0; t < T::NOUT; ++t) {
        int o, c;
        T::coords(t, o, c);
        const int64_t drow = (w.scratch != 0) ? vbase + row0 + o : w.pvoff + di[t];
        if (di[t] >= 0) Vout[drow * T::KB + c] = ov[t];
      }
    }
  };

  double acc[T::NOUT], cg[T::NOUT];
  int di[T::NOUT];
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    acc[t] = cg[t] = 0.0;
    di[t] = -1;
  }

  if constexpr (SINGLE && KPT == 4) {  // (at KPT = 8 fragments + carries exceed the register budget of 3 waves per SIMD)
    // One column tile, MFMA path: v1 goes to LDS once; after that barrier the four waves run on their own.  A wave
    // owns output rows 16w..16w+15 of every row tile and loads exactly the matrix fragments its MFMAs consume,
    // straight from global memory into the operand layout (lane (i, k) holds R(row0 + 16w + i, 4*kk + k)): no matrix
    // tile in LDS, no barrier per tile, a third of the LDS footprint -- more workgroups per CU on the levels that
    // hold most of the factor.  The next tile's fragments are requested as soon as the MFMAs of this one are issued.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int orow = 16 * wave + li;
    auto fetch_frag = [&](int rt, double (&a)[16]) {
      const bool own = rt < nst;
      const int row0 = own ? rt * TW : ns + (rt - nst) * TW;
      const int rows = min(TW, (own ? ns : d) - row0);
      const int ld = own ? static_cast<int>(w.ldt) : d;
      const double* Ab = (own ? Tf : Ff) + row0;  // uniform base, 32-bit lane offsets (a tile spans < 2^31 doubles)
      const ptrdiff_t zoff = fa.zero - Ab;        // masked lanes read the zero word
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int k = 4 * kk + lk;
        a[kk] = (k < ns && orow < rows) ? Ab[static_cast<unsigned>(k * ld + orow)] : Ab[zoff];
      }
    };
    auto stepd = [&](int rt, double (&cur)[16], int nxt) {
      // 32 columns: the carries are requested after the products (registers: 3 waves per SIMD without spills)
      if (KPT < 8 && rt >= nst) fetch_carry(rt, cg, di);
      double4_t c[T::NT];
#pragma unroll
      for (int n = 0; n < T::NT; ++n) c[n] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int k = 4 * kk + lk;
#pragma unroll
        for (int n = 0; n < T::NT; ++n)
          c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[kk], Bs[k * T::BLD + 16 * n + li], c[n], 0, 0, 0);
      }
      if (nxt < w.s1) fetch_frag(nxt, cur);
#pragma unroll
      for (int n = 0; n < T::NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[4 * n + r] = c[n][r];
      if (KPT >= 8 && rt >= nst) fetch_carry(rt, cg, di);
      store_tile(rt, acc, cg, di);
    };
    double a0[16];
    fetch_idx(0);
    fetch_frag(w.s0, a0);
    // v1 -> LDS in two halves (register budget): X row + carries of half the lane's elements at a time
    constexpr int HB = (IT > 4) ? IT / 2 : IT;
#pragma unroll
    for (int h = 0; h < IT; h += HB) {
      double vh[HB];
#pragma unroll
      for (int e = h; e < h + HB; ++e) {
        const int idx = threadIdx.x + e * kThreads;
        const int c = idx & (T::KB - 1);
        const bool ok = c < kb && xi[e] >= 0;
        const double xv = alpha * *(ok ? X + static_cast<int64_t>(xi[e]) * ldx + c : fa.zero);
        const double* cp = V + (vbase + idx / T::KB) * T::KB + c;
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < NSL; ++s) v += *((ok && s < nslot) ? cp + s * vslot : fa.zero);
        vh[e - h] = xv + v;
      }
#pragma unroll
      for (int e = h; e < h + HB; ++e) {
        const int idx = threadIdx.x + e * kThreads;
        Bs[min(idx / T::KB, kdl) * T::BLD + (idx & (T::KB - 1))] = vh[e - h];
      }
      asm volatile("" ::: "memory");  // keep the halves apart: the second half's loads are not hoisted above these stores
    }
    __syncthreads();
    for (int rt = w.s0; rt < w.s1; ++rt) stepd(rt, a0, rt + 1);
  } else if constexpr (SINGLE) {  // one column tile: v1 is loaded once, the row tiles [s0, s1) are walked with it
    auto step = [&](int rt, double (&cur)[TILE_IT], int nxt) {
      commit_a(cur);
      __syncthreads();
      if (nxt < w.s1) fetch_a(nxt, 0, cur);
      if (rt >= nst) fetch_carry(rt, cg, di);  // (requested a tile ahead they would cost the third workgroup per CU)
#pragma unroll
      for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
      T::mac(As, Bs, ns, acc);
      store_tile(rt, acc, cg, di);
      __syncthreads();
    };
    fetch_idx(0);
    fetch_b(0);
    fetch_a(w.s0, 0, av);
    commit_b();
    for (int rt = w.s0; rt < w.s1; ++rt) step(rt, av, rt + 1);
  } else if constexpr (FRAG) {
    static_assert(!FRAG || T::kMfma, "fragments feed MFMAs");
    // Several column tiles, MFMA path: the chain over the column tiles with the matrix operands loaded from global
    // memory straight into the MFMA layout: wave v owns output rows 16v..16v+15 and reads exactly the fragments its
    // MFMAs consume, one tile ahead, from the fragment-major copy of the front (pack_frag_kernel: a tile is one
    // contiguous 32 KB block in consumption order).  Only the vector block goes through LDS -- a third of the
    // footprint of the staged form, no matrix commit between the barriers, no lane masks.  Same products in the same
    // order as the staged form (K ascending in steps of 4): bitwise the same result.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int rt = w.tile;
    const bool own = rt < nst;
    const int row0 = own ? rt * TW : ns + (rt - nst) * TW;
    const int rows = min(TW, (own ? ns : d) - row0);
    const bool wok = 16 * wave < rows;  // (wave-uniform) the wave's 16 rows meet the front
    // F = the fragment-major copy Fm here: tile ct of this row of tiles at moff + ct * 4096 doubles, K-step kk of the
    // workgroup 2 KB further on each time, no masks (cells outside the front hold zeros)
    const double* Am = F + w.moff + wave * 64 + lane;
    auto fetch_frag = [&](int ct, double (&a)[16]) {
      const int nk = (min(TW, ns - ct * TW) + 3) >> 2;
      const double* Ac = Am + static_cast<int64_t>(ct) * (TW * TW);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a[kk] = (wok && kk < nk) ? Ac[kk * 256] : 0.0;
    };
    double4_t c[T::NT];
#pragma unroll
    for (int n = 0; n < T::NT; ++n) c[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    auto stepf = [&](int ct, double (&cur)[16], double (&nxt)[16]) {
      commit_b();
      __syncthreads();
      if (ct + 1 < w.s1) {
        fetch_b(ct + 1);
        fetch_frag(ct + 1, nxt);
        if (ct + 2 < w.s1) fetch_idx(ct + 2);
      }
      const int nk = (min(TW, ns - ct * TW) + 3) >> 2;  // K-steps that meet columns of the front (uniform)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        if (kk < nk) {
          const int k = 4 * kk + lk;
#pragma unroll
          for (int n = 0; n < T::NT; ++n)
            c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[kk], Bs[k * T::BLD + 16 * n + li], c[n], 0, 0, 0);
        }
      }
      __syncthreads();
    };
    double a0[16], a1[16];
    fetch_frag(w.s0, a0);  // (no index round in front of the matrix operands)
    fetch_idx(w.s0);
    fetch_b(w.s0);
    if (w.s0 + 1 < w.s1) fetch_idx(w.s0 + 1);
    for (int ct = w.s0; ct < w.s1; ct += 2) {
      stepf(ct, a0, a1);
      if (ct + 1 < w.s1) stepf(ct + 1, a1, a0);
    }
#pragma unroll
    for (int n = 0; n < T::NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[4 * n + r] = c[n][r];
    if (rt >= nst) fetch_carry(rt, cg, di);  // in flight while the groups of a split chain are joined
    if (w.G > 1 && !fold_groups<KPT>(w, la, acc, reinterpret_cast<int*>(Bs))) return;
    store_tile(rt, acc, cg, di);
  } else {
  const int rt = w.tile;
  fetch_idx(w.s0);
  fetch_b(w.s0);
  fetch_a(rt, w.s0, av);
  if (w.s0 + 1 < w.s1) fetch_idx(w.s0 + 1);
  for (int ct = w.s0; ct < w.s1; ++ct) {
    commit_a(av);
    commit_b();
    __syncthreads();
    if (ct + 1 < w.s1) {
      fetch_b(ct + 1);
      fetch_a(rt, ct + 1, av);
      if (ct + 2 < w.s1) fetch_idx(ct + 2);
    }
    T::mac(As, Bs, min(TW, ns - ct * TW), acc);
    __syncthreads();
  }
  if (rt >= nst) fetch_carry(rt, cg, di);  // in flight while the groups of a split chain are joined
  if (w.G > 1 && !fold_groups<KPT>(w, la, acc, reinterpret_cast<int*>(As))) return;
  store_tile(rt, acc, cg, di);
  }
}

// v1 of one level, written once: workgroup (front, column tile ct) forms alpha X[own rows] + (p0 + p1 + ..) for the 64 rows
// of the tile -- the expression of fetch_b above, entry by entry: bitwise what the row-tile workgroups would have formed
// themselves -- and leaves it in the plane V1 at the front's own rows.
template <int KPT, int NSL>
__global__ __launch_bounds__(kThreads) void v1_assemble_kernel(FrontArrays fa, const WgRec* __restrict__ recs, const double* X,
                                                              int ldx, double alpha, const double* V, double* V1p, int kb) {
  using T = Tile<KPT>;
  constexpr int IT = KPT;
  const WgRec w = recs[blockIdx.x];
  const int ct = w.tile;
  const int wd = min(TW, w.ns - ct * TW);
  const int64_t vrow0 = w.voff + ct * TW;
  const int nslot = ((w.flags & 2) != 0) ? fa.nslot : 0;
  const int64_t vslot = fa.vrows * T::KB;
  int xi[IT];
#pragma unroll
  for (int e = 0; e < IT; ++e) {
    const int r = (threadIdx.x + e * kThreads) / T::KB;
    xi[e] = *((r < wd) ? fa.v_src + vrow0 + r : fa.neg1);
  }
  double bv[IT];
#pragma unroll
  for (int e = 0; e < IT; ++e) {
    const int idx = threadIdx.x + e * kThreads;
    const int c = idx & (T::KB - 1);
    const bool ok = c < kb && xi[e] >= 0;
    const double xv = alpha * *(ok ? X + static_cast<int64_t>(xi[e]) * ldx + c : fa.zero);
    const double* cp = V + vrow0 * T::KB + idx;
    double v = 0.0;
#pragma unroll
    for (int s = 0; s < NSL; ++s) v += *((ok && s < nslot) ? cp + s * vslot : fa.zero);
    bv[e] = xv + v;
  }
  double* V1 = V1p + vrow0 * T::KB;
#pragma unroll
  for (int e = 0; e < IT; ++e) {
    const int idx = threadIdx.x + e * kThreads;
    if ((idx & (T::KB - 1)) < kb && xi[e] >= 0) V1[idx] = bv[e];
  }
}

// Backward sweep, one level: workgroup (front f, column tile ct) forms
//   x1(ct) = sum_{rt >= ct} T(rt, ct)^T y(rt)  -  sum_bt M21(bt, ct)^T x_border(bt)
// y = S z from the forward sweep (Y); x_border are rows of the caller's block Out that the ancestors' launches
// have already written (bout = their row numbers).  The solution goes straight to Out.
// SINGLE: fronts with one column tile, LDS tiles of la.kd rows; FRAG: as in the forward kernel (Ft = the copy Bm then)
// LIDX (FRAG, 32 columns): the front's border rows in the caller's block (bout) are read ONCE into LDS behind the vector
// block instead of eight index registers per lane requested two chain steps ahead: 170 -> under 168 registers, the third
// wave per SIMD without spills (levels whose longest border fits kLidxMax entries: 25 + 28 KB of LDS at most, three workgroups per CU)
constexpr int kLidxMax = 7168;
template <int KPT, bool SINGLE, bool FRAG = false, bool LIDX = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(FRAG ? (KPT >= 8 ? (LIDX ? 3 : kFragWavesBwd32) : kFragWaves) : 1)))
void bwd_level_kernel(FrontArrays fa, LevelArgs la, const double* __restrict__ F,
                                                            const double* __restrict__ Tb, const double* __restrict__ Ft,
                                                            const double* __restrict__ Y, double* Out, int ldo) {
  using T = Tile<KPT>;
  constexpr int IT = KPT;
  extern __shared__ double lds_tiles[];
  double* const As = lds_tiles;
  const int kdl = SINGLE ? la.kd : TW;
  double* const Bs = lds_tiles + (FRAG ? 0 : (kdl + 1) * TLD);  // (direct fragments: no matrix tile)
  const int rec = SINGLE ? static_cast<int>(blockIdx.x) : xcd_record(la);
  if (!SINGLE && rec >= la.nwg) return;
  const WgRec w = la.wg[rec];
  const int ct = w.tile;
  const int kb = la.kb;
  const int ns = w.ns, bs = w.bs;
  const int d = ns + bs;
  const int nst = (ns + TW - 1) / TW;
  const int c0t = ct * TW;
  const int wc = min(TW, ns - c0t);
  const int nown = nst - ct;
  const int64_t vbase = w.voff;
  const double* Tp = Tb + w.toff + static_cast<int64_t>(c0t) * w.ldt;  // T(r, c0t + o) at o*ldt + r
  const double* Mp = F + w.foff + static_cast<int64_t>(c0t) * d;     // M21(r, c0t + o) at o*d + r
  const int* __restrict__ bout = fa.bout + w.bptr;
  const int ar = threadIdx.x & (TW - 1), ajb = threadIdx.x >> 6;

  int ri[IT];
  double bv[IT], av[TILE_IT];
  int pend_rows = 0;

  auto fetch_idx = [&](int t) {  // border tiles only: rows of Out
    if (t < nown) return;
    const int b0 = (t - nown) * TW;
    const int rows = min(TW, bs - b0);
#pragma unroll
    for (int e = 0; e < IT; ++e) {
      const int r = (threadIdx.x + e * kThreads) / T::KB;
      ri[e] = *((r < rows) ? bout + b0 + r : fa.neg1);
    }
  };
  auto fetch_val = [&](int t) {
    const bool border = t >= nown;
    const int rbase = border ? ns + (t - nown) * TW : (ct + t) * TW;
    const int rows = min(TW, (border ? d : ns) - rbase);
    if (!border) {
      const double* Yp = Y + (vbase + rbase) * kb;
#pragma unroll
      for (int e = 0; e < IT; ++e) {
        const int idx = threadIdx.x + e * kThreads;
        const int r = idx / T::KB, c = idx & (T::KB - 1);
        bv[e] = *((r < rows && c < kb) ? Yp + static_cast<int64_t>(r) * kb + c : fa.zero);
      }
    } else {
#pragma unroll
      for (int e = 0; e < IT; ++e) {
        const int c = (threadIdx.x + e * kThreads) & (T::KB - 1);
        bv[e] = -*((ri[e] >= 0 && c < kb) ? Out + static_cast<int64_t>(ri[e]) * ldo + c : fa.zero);
      }
    }
    const double* Ap = (border ? Mp : Tp) + rbase + ar;
    const int64_t ld = border ? d : w.ldt;
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
      const int j = ajb + it * (kThreads / TW);
      av[it] = *((j < wc && ar < rows) ? Ap + static_cast<int64_t>(j) * ld : fa.zero);
    }
    pend_rows = rows;
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it)  // As[k][o] = R(rbase + k, c0t + o); rows past the allocation -> spare row kdl
      As[min(ar, kdl) * TLD + ajb + it * (kThreads / TW)] = av[it];
#pragma unroll
    for (int e = 0; e < IT; ++e) {
      const int idx = threadIdx.x + e * kThreads;
      Bs[min(idx / T::KB, kdl) * T::BLD + (idx & (T::KB - 1))] = bv[e];
    }
  };

  double acc[T::NOUT];
  int oi[T::NOUT];  // rows of the caller's block the results go to
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) acc[t] = 0.0;
  if constexpr (FRAG) {
    static_assert(!FRAG || (!SINGLE && T::kMfma), "fragments: several column tiles, MFMA widths");
    // Direct fragments (see the forward kernel): wave v owns the output rows (= own columns of the front) 16v..16v+15
    // and reads its MFMA operands from the fragment-major copy Bm, one tile ahead; only the vector block goes
    // through LDS.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const bool wok = 16 * wave < wc;  // (wave-uniform) the wave's 16 columns meet the front
    // Ft = the fragment-major copy Bm here: the tiles of this column tile's chain one after the other from moff
    const int nko = (ns - TW * (nst - 1) + 3) >> 2;  // K-steps of the last own tile
    const double* Am = Ft + w.moff + wave * 64 + lane;
    int frag_rows = 0;
    static_assert(!LIDX || FRAG, "index list in LDS: the fragment path only");
    int* const Is = reinterpret_cast<int*>(Bs + (TW + 1) * T::BLD);  // LIDX: bout of this front, behind the vector block
    auto fetch_vec = [&](int t) {
      const bool border = t >= nown;
      const int rbase = border ? ns + (t - nown) * TW : (ct + t) * TW;
      const int rows = min(TW, (border ? d : ns) - rbase);
      if (!border) {
        const double* Yp = Y + (vbase + rbase) * kb;
#pragma unroll
        for (int e = 0; e < IT; ++e) {
          const int idx = threadIdx.x + e * kThreads;
          const int r = idx / T::KB, c = idx & (T::KB - 1);
          bv[e] = *((r < rows && c < kb) ? Yp + static_cast<int64_t>(r) * kb + c : fa.zero);
        }
      } else if constexpr (LIDX) {
        const int b0 = (t - nown) * TW;
        int rr[IT];
#pragma unroll
        for (int e = 0; e < IT; ++e) {
          const int r = (threadIdx.x + e * kThreads) / T::KB;
          rr[e] = (r < rows) ? Is[b0 + r] : -1;
        }
#pragma unroll
        for (int e = 0; e < IT; ++e) {
          const int c = (threadIdx.x + e * kThreads) & (T::KB - 1);
          bv[e] = -*((rr[e] >= 0 && c < kb) ? Out + static_cast<int64_t>(rr[e]) * ldo + c : fa.zero);
        }
      } else {
#pragma unroll
        for (int e = 0; e < IT; ++e) {
          const int c = (threadIdx.x + e * kThreads) & (T::KB - 1);
          bv[e] = -*((ri[e] >= 0 && c < kb) ? Out + static_cast<int64_t>(ri[e]) * ldo + c : fa.zero);
        }
      }
    };
    auto fetch_frag = [&](int t, double (&a)[16]) {
      const bool border = t >= nown;
      const int rbase = border ? ns + (t - nown) * TW : (ct + t) * TW;
      const int rows = min(TW, (border ? d : ns) - rbase);
      const int nk = (rows + 3) >> 2;
      const double* Ac = Am + 256 * frag_bwd_in_chain(nown, nko, t);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a[kk] = (wok && kk < nk) ? Ac[kk * 256] : 0.0;
      frag_rows = rows;
    };
    auto commit_vec = [&]() {
#pragma unroll
      for (int e = 0; e < IT; ++e) {
        const int idx = threadIdx.x + e * kThreads;
        Bs[(idx / T::KB) * T::BLD + (idx & (T::KB - 1))] = bv[e];
      }
    };
    double4_t c[T::NT];
#pragma unroll
    for (int n = 0; n < T::NT; ++n) c[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    auto stepb = [&](int t, double (&cur)[16], double (&nxt)[16]) {
      const int nk = (frag_rows + 3) >> 2;  // (of the tile in `cur`)
      commit_vec();
      __syncthreads();
      if (t + 1 < w.s1) {
        fetch_vec(t + 1);
        fetch_frag(t + 1, nxt);
        if (!LIDX && t + 2 < w.s1) fetch_idx(t + 2);
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        if (kk < nk) {
          const int k = 4 * kk + lk;
#pragma unroll
          for (int n = 0; n < T::NT; ++n)
            c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[kk], Bs[k * T::BLD + 16 * n + li], c[n], 0, 0, 0);
        }
      }
      __syncthreads();
    };
    double a0[16], a1[16];
    fetch_frag(w.s0, a0);
    if constexpr (LIDX) {
      // the front's border rows: requested with the first fragments; in LDS before the first border tile is fetched (a
      // chain that starts with its own tiles publishes them with the barrier of its first step)
      constexpr int NI = 8;          // entries per lane and round: 2048 per workgroup, their loads in flight together
      int ib[NI];
      const bool any = w.s1 > nown;  // (uniform) the chain meets border tiles
#pragma unroll
      for (int q = 0; q < NI; ++q) ib[q] = *((any && threadIdx.x + q * kThreads < bs) ? bout + threadIdx.x + q * kThreads : fa.neg1);
      if (w.s0 < nown) fetch_vec(w.s0);
#pragma unroll
      for (int q = 0; q < NI; ++q)
        if (any && threadIdx.x + q * kThreads < bs) Is[threadIdx.x + q * kThreads] = ib[q];
      if (any)
        for (int i = NI * kThreads + threadIdx.x; i < bs; i += kThreads) Is[i] = bout[i];  // (borders beyond 2048 rows)
      if (w.s0 >= nown) {
        __syncthreads();
        fetch_vec(w.s0);
      }
    } else {
      fetch_idx(w.s0);
      fetch_vec(w.s0);
      if (w.s0 + 1 < w.s1) fetch_idx(w.s0 + 1);
    }
    for (int t = w.s0; t < w.s1; t += 2) {
      stepb(t, a0, a1);
      if (t + 1 < w.s1) stepb(t + 1, a1, a0);
    }
#pragma unroll
    for (int n = 0; n < T::NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[4 * n + r] = c[n][r];
  } else {
  fetch_idx(w.s0);
  fetch_val(w.s0);
  if (w.s0 + 1 < w.s1) fetch_idx(w.s0 + 1);
  for (int t = w.s0; t < w.s1; ++t) {
    const int kdim = pend_rows;
    commit();
    __syncthreads();
    if (t + 1 < w.s1) {
      fetch_val(t + 1);
      if (t + 2 < w.s1) fetch_idx(t + 2);
    }
    T::mac(As, Bs, kdim, acc);
    __syncthreads();
  }
  }
  // destination rows first (in flight while the groups of a split chain are joined), ONE explicit wait, then the
  // stores: an index load inside each masked store block would make every store wait for the one before it
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    oi[t] = *((o < wc && c < kb) ? fa.v_src + vbase + c0t + o : fa.neg1);
  }
  if (w.G > 1 && !fold_groups<KPT>(w, la, acc, reinterpret_cast<int*>(lds_tiles))) return;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
  for (int t = 0; t < T::NOUT; ++t) {
    int o, c;
    T::coords(t, o, c);
    if (oi[t] >= 0) Out[static_cast<int64_t>(oi[t]) * ldo + c] = acc[t];
  }
}

// ------------------------------------------------------------------ narrow sweeps (k <= 8) of single-tile fronts
// The bottom levels of the tree hold most of the factor in fronts with at most 64 own columns.  For a handful of
// right-hand sides the tile kernels above are bound by LDS traffic and barriers, so these fronts get a leaner form:
// ONE WAVE per 64-row tile, no LDS, no barrier.  Lane r of the wave owns output row r; it reads its matrix row
// element by element (the wave reads whole 512-byte column segments), the right-hand side block lives one row per
// lane in registers and is broadcast with v_readlane.  The backward sweep runs the same scheme on a transposed copy
// of [T; M21] (Ft, row-major d x ns per front, made once after the factorisation).
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <int KB, int NSL, int WPF>  // WPF: waves per front (each loads the front's right-hand side block)
__global__ __launch_bounds__(64 * WPF) void fwd_wave_kernel(FrontArrays fa, const WgRec* __restrict__ recs,
                                                       const double* __restrict__ F, const double* __restrict__ Tb,
                                                       const double* X, int ldx, double alpha, double* V,
                                                       double* __restrict__ Y, int kb) {
  const WgRec w = recs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ns = w.ns, d = ns + w.bs;
  const bool kids = (w.flags & 2) != 0;
  const int nslot = kids ? fa.nslot : 0;
  const int64_t vslot = fa.vrows * KB, vbase = w.voff;
  // v1, one row per lane: alpha * X[own row] + the carries on that row
  double b[KB];
  {
    const bool ok = lane < ns;
    const int xi = *(ok ? fa.v_src + vbase + lane : fa.neg1);
    const double* cp = V + (vbase + lane) * KB;
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const bool okc = ok && c < kb;
      const double xv = alpha * *(okc ? X + static_cast<int64_t>(xi) * ldx + c : fa.zero);
      double v = 0.0;
#pragma unroll
      for (int s = 0; s < NSL; ++s) v += *((okc && s < nslot) ? cp + s * vslot + c : fa.zero);
      b[c] = xv + v;
    }
  }
  double* Vout = V + static_cast<int64_t>(w.slot) * vslot;
  for (int rt = w.s0 + wave; rt < w.s1; rt += WPF) {
    const bool own = rt < 1;
    const int row0 = own ? 0 : ns + (rt - 1) * TW;
    const int rows = min(TW, (own ? ns : d) - row0);
    const int ld = own ? static_cast<int>(w.ldt) : d;
    const double* Ab = (own ? Tb + w.toff : F + w.foff) + row0;
    const bool rok = lane < rows;
    // carries on the tile's rows and where the results go (border tiles)
    double cg[KB];
    int di = -1;
    if (!own) {
      di = *(rok ? fa.rel + w.bptr + (row0 - ns) + lane : fa.neg1);
      const double* cp = V + (vbase + row0 + lane) * KB;
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < NSL; ++s) v += *((rok && c < kb && s < nslot) ? cp + s * vslot + c : fa.zero);
        cg[c] = v;
      }
    }
    double acc[KB];
#pragma unroll
    for (int c = 0; c < KB; ++c) acc[c] = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < TW; k0 += 16) {
      if (k0 < ns) {  // uniform
        double a[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const int k = k0 + kk;
          a[kk] = *((rok && k < ns) ? Ab + static_cast<int64_t>(k) * ld + lane : fa.zero);
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
          for (int c = 0; c < KB; ++c) acc[c] += a[kk] * readlane_f64(b[c], k0 + kk);
      }
    }
    if (own) {
      const double sg = *(rok ? fa.sgn + w.c0 + lane : fa.zero);
      double* yp = Y + (vbase + lane) * kb;
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one wait, then the stores back to back
#pragma unroll
      for (int c = 0; c < KB; ++c)
        if (rok && c < kb) yp[c] = sg * acc[c];
    } else {
      const int64_t drow = (w.scratch != 0) ? vbase + row0 + lane : w.pvoff + di;
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int c = 0; c < KB; ++c)
        if (di >= 0 && c < kb) Vout[drow * KB + c] = cg[c] - acc[c];
    }
  }
}

// one wave per front (single column tile): x1 = T^T y - M21^T x_border with the transposed copy Ft
template <int KB>
__global__ __launch_bounds__(64) void bwd_wave_kernel(FrontArrays fa, const WgRec* __restrict__ recs,
                                                      const double* __restrict__ Ft, const double* __restrict__ Y,
                                                      double* Out, int ldo, int kb) {
  const WgRec w = recs[blockIdx.x];
  const int lane = threadIdx.x;
  const int ns = w.ns, bs = w.bs, d = ns + bs;
  const int64_t vbase = w.voff;
  const int ntile = w.s1;  // 1 + border tiles (0 for a root)
  const int* __restrict__ bout = fa.bout + w.bptr;
  const double* Fp = Ft + w.ftoff;  // row r of [T; M21] at r * ns
  const bool ook = lane < ns;
  auto load_b = [&](int t, int ri, double (&b)[KB]) {
    if (t == 0) {
      const double* yp = Y + (vbase + lane) * kb;
#pragma unroll
      for (int c = 0; c < KB; ++c) b[c] = *((lane < ns && c < kb) ? yp + c : fa.zero);
    } else {
#pragma unroll
      for (int c = 0; c < KB; ++c) b[c] = -*((ri >= 0 && c < kb) ? Out + static_cast<int64_t>(ri) * ldo + c : fa.zero);
    }
  };
  auto load_idx = [&](int t) {  // border tile t >= 1: rows of the caller's block
    const int r = (t - 1) * TW + lane;
    return *((t >= 1 && t < ntile && r < bs) ? bout + r : fa.neg1);
  };
  double acc[KB], bc[KB], bn[KB];
#pragma unroll
  for (int c = 0; c < KB; ++c) acc[c] = bn[c] = 0.0;
  load_b(0, -1, bc);
  int ri = load_idx(1);
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) load_b(t + 1, ri, bn);
    ri = load_idx(t + 2);
    const int rbase = (t == 0) ? 0 : ns + (t - 1) * TW;
    const int rows = min(TW, ((t == 0) ? ns : d) - rbase);
    const double* Ap = Fp + static_cast<int64_t>(rbase) * ns + lane;
#pragma unroll
    for (int k0 = 0; k0 < TW; k0 += 16) {
      if (k0 < rows) {  // uniform
        double a[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const int k = k0 + kk;
          a[kk] = *((ook && k < rows) ? Ap + static_cast<int64_t>(k) * ns : fa.zero);
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
          for (int c = 0; c < KB; ++c) acc[c] += a[kk] * readlane_f64(bc[c], k0 + kk);
      }
    }
#pragma unroll
    for (int c = 0; c < KB; ++c) bc[c] = bn[c];
  }
  const int oi = *(ook ? fa.v_src + vbase + lane : fa.neg1);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one wait, then the stores back to back
  if (oi >= 0) {
    double* op = Out + static_cast<int64_t>(oi) * ldo;
#pragma unroll
    for (int c = 0; c < KB; ++c)
      if (c < kb) op[c] = acc[c];
  }
}


// Raw buffer access (stride 0, byte offsets): an offset at or past num_records reads 0.0 / drops the store -- the
// masked lanes of the wave-per-front kernels need neither a second (64-bit) address nor a load from the zero word.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
constexpr unsigned kBufOob = 0xFFFFFFF8u;   // with num_records <= kBufMax every access at this offset is out of range
constexpr int64_t kBufMax = 0xFFFFFFF0ll;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(static_cast<unsigned>(bytes)), 0x00020000);
}
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, static_cast<int>(off), 0, 0));
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned off, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, static_cast<int>(off), 0, 0);
}

// ------------------------------------------------------------------ thin fronts (few own columns), 16 or 32 columns
// Two or three levels above the leaves the fronts have a handful of own columns and a long border: almost no flops,
// but every tile step of the workgroup kernels above is a round of dependent loads with two workgroups per CU to hide
// it.  These fronts get one WAVE per unit of work, no LDS, no barrier: the MFMA operands are loaded from global
// memory straight into the operand layout (lane (i, k) of v_mfma_f64_16x16x4: A[i][k] and B[k][i]), everything a
// wave needs is requested in one or two rounds, and 12 to 16 waves per CU are in flight.
//
// forward: wave = 16-row blocks rb, rb + WPF, ... of [T; M21].  K = own columns (NKS steps of 4).
// WPF: waves per front (1: the right-hand side block is loaded once per front; more on levels of fewer than 2048 fronts:
// the waves of a workgroup share a front's row blocks); TRI: T has lower triangular diagonal blocks (Cholesky path) --
// false on the Bunch-Kaufman path, whose diagonal blocks are dense.
// The right-hand sides, the row blocks' matrix operands, carries and results go through raw buffer accesses (planes and
// the caller's block below 4 GB: checked by the launcher, which sends larger problems to the tile kernels): one 32-bit
// offset per access instead of a selected 64-bit address, masked lanes out of range -- the registers that buys are a
// third and fourth wave per SIMD.  Fb is [T; M21] in blocks of 16 rows, each block column-major: the operand of a
// K-step is one contiguous piece.
template <int KB, int NKS, int NSL, int WPF, bool TRI>
__global__ __launch_bounds__(64 * WPF) __attribute__((amdgpu_waves_per_eu((KB == 32) ? (NSL == 0 ? kThinWavesLeaf : ((NSL == 2 && NKS <= 8) ? kThinWavesKids : 1)) : 1)))
void fwd_thin_kernel(FrontArrays fa, const WgRec* __restrict__ recs, const double* X, int ldx, double alpha, double* V,
                     double* __restrict__ Y, int kb, const double* __restrict__ Fb) {
  constexpr int NB = KB / 16;
  const WgRec w = recs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int ns = w.ns, d = ns + w.bs;
  const int nslot = ((w.flags & 2) != 0) ? fa.nslot : 0;
  const int64_t vslot = fa.vrows * KB, vbase = w.voff;
  // v1 = alpha * X[own rows] + carries, as B operands: lane (k, n) holds v1[4 s + k][16 nb + n]
  double b[NKS][NB];
  {
    int xi[NKS], mo[NKS];  // planes no child wrote on a row are not read (mo: which ones did)
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      xi[s] = *((4 * s + lk < ns) ? fa.v_src + vbase + 4 * s + lk : fa.neg1);
      mo[s] = (NSL > 0) ? *((4 * s + lk < ns && nslot > 0) ? fa.cmask + vbase + 4 * s + lk : reinterpret_cast<const int*>(fa.zero)) : 0;
    }
    {  // (the launcher has checked that the caller's block and a plane stay below 4 GB)
      const __amdgpu_buffer_rsrc_t rx = buf_rsrc(X, kBufMax);
#pragma unroll
      for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int o = 4 * s + lk, n = 16 * nb + li;
          const bool ok = xi[s] >= 0 && n < kb;
          const double xv = alpha * buf_load(rx, ok ? static_cast<unsigned>((static_cast<int64_t>(xi[s]) * ldx + n) * 8) : kBufOob);
          const unsigned coff = static_cast<unsigned>(((vbase + o) * KB + n) * 8);
          double v = 0.0;
#pragma unroll
          for (int sl = 0; sl < NSL; ++sl)
            v += buf_load(buf_rsrc(V + sl * vslot, vslot * 8), (ok && ((mo[s] >> sl) & 1)) ? coff : kBufOob);
          b[s][nb] = xv + v;
        }
    }
  }
  double* Vout = V + static_cast<int64_t>(w.slot) * vslot;
  const int nrb = (d + 15) >> 4;
  // the plane masks of all rows, lane by lane (fronts of up to 384 rows): read with ds_bpermute in the row blocks, so
  // that their carry loads do not wait for a mask load of their own
  const bool mreg = NSL > 0 && nslot > 0 && d <= 384;
  int M[6];
#pragma unroll
  for (int q = 0; q < 6; ++q)
    M[q] = (NSL > 0) ? *((mreg && 64 * q + lane < d) ? fa.cmask + vbase + 64 * q + lane : reinterpret_cast<const int*>(fa.zero)) : 0;
  {
    // lane offsets are bytes inside the front's block of Fb (d x ns doubles), a plane of V or the block Y
    const __amdgpu_buffer_rsrc_t ra = buf_rsrc(Fb + w.ftoff, static_cast<int64_t>(d) * ns * 8);
    const __amdgpu_buffer_rsrc_t ry = buf_rsrc(Y, (vbase + d) * kb * 8);
    const __amdgpu_buffer_rsrc_t rv = buf_rsrc(Vout, vslot * 8);
    for (int rb = wave; rb < nrb; rb += WPF) {
      const int smax = (TRI && 16 * rb + 16 <= ns) ? 4 * (rb + 1) : NKS;
      // K-steps past the diagonal block of T are not requested either (offset out of range: 0.0 without a memory access)
      // (row block rb of Fb at 16 rb ns, column-major with `rows` rows -- the operand of
      // K-step s is one contiguous piece of 4 x rows doubles, lanes past the block or the front's columns request nothing)
      const int rows = min(16, d - 16 * rb);
      const unsigned aoff = (li < rows) ? static_cast<unsigned>(16 * rb * ns + lk * rows + li) * 8u : 0x80000000u;
      const int nkse = (ns + 3) >> 2, kend = min(smax, nkse);                   // (wave-uniform)
      const unsigned aoffl = (4 * (nkse - 1) + lk < ns) ? aoff : 0x80000000u;  // the front's last K-step: its existing columns
      double a[NKS];
#pragma unroll
      for (int s = 0; s < NKS; ++s)  // (a front's block is far below 2 GB: the offsets stay out of range once they are)
        a[s] = buf_load(ra, (s < kend) ? ((s == nkse - 1) ? aoffl : aoff) + static_cast<unsigned>(32 * s * rows) : 0x80000000u);
      int di[4];
      double sg[4], cg[4][NB];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ro = 16 * rb + lk + 4 * reg;
        const bool border = ro >= ns && ro < d;
        di[reg] = *(border ? fa.rel + w.bptr + (ro - ns) : fa.neg1);
        sg[reg] = *((ro < ns) ? fa.sgn + w.c0 + ro : fa.zero);
      }
      if constexpr (NSL > 0) {
        int mk[4];
        if (mreg) {
          const int q = rb >> 2;
          const int Mq = (q == 0) ? M[0] : (q == 1) ? M[1] : (q == 2) ? M[2] : (q == 3) ? M[3] : (q == 4) ? M[4] : M[5];
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int ro = 16 * rb + lk + 4 * reg;
            const int m = __shfl(Mq, ro & 63);
            mk[reg] = (ro >= ns && ro < d) ? m : 0;
          }
        } else {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int ro = 16 * rb + lk + 4 * reg;
            mk[reg] = *((ro >= ns && ro < d && nslot > 0) ? fa.cmask + vbase + ro : reinterpret_cast<const int*>(fa.zero));
          }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ro = 16 * rb + lk + 4 * reg;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int n = 16 * nb + li;
            const unsigned coff = static_cast<unsigned>(((vbase + ro) * KB + n) * 8);
            double v = 0.0;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
              v += buf_load(buf_rsrc(V + sl * vslot, vslot * 8), (n < kb && ((mk[reg] >> sl) & 1)) ? coff : kBufOob);
            cg[reg][nb] = v;
          }
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) cg[reg][nb] = 0.0;
      }
      double4_t c[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) c[nb] = double4_t{0.0, 0.0, 0.0, 0.0};
      // products with K-steps past the front's columns or past the diagonal block of T meet zeros: skipped (wave-uniform
      // bound; only in the variants of up to 8 K-steps: guarding all costs 33 registers, 197-205 against 190 us on the leaf launch)
      const int kmax = (NKS <= 8) ? kend : NKS;
#pragma unroll
      for (int s = 0; s < NKS; ++s)
        if (NKS > 8 || s < kmax) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) c[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s][nb], c[nb], 0, 0, 0);
        }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one wait, then the stores back to back
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ro = 16 * rb + lk + 4 * reg;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int n = 16 * nb + li;
          if (ro < ns) {  // (wave-uniform per register only at multiples of 4 rows: the mask goes into the offset)
            buf_store(ry, (n < kb) ? static_cast<unsigned>(((vbase + ro) * kb + n) * 8) : kBufOob, sg[reg] * c[nb][reg]);
          } else {
            const int64_t drow = (w.scratch != 0) ? vbase + ro : w.pvoff + di[reg];
            buf_store(rv, (di[reg] >= 0 && n < kb) ? static_cast<unsigned>((drow * KB + n) * 8) : kBufOob, cg[reg][nb] - c[nb][reg]);
          }
        }
      }
    }
  }
}

// backward: wave = front (all NOB blocks of 16 own columns: the gathered rows are loaded once); K = all d rows of the
// front in chunks of 4 CH, from the transposed copy Ft (row r of [T; M21] at r * ns: a K-step reads whole rows).  The
// border rows of the caller's block are found through bout, which the wave holds lane by lane (bs <= 320) and reads
// with ds_bpermute: no dependent index round per chunk.
template <int KB, int NOB, int CH, bool TRI>
__global__ __launch_bounds__(64) void bwd_thin_kernel(FrontArrays fa, const WgRec* __restrict__ recs,
                                                      const double* __restrict__ Ft, const double* __restrict__ Y,
                                                      double* Out, int ldo, int kb) {
  constexpr int NB = KB / 16;
  const WgRec w = recs[blockIdx.x];
  const int ns = w.ns, bs = w.bs, d = ns + bs;
  const int lane = threadIdx.x;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t vbase = w.voff;
  const int* __restrict__ bout = fa.bout + w.bptr;
  const double* Fp = Ft + w.ftoff;
  int I[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) I[q] = *((64 * q + lane < bs) ? bout + 64 * q + lane : fa.neg1);
  int oi[NOB][4];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int oo = 16 * ob + lk + 4 * reg;
      oi[ob][reg] = *((oo < ns) ? fa.v_src + vbase + oo : fa.neg1);
    }
  double4_t c[NOB][NB];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) c[ob][nb] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int nchunk = (d + 4 * CH - 1) / (4 * CH);
  for (int ch = 0; ch < nchunk; ++ch) {
    double a[CH][NOB], b[CH][NB];
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const int k = 4 * CH * ch + 4 * s + lk;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const int o = 16 * ob + li;
        // (rows of T above the diagonal block of these 16 columns hold zeros: wave-uniform test, not fetched)
        a[s][ob] = *((k < d && o < ns && !(TRI && 4 * CH * ch + 4 * s + 3 < 16 * ob)) ? Fp + static_cast<int64_t>(k) * ns + o : fa.zero);
      }
      const int e = k - ns;  // border entry
      const int r0 = __shfl(I[0], e & 63), r1 = __shfl(I[1], e & 63), r2 = __shfl(I[2], e & 63), r3 = __shfl(I[3], e & 63);
      const int r4 = __shfl(I[4], e & 63);
      const int q = e >> 6;
      const int ri = (e < 0 || k >= d) ? -1 : (q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : q == 3 ? r3 : r4);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = 16 * nb + li;
        const double* p = (e < 0) ? Y + (vbase + k) * kb + n : Out + static_cast<int64_t>(ri) * ldo + n;
        const double v = *((n < kb && (e < 0 || ri >= 0)) ? p : fa.zero);
        b[s][nb] = (e < 0) ? v : -v;
      }
    }
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const int k0 = 4 * CH * ch + 4 * s;  // products with rows past the front, columns past the front's or entries above
      if (KB < 32 || k0 < d) {             // the diagonal block are products with zeros: skipped (wave-uniform tests;
#pragma unroll                             // 32 columns only: the 16-column variants schedule worse with the guards)
        for (int ob = 0; ob < NOB; ++ob)
          if (KB < 32 || (16 * ob < ns && !(TRI && k0 + 3 < 16 * ob))) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              c[ob][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][ob], b[s][nb], c[ob][nb], 0, 0, 0);
          }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = 16 * nb + li;
        if (oi[ob][reg] >= 0 && n < kb) Out[static_cast<int64_t>(oi[ob][reg]) * ldo + n] = c[ob][nb][reg];
      }
}


// Ft(r, o) = [T; M21](r, o), row-major d x ns per front: 64 x 64 tiles through LDS
__global__ __launch_bounds__(kThreads) void transpose_front_kernel(FrontArrays fa, const int* __restrict__ tr_pref,
                                                                  int nfronts, const int64_t* __restrict__ ftoff,
                                                                  const double* __restrict__ F,
                                                                  const double* __restrict__ Tb, double* __restrict__ Ft,
                                                                  double* __restrict__ Fb) {
  __shared__ double tile[TW][TW + 1];
  const int f = find_slot(tr_pref, nfronts, blockIdx.x);
  const int ns = fa.ns[f], d = ns + fa.bs[f];
  const int nct = (ns + TW - 1) / TW;
  const int local = blockIdx.x - tr_pref[f];
  const int rt = local / nct, ct = local - rt * nct;
  const int r0 = rt * TW, o0 = ct * TW;
  const int r = threadIdx.x & (TW - 1), q = threadIdx.x >> 6;
  for (int oo = q; oo < TW; oo += kThreads / TW) {  // coalesced along r
    const int gr = r0 + r, go = o0 + oo;
    double v = 0.0;
    if (gr < d && go < ns)
      v = (gr < ns) ? Tb[fa.toff[f] + static_cast<int64_t>(go) * ns + gr] : F[fa.foff[f] + static_cast<int64_t>(go) * d + gr];
    tile[r][oo] = v;
  }
  __syncthreads();
  for (int rr = q; rr < TW; rr += kThreads / TW) {  // coalesced along o
    const int gr = r0 + rr, go = o0 + r;
    if (gr < d && go < ns) Ft[ftoff[f] + static_cast<int64_t>(gr) * ns + go] = tile[rr][r];
  }
  // Fb: the same entries in blocks of 16 rows, a block stored column-major with its own row count as leading dimension
  // (the last block of a front may have fewer than 16 rows): entry (r, o) at (r / 16) * 16 * ns + o * rows + r % 16
  for (int oo = q; oo < TW; oo += kThreads / TW) {  // 16 consecutive lanes write one 128-byte piece
    const int gr = r0 + r, go = o0 + oo;
    if (gr < d && go < ns) {
      const int rb = gr >> 4, rows = min(16, d - 16 * rb);
      Fb[ftoff[f] + static_cast<int64_t>(rb) * 16 * ns + static_cast<int64_t>(go) * rows + (gr & 15)] = tile[r][oo];
    }
  }
}

// Fragment-major copies of [T; M21] for the fronts with several column tiles.  The level kernels of those fronts feed
// v_mfma_f64_16x16x4 from global memory; read from the column-major panel, a wave's operand load is four 128-byte
// pieces a leading dimension apart (measured: slower than staging 512-byte column pieces through LDS).  Here every
// 64 x 64 tile is stored in the order the four waves of a workgroup consume it: K-step kk, wave v, lane l at
// kk*256 + v*64 + l, so one K-step of the workgroup is 2 KB of consecutive memory and a tile one contiguous block.
//   forward  (Fm): tile (row tile rt, column tile ct), lane (i, k) = R(row0 + 16 v + i, 64 ct + 4 kk + k); row tiles =
//                  the own tiles (ct <= rt) and then the border tiles (from row ns); K-steps: those meeting columns < ns
//   backward (Bm): tile (column tile ct, chain step t), lane (i, k) = R(rbase + 4 kk + k, 64 ct + 16 v + i); K-steps:
//                  those meeting rows of the tile
// Cells outside the front hold zeros: the sweeps need no lane masks.
__global__ __launch_bounds__(kThreads) void pack_frag_kernel(FrontArrays fa, const FragFront* __restrict__ ff,
                                                            const int* __restrict__ mt_pref, int nff,
                                                            const double* __restrict__ F, const double* __restrict__ Tb,
                                                            double* __restrict__ Fm, double* __restrict__ Bm) {
  __shared__ double tile[TW][TW + 1];
  const int q = find_slot(mt_pref, nff, blockIdx.x);
  const FragFront r = ff[q];
  const int f = r.f;
  const int ns = fa.ns[f], d = ns + fa.bs[f];
  const int local = blockIdx.x - mt_pref[q];
  const int rt = local / r.nst, ct = local - rt * r.nst;
  const bool own = rt < r.nst;
  if (own && ct > rt) return;  // T is block lower triangular: those tiles are never read
  const int row0 = own ? rt * TW : ns + (rt - r.nst) * TW;
  const int rows = min(TW, (own ? ns : d) - row0);
  const int c0 = ct * TW, cols = min(TW, ns - c0);
  const int tr = threadIdx.x & (TW - 1), tq = threadIdx.x >> 6;
  for (int oo = tq; oo < TW; oo += kThreads / TW) {  // coalesced along the rows of the column-major source
    double v = 0.0;
    if (tr < rows && oo < cols)
      v = own ? Tb[fa.toff[f] + static_cast<int64_t>(c0 + oo) * ns + row0 + tr]
              : F[fa.foff[f] + static_cast<int64_t>(c0 + oo) * d + row0 + tr];
    tile[tr][oo] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  {
    const int nk = (cols + 3) >> 2;
    double* dst = Fm + r.fm + 256 * (frag_fwd_steps(r.nst, r.nko, rt) + 16LL * ct) + wave * 64 + lane;
    for (int kk = 0; kk < nk; ++kk) dst[kk * 256] = tile[16 * wave + li][4 * kk + lk];
  }
  {
    const int nown = r.nst - ct;
    const int t = own ? rt - ct : nown + (rt - r.nst);
    const int nk = (rows + 3) >> 2;
    double* dst = Bm + r.bm + 256 * (frag_bwd_steps(r.nst, r.nbt, r.nko, r.nkb, ct) + frag_bwd_in_chain(nown, r.nko, t)) +
                  wave * 64 + lane;
    for (int kk = 0; kk < nk; ++kk) dst[kk * 256] = tile[4 * kk + lk][16 * wave + li];
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
  // sweeps: per-level workgroup prefixes (forward: row tiles of [T; M21], backward: column tiles), gather lists
  WgRec *d_fwd_wg = nullptr, *d_bwd_wg = nullptr;
  int *d_tri_pref = nullptr, *d_m_pref = nullptr, *d_bout = nullptr, *d_tickets = nullptr;
  double* d_P = nullptr;
  int64_t n_slabs = 0;
  int n_tickets = 0;
  int *d_ov_dst = nullptr, *d_ov_ptr = nullptr, *d_ov_src = nullptr;
  int nslot = 0, nplanes = 0;        // planes the parents read; planes allocated (+ scratch when there are surplus children)
  int64_t* d_toff = nullptr;
  double *d_T = nullptr, *d_aux = nullptr;  // aux: {0.0, (int) -1}
  double* d_red = nullptr;                  // 512 partial results of small reductions
  std::vector<int> h_fwd_ptr, h_bwd_ptr;  // per level: first workgroup record
  std::vector<int> h_thin_fwd, h_thin_bwd;  // per level: 4, 8 or 16 K-steps (of 4 own columns) of the wave-per-block kernels, 0: the tile kernels
  std::vector<int> h_fwd_kd, h_bwd_kd;    // per level: LDS tile rows of the single-column-tile launches (multiple of 8)
  std::vector<char> h_lvl_two;            // per level: every front has at most two children (two-plane kernels)
  std::vector<char> h_lvl_leaf;           // per level: no front has children (kernels without carry loads)
  std::vector<int> h_fwd_nsingle;         // per level: leading records of single-column-tile fronts (own kernel)
  // levels whose multi-tile fronts get v1 pre-assembled (v1_assemble_kernel): one record per front and column tile
  WgRec* d_pre_wg = nullptr;
  std::vector<int> h_pre_ptr;             // per level: first record (none: the level's workgroups gather v1 themselves)
  bool has_v1 = false;                    // one more plane per sweep width, behind all carry planes: receives v1
  std::vector<int> h_bwd_nsingle;         // per level: leading backward records of single-column-tile fronts
  std::vector<int> h_bwd_mxbs;            // per level: the longest border among the fronts with several column tiles
  // narrow sweeps (k <= 8): one record per single-tile front and level, the transposed copy of [T; M21]
  WgRec* d_wave_wg = nullptr;
  std::vector<int> h_wave_ptr;
  int64_t* d_ftoff = nullptr;
  int* d_tr_pref = nullptr;
  double* d_Ft = nullptr;
  double* d_Fb = nullptr;  // [T; M21] in blocks of 16 rows, each block column-major (thin forward kernels: one contiguous piece per MFMA operand load)
  double *d_Fm = nullptr, *d_Bm = nullptr;  // fragment-major copies (fronts with several column tiles)
  FragFront* d_ff = nullptr;
  int* d_mt_pref = nullptr;
  int n_ff = 0, n_mt = 0;
  int64_t fm_doubles = 0, bm_doubles = 0;
  int64_t ft_doubles = 0;
  int n_tr = 0;
  std::vector<int> ov_lvl_ptr;       // per level: range of overflow rows (extra rows of V after the sumd front rows)
  int64_t t_doubles = 0, v_rows = 0;
  int n_tri = 0, n_m21 = 0;
  int64_t *d_a_src = nullptr, *d_a_dst = nullptr;
  int* d_v_src = nullptr;
  int* d_cmask = nullptr;
  double *d_data = nullptr, *d_F = nullptr, *d_Inv = nullptr, *d_V = nullptr, *d_Y = nullptr, *d_sgn = nullptr;
  int n_negative = 0;
  int n_perturbed = 0;     // static pivots of the Bunch-Kaufman path (columns singular inside their panel block)
  double pivtol = 0.0;
  bool pivoted = false;  // the last numeric phase ran the Bunch-Kaufman panel kernel (dense diagonal blocks in T)
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
    a.toff = d_toff;
    a.nslot = nslot;
    a.vrows = v_rows;
    a.v_src = d_v_src;
    a.cmask = d_cmask;
    a.bout = d_bout;
    a.sgn = d_sgn;
    a.zero = d_aux;
    a.neg1 = reinterpret_cast<const int*>(d_aux + 1);
    a.W = sym->W;
    a.tri = pivoted ? 0 : 1;
    a.pivtol = pivtol;
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

int numeric(eigd_factor* f, const double* data, bool on_device = false, bool pivot = false) {
  const Symbolic& s = *f->sym;
  hipStream_t st = f->ctx->stream;
  f->pivoted = pivot;
  if (data != nullptr)  // (the pivoting pass re-reads the values the first pass brought in)
    EIGD_HIP(hipMemcpyAsync(f->d_data, data, sizeof(double) * f->data_len,
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  EIGD_HIP(hipMemsetAsync(f->d_F, 0, sizeof(double) * s.front_doubles, st));
  EIGD_HIP(hipMemsetAsync(f->d_flag, 0, 3 * sizeof(int), st));
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
      if (pivot)
        hipLaunchKernelGGL(ldlt_bk_inv_kernel, dim3(na), dim3(kThreads), 0, st, fa, fronts, step, f->d_F, f->d_Inv,
                           f->d_flag);
      else
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
  if (f->n_tri > 0) {
    hipLaunchKernelGGL(trinv_kernel, dim3(f->n_tri), dim3(kThreads), 0, st, fa, f->d_tri_pref, s.nfronts, f->d_F,
                       f->d_Inv, f->d_T);
    EIGD_LAUNCH_CHECK();
  }
  if (f->n_m21 > 0) {
    hipLaunchKernelGGL(m21_kernel, dim3(f->n_m21), dim3(kThreads), 0, st, fa, f->d_m_pref, s.nfronts, f->d_F, f->d_T);
    EIGD_LAUNCH_CHECK();
  }
  if (f->n_tr > 0) {
    hipLaunchKernelGGL(transpose_front_kernel, dim3(f->n_tr), dim3(kThreads), 0, st, fa, f->d_tr_pref, s.nfronts, f->d_ftoff,
                       f->d_F, f->d_T, f->d_Ft, f->d_Fb);
    EIGD_LAUNCH_CHECK();
  }
  if (f->n_mt > 0) {
    hipLaunchKernelGGL(pack_frag_kernel, dim3(f->n_mt), dim3(kThreads), 0, st, fa, f->d_ff, f->d_mt_pref, f->n_ff, f->d_F,
                       f->d_T, f->d_Fm, f->d_Bm);
    EIGD_LAUNCH_CHECK();
  }
  int flag[3] = {0, 0, 0};
  EIGD_HIP(hipMemcpyAsync(flag, f->d_flag, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
  EIGD_HIP(hipStreamSynchronize(st));
  f->n_negative = flag[1];
  f->n_perturbed = flag[2];
  // Not positive definite (negative or unusable pivots on the Cholesky path): the shift lies inside the spectrum.
  // Factor again with Bunch-Kaufman pivoting inside the panels -- the positive definite shifts of the reference's
  // examples never come here and keep the plain (bitwise unchanged) Cholesky path.
  if (!pivot && (flag[0] != 0 || flag[1] != 0)) {
    // the threshold of the static pivots: sqrt(eps) * max |a_ij| (the values are on the device: reduced there)
    constexpr int nbm = 512;
    hipLaunchKernelGGL(absmax_kernel, dim3(nbm), dim3(kThreads), 0, st, f->data_len, f->d_data, f->d_red);
    EIGD_LAUNCH_CHECK();
    double part[nbm];
    EIGD_HIP(hipMemcpyAsync(part, f->d_red, sizeof(part), hipMemcpyDeviceToHost, st));
    EIGD_HIP(hipStreamSynchronize(st));
    double amax = 0.0;
    for (double v : part) amax = std::max(amax, v);
    f->pivtol = 1.4901161193847656e-08 * amax;
    return numeric(f, nullptr, true, true);
  }
  if (flag[0] != 0) {
    set_error("zero or non-finite pivot in front %d: the (shifted) matrix is singular to working precision -- move the "
              "shift away from an eigenvalue",
              flag[0] - 1);
    return EIGD_E_NOTSPD;
  }
  return EIGD_OK;
}


// Fewest multi-tile workgroups of a level whose right-hand sides are pre-assembled (EIGD_PRE_MIN_WG: experiments, tests)
int pre_assembly_min_workgroups() {
  const char* e = std::getenv("EIGD_PRE_MIN_WG");
  const int v = e ? std::atoi(e) : 0;
  return v > 0 ? v : 512;
}

// Launch policy of the sweeps (all measured on the 1 M-dof benchmark; the experiments behind each number are in
// docs/LOG.md).  Sweeps of up to 8 columns run their single-tile fronts wave by wave (VALU, readlane broadcast); 5 to 8
// columns otherwise go through the 16-column MFMA kernels.
constexpr int kWaveMaxKpt = 2;   // widest sweep (units of 4 columns) whose single-tile fronts use the readlane wave kernels

template <int KPT>
int sweep(eigd_factor* f, hipStream_t st, double* wV, double* wY, double* wP, int* wT, const double* dIn, int ldin,
          double* dX, int ldx, int kb, double alpha) {
  const Symbolic& s = *f->sym;
  const FrontArrays fa = f->fa();
  // every sweep width (KB = 4, 8, 16, 32 columns) has its own set of carry planes: rows are KB wide and the
  // entries no child writes stay zero for good
  constexpr int KB = 4 * KPT;
  // (the planes of pre-assembled right-hand sides lie behind the carry planes of ALL widths: with one more plane inside
  // every width's set the thin forward kernels of the shell model ran 4 to 10 % slower -- same kernels, same bytes, other
  // addresses)
  const int ncarry = f->nplanes - (f->has_v1 ? 1 : 0);
  double* const wV1 = f->has_v1 ? wV + static_cast<int64_t>(ncarry) * f->v_rows * kPlaneCols + f->v_rows * (KB - 4) : nullptr;
  wV += static_cast<int64_t>(ncarry) * f->v_rows * (KB - 4);  // 4 + 8 + ... below KB = KB - 4
  // multi-tile levels: every XCD gets a contiguous range of the level's records (see xcd_record)
  auto level_args = [&](const WgRec* wg, int kd = TW, int nwg = 0, bool multi = false) {
    LevelArgs la;
    la.wg = wg;
    la.P = wP;
    la.tickets = wT;
    la.kb = kb;
    la.kd = kd;
    la.nwg = nwg;
    la.per_xcd = multi ? (nwg + 7) / 8 : 0;
    la.V1 = wV1;
    return la;
  };
  auto multi_grid = [&](int nwg) { return dim3(8 * ((nwg + 7) / 8)); };
  auto lds_bytes = [](int kd) { return static_cast<unsigned>(sizeof(double) * (kd + 1) * (TLD + Tile<KPT>::BLD)); };
  // fronts with several column tiles, MFMA widths: matrix operands straight from the fragment-major copies (only the
  // vector block in LDS); the vector-FMA widths stage the column-major panels through LDS
  const unsigned lds_frag = static_cast<unsigned>(sizeof(double) * (TW + 1) * Tile<KPT>::BLD);
  const double* sF = f->d_F;  // the panels [T; M21] in place (F and T)
  const double* sT = f->d_T;
  // raw buffer accesses take 32-bit byte offsets: a carry plane and the caller's block must stay below 4 GB, else the
  // single-tile fronts of the MFMA widths go through the tile kernels
  const bool thin_buf = f->v_rows * static_cast<int64_t>(KB) * 8 <= kBufMax && static_cast<int64_t>(s.n) * ldin * 8 <= kBufMax;
  // ---- forward: leaves -> root.  Y receives S z, the border rows of V the carries.
  for (int l = 0; l < s.nlevels; ++l) {
    const int nov = f->ov_lvl_ptr[l + 1] - f->ov_lvl_ptr[l];
    if (nov > 0) {
      const int first = f->ov_lvl_ptr[l];
      const int64_t plane = f->v_rows * KB;
      hipLaunchKernelGGL(overflow_sum_kernel, dim3((nov * kb + 255) / 256), dim3(256), 0, st, nov, f->d_ov_dst + first,
                         f->d_ov_ptr + first, f->d_ov_src, kb, KB, wV + (f->nslot - 1) * plane, wV + f->nslot * plane);
      EIGD_LAUNCH_CHECK();
    }
    const int nwg = f->h_fwd_ptr[l + 1] - f->h_fwd_ptr[l], nsingle = f->h_fwd_nsingle[l];
    const bool two = f->h_lvl_two[l] != 0;  // no front of this level has more than two children
    const bool leaf = f->h_lvl_leaf[l] != 0;  // ... has children at all
    const int nwave = f->h_wave_ptr[l + 1] - f->h_wave_ptr[l];
    bool narrow = false;
    if (KPT <= kWaveMaxKpt) {
      if (nwave > 0) {  // narrow sweep: one wave per tile of the single-tile fronts, two waves per front
        const WgRec* recs = f->d_wave_wg + f->h_wave_ptr[l];
        if (leaf)
          hipLaunchKernelGGL((fwd_wave_kernel<KB, 0, 2>), dim3(nwave), dim3(128), 0, st, fa, recs, sF, sT, dIn, ldin, alpha, wV,
                             wY, kb);
        else if (two)
          hipLaunchKernelGGL((fwd_wave_kernel<KB, 2, 2>), dim3(nwave), dim3(128), 0, st, fa, recs, sF, sT, dIn, ldin, alpha, wV,
                             wY, kb);
        else
          hipLaunchKernelGGL((fwd_wave_kernel<KB, kMaxS + 1, 2>), dim3(nwave), dim3(128), 0, st, fa, recs, sF, sT, dIn, ldin,
                             alpha, wV, wY, kb);
        EIGD_LAUNCH_CHECK();
        narrow = true;
      }
    }
    if constexpr (KPT >= 4) {
      const int nks = f->h_thin_fwd[l];
      if (!narrow && nwave > 0 && nks > 0 && thin_buf) {  // thin fronts: one wave per block of rows, operands straight from memory
        const WgRec* recs = f->d_wave_wg + f->h_wave_ptr[l];
#define EIGD_THIN_FWD(NKS, NSLV, WPFV)                                                                                   \
  do {                                                                                                                   \
    if (fa.tri)                                                                                                          \
      hipLaunchKernelGGL((fwd_thin_kernel<KB, NKS, NSLV, WPFV, true>), dim3(nwave), dim3(64 * WPFV), 0, st, fa, recs, dIn, ldin, \
                         alpha, wV, wY, kb, f->d_Fb);                                                                    \
    else                                                                                                                 \
      hipLaunchKernelGGL((fwd_thin_kernel<KB, NKS, NSLV, WPFV, false>), dim3(nwave), dim3(64 * WPFV), 0, st, fa, recs, dIn, ldin, \
                         alpha, wV, wY, kb, f->d_Fb);                                                                    \
  } while (0)
        if (!leaf && two && nks >= 12) {
          // fronts of 33 to 64 own columns with carry planes: as many waves per front as give the level >= 2048 waves
          const int wpf = (nwave >= 2048) ? 1 : (nwave >= 1024) ? 2 : 4;
          if (nks == 12 && wpf == 1)
            EIGD_THIN_FWD(12, 2, 1);
          else if (nks == 12 && wpf == 2)
            EIGD_THIN_FWD(12, 2, 2);
          else if (nks == 12)
            EIGD_THIN_FWD(12, 2, 4);
          else if (wpf == 1)
            EIGD_THIN_FWD(16, 2, 1);
          else if (wpf == 2)
            EIGD_THIN_FWD(16, 2, 2);
          else
            EIGD_THIN_FWD(16, 2, 4);
        } else if (leaf && nks == 4)
          EIGD_THIN_FWD(4, 0, 1);
        else if (leaf && nks == 8)
          EIGD_THIN_FWD(8, 0, 1);
        else if (leaf && nks == 12)
          EIGD_THIN_FWD(12, 0, 1);
        else if (leaf && nks == 14)
          EIGD_THIN_FWD(14, 0, 1);
        else if (leaf)
          EIGD_THIN_FWD(16, 0, 1);
        else if (nks == 4 && two)
          EIGD_THIN_FWD(4, 2, 1);
        else if (nks == 4)
          EIGD_THIN_FWD(4, kMaxS + 1, 1);
        else if (nks == 8 && two)
          EIGD_THIN_FWD(8, 2, 1);
        else if (nks == 8)
          EIGD_THIN_FWD(8, kMaxS + 1, 1);
        else if (two)
          EIGD_THIN_FWD(16, 2, 1);
        else
          EIGD_THIN_FWD(16, kMaxS + 1, 1);
#undef EIGD_THIN_FWD
        EIGD_LAUNCH_CHECK();
        narrow = true;
      }
    }
    if (!narrow && nsingle > 0) {
      const int kd = (KPT == 4) ? TW : f->h_fwd_kd[l];  // (the direct-fragment path reads all 64 rows of the block)
      const LevelArgs la = level_args(f->d_fwd_wg + f->h_fwd_ptr[l], kd);
      auto lds_bytes = [](int kd) {  // (shadows the general one: the direct-fragment path keeps only the vector block)
        return static_cast<unsigned>(sizeof(double) * (kd + 1) * ((KPT == 4 ? 0 : TLD) + Tile<KPT>::BLD));
      };
      if (leaf)
        hipLaunchKernelGGL((fwd_level_kernel<KPT, true, 0>), dim3(nsingle), dim3(kThreads), lds_bytes(kd), st, fa, la,
                           sF, sT, dIn, ldin, alpha, wV, wY);
      else if (two)
        hipLaunchKernelGGL((fwd_level_kernel<KPT, true, 2>), dim3(nsingle), dim3(kThreads), lds_bytes(kd), st, fa, la,
                           sF, sT, dIn, ldin, alpha, wV, wY);
      else
        hipLaunchKernelGGL((fwd_level_kernel<KPT, true, kMaxS + 1>), dim3(nsingle), dim3(kThreads), lds_bytes(kd), st,
                           fa, la, sF, sT, dIn, ldin, alpha, wV, wY);
      EIGD_LAUNCH_CHECK();
    }
    if (nwg > nsingle) {
      const LevelArgs la = level_args(f->d_fwd_wg + f->h_fwd_ptr[l] + nsingle, TW, nwg - nsingle, true);
      if constexpr (Tile<KPT>::kMfma) {
        const int npre = f->h_pre_ptr[l + 1] - f->h_pre_ptr[l];
        if (npre > 0) {  // v1 of the level's fronts once, then row-tile workgroups that read it as it lies
          const WgRec* pre = f->d_pre_wg + f->h_pre_ptr[l];
          if (two) {
            hipLaunchKernelGGL((v1_assemble_kernel<KPT, 2>), dim3(npre), dim3(kThreads), 0, st, fa, pre, dIn, ldin, alpha, wV,
                               wV1, kb);
            hipLaunchKernelGGL((fwd_level_kernel<KPT, false, 2, true, true>), multi_grid(nwg - nsingle), dim3(kThreads),
                               lds_frag, st, fa, la, f->d_Fm, sT, dIn, ldin, alpha, wV, wY);
          } else {
            hipLaunchKernelGGL((v1_assemble_kernel<KPT, kMaxS + 1>), dim3(npre), dim3(kThreads), 0, st, fa, pre, dIn, ldin,
                               alpha, wV, wV1, kb);
            hipLaunchKernelGGL((fwd_level_kernel<KPT, false, kMaxS + 1, true, true>), multi_grid(nwg - nsingle),
                               dim3(kThreads), lds_frag, st, fa, la, f->d_Fm, sT, dIn, ldin, alpha, wV, wY);
          }
        } else if (two)
          hipLaunchKernelGGL((fwd_level_kernel<KPT, false, 2, true>), multi_grid(nwg - nsingle), dim3(kThreads),
                             lds_frag, st, fa, la, f->d_Fm, sT, dIn, ldin, alpha, wV, wY);
        else
          hipLaunchKernelGGL((fwd_level_kernel<KPT, false, kMaxS + 1, true>), multi_grid(nwg - nsingle),
                             dim3(kThreads), lds_frag, st, fa, la, f->d_Fm, sT, dIn, ldin, alpha, wV, wY);
      } else {
        if (two)
          hipLaunchKernelGGL((fwd_level_kernel<KPT, false, 2>), multi_grid(nwg - nsingle), dim3(kThreads),
                             lds_bytes(TW), st, fa, la, sF, sT, dIn, ldin, alpha, wV, wY);
        else
          hipLaunchKernelGGL((fwd_level_kernel<KPT, false, kMaxS + 1>), multi_grid(nwg - nsingle), dim3(kThreads),
                             lds_bytes(TW), st, fa, la, sF, sT, dIn, ldin, alpha, wV, wY);
      }
      EIGD_LAUNCH_CHECK();
    }
  }
  // ---- backward: root -> leaves, straight into the caller's block.
  for (int l = s.nlevels - 1; l >= 0; --l) {
    const int nwg = f->h_bwd_ptr[l + 1] - f->h_bwd_ptr[l];
    if (nwg == 0) continue;
    const int nwave = f->h_wave_ptr[l + 1] - f->h_wave_ptr[l];
    const int nsb = f->h_bwd_nsingle[l];
    const bool narrow = KPT <= kWaveMaxKpt && nwave > 0;  // narrow sweep: the single-tile fronts go wave by wave
    bool thin = false;
    if constexpr (KPT >= 4) thin = !narrow && nwave > 0 && f->h_thin_bwd[l] > 0;
    if (nwg > nsb) {  // fronts with several column tiles: full 64-row tiles
      const LevelArgs la = level_args(f->d_bwd_wg + f->h_bwd_ptr[l] + nsb, TW, nwg - nsb, true);
      if constexpr (Tile<KPT>::kMfma) {
        if (KPT >= 8 && f->h_bwd_mxbs[l] <= kLidxMax)
          hipLaunchKernelGGL((bwd_level_kernel<KPT, false, true, (KPT >= 8)>), multi_grid(nwg - nsb), dim3(kThreads),
                             lds_frag + 4u * static_cast<unsigned>((f->h_bwd_mxbs[l] + 3) & ~3), st, fa, la, sF, sT, f->d_Bm,
                             wY, dX, ldx);
        else
          hipLaunchKernelGGL((bwd_level_kernel<KPT, false, true>), multi_grid(nwg - nsb), dim3(kThreads), lds_frag, st, fa,
                             la, sF, sT, f->d_Bm, wY, dX, ldx);
      }
      else
        hipLaunchKernelGGL((bwd_level_kernel<KPT, false>), multi_grid(nwg - nsb), dim3(kThreads), lds_bytes(TW), st, fa, la, sF,
                           sT, f->d_Ft, wY, dX, ldx);
      EIGD_LAUNCH_CHECK();
    }
    if (narrow) {
      hipLaunchKernelGGL(bwd_wave_kernel<KB>, dim3(nwave), dim3(64), 0, st, fa, f->d_wave_wg + f->h_wave_ptr[l], f->d_Ft,
                         wY, dX, ldx, kb);
      EIGD_LAUNCH_CHECK();
    } else if (thin) {
      if constexpr (KPT >= 4) {
        const WgRec* recs = f->d_wave_wg + f->h_wave_ptr[l];
        const int nks = f->h_thin_bwd[l];
#define EIGD_THIN_BWD(NOB, CHV)                                                                                   \
  do {                                                                                                            \
    if (fa.tri)                                                                                                   \
      hipLaunchKernelGGL((bwd_thin_kernel<KB, NOB, CHV, true>), dim3(nwave), dim3(64), 0, st, fa, recs, f->d_Ft, wY, \
                         dX, ldx, kb);                                                                            \
    else                                                                                                          \
      hipLaunchKernelGGL((bwd_thin_kernel<KB, NOB, CHV, false>), dim3(nwave), dim3(64), 0, st, fa, recs, f->d_Ft, \
                         wY, dX, ldx, kb);                                                                        \
  } while (0)
        if (nks == 4)
          EIGD_THIN_BWD(1, 8);
        else if (nks == 8)
          EIGD_THIN_BWD(2, 8);
        else
          EIGD_THIN_BWD(4, 4);
#undef EIGD_THIN_BWD
        EIGD_LAUNCH_CHECK();
      }
    } else if (nsb > 0) {  // single-column-tile fronts: LDS tiles as tall as the level needs
      hipLaunchKernelGGL((bwd_level_kernel<KPT, true>), dim3(nsb), dim3(kThreads), lds_bytes(f->h_bwd_kd[l]), st, fa,
                         level_args(f->d_bwd_wg + f->h_bwd_ptr[l], f->h_bwd_kd[l]), sF, sT, f->d_Ft, wY, dX, ldx);
      EIGD_LAUNCH_CHECK();
    }
  }
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
                  f->d_pref_tiles, f->d_cs_child,   f->d_a_src,      f->d_a_dst,    f->d_v_src, f->d_data,   f->d_cmask,
                  f->d_F,         f->d_Inv,         f->d_V,          f->d_Y,        f->d_flag,  f->d_fwd_wg,
                  f->d_bwd_wg,    f->d_tri_pref,    f->d_m_pref,     f->d_ov_dst,   f->d_ov_ptr, f->d_ov_src,
                  f->d_toff,      f->d_T,           f->d_sgn,        f->d_aux,      f->d_bout,  f->d_tickets,
                  f->d_P,         f->d_wave_wg,     f->d_ftoff,      f->d_tr_pref,  f->d_Ft,  f->d_Fb,
                  f->d_Fm,        f->d_Bm,          f->d_ff,         f->d_mt_pref,  f->d_red,   f->d_pre_wg};
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
  // ---- host tables of the sweeps
  const int nf = s.nfronts;
  std::vector<int64_t> toff(static_cast<size_t>(nf) + 1, 0), ftoff(static_cast<size_t>(nf) + 1, 0);
  std::vector<int> tri_pref(static_cast<size_t>(nf) + 1, 0), m_pref(static_cast<size_t>(nf) + 1, 0);
  std::vector<int> tr_pref(static_cast<size_t>(nf) + 1, 0);
  // The copies the sweeps read (Ft, Fb, Fm, Bm) are laid out LEVEL BY LEVEL, in the order the level's workgroups take
  // their fronts: a launch streams one contiguous region (in front order, a postorder, the 8192 six-KB blocks of the
  // level above the leaves lay 128 KB apart in a 1 GB array).  Measured neutral on the mean sweep time; the +-2 % by
  // which a sweep's time moves with where a factor's arrays landed are there in either layout (docs/LOG.md, round 5).
  {
    int64_t acc = 0;
    for (int l = 0; l < s.nlevels; ++l)
      for (int q = s.lvl_ptr[l]; q < s.lvl_ptr[l + 1]; ++q) {
        const int fr = s.lvl_fronts[q];
        ftoff[fr] = acc;
        acc += static_cast<int64_t>(s.f_ns[fr] + s.f_bs[fr]) * s.f_ns[fr];
      }
    ftoff[nf] = acc;  // (the total; entries 0 .. nf - 1 are per front, not a prefix array any more)
  }
  for (int q = 0; q < nf; ++q) {
    const int64_t ns = s.f_ns[q];
    const int64_t dq = ns + s.f_bs[q];
    tr_pref[q + 1] = tr_pref[q] + static_cast<int>(((dq + TW - 1) / TW) * ((ns + TW - 1) / TW));
    toff[q + 1] = toff[q] + ns * ns;
    tri_pref[q + 1] = tri_pref[q] + static_cast<int>((ns + s.W - 1) / s.W);
    m_pref[q + 1] = m_pref[q] + (s.f_bs[q] + TW - 1) / TW;
  }
  // fragment-major copies for the fronts with several column tiles (see pack_frag_kernel)
  std::vector<FragFront> ffr;
  std::vector<int> mt_pref(1, 0), ff_of(static_cast<size_t>(nf), -1);
  int64_t fm_doubles = 0, bm_doubles = 0;
  for (int lq = 0; lq < nf; ++lq) {  // (level by level as well)
    const int q = s.lvl_fronts[lq];
    const int ns = s.f_ns[q], bs = s.f_bs[q];
    const int nst = (ns + TW - 1) / TW, nbt = (bs + TW - 1) / TW;
    if (nst < 2) continue;
    FragFront r;
    r.f = q;
    r.nst = nst;
    r.nbt = nbt;
    r.nko = (ns - TW * (nst - 1) + 3) / 4;
    r.nkb = (nbt > 0) ? (bs - TW * (nbt - 1) + 3) / 4 : 0;
    r.pad = 0;
    r.fm = fm_doubles;
    r.bm = bm_doubles;
    fm_doubles += 256 * frag_fwd_steps(nst, r.nko, nst + nbt);
    bm_doubles += 256 * frag_bwd_steps(nst, nbt, r.nko, r.nkb, nst);
    ff_of[q] = static_cast<int>(ffr.size());
    ffr.push_back(r);
    mt_pref.push_back(mt_pref.back() + (nst + nbt) * nst);
  }
  // carry planes: child number q of a front (ascending front order) writes plane q; children beyond kMaxS write
  // the scratch plane and are summed into the extra plane (overflow_sum_kernel) before their parent's level runs
  std::vector<int> child_no(static_cast<size_t>(nf), 0), nchild(static_cast<size_t>(nf), 0);
  int maxchild = 0;
  for (int c = 0; c < nf; ++c) {
    const int p = s.f_parent[c];
    if (p < 0 || s.f_bs[c] == 0) continue;
    child_no[c] = nchild[p]++;
    maxchild = std::max(maxchild, nchild[p]);
  }
  const bool surplus = maxchild > kMaxS;
  const int ndirect = std::min(maxchild, kMaxS);
  const int nslot = ndirect + (surplus ? 1 : 0);
  int nplanes = std::max(1, nslot + (surplus ? 1 : 0));   // (+ 1 below: the plane of pre-assembled right-hand sides)
  struct Extra { int level; int dst; std::vector<int> src; };
  std::vector<Extra> extras;
  if (surplus) {
    std::vector<int64_t> extra_of(static_cast<size_t>(s.sumd), -1);
    for (int c = 0; c < nf; ++c) {
      const int p = s.f_parent[c];
      if (p < 0 || child_no[c] < kMaxS) continue;
      const int64_t b0 = s.f_bptr[c];
      for (int i = 0; i < s.f_bs[c]; ++i) {
        const int64_t dst = s.f_voff[p] + s.rel[b0 + i];
        if (extra_of[dst] < 0) {
          extra_of[dst] = static_cast<int64_t>(extras.size());
          extras.push_back(Extra{s.f_level[p], static_cast<int>(dst), {}});
        }
        extras[extra_of[dst]].src.push_back(static_cast<int>(s.f_voff[c] + s.f_ns[c] + i));
      }
    }
    std::stable_sort(extras.begin(), extras.end(), [](const Extra& a, const Extra& b) { return a.level < b.level; });
  }
  std::vector<int> ov_dst, ov_ptr, ov_src, ov_lvl_ptr(static_cast<size_t>(s.nlevels) + 1, 0);
  {
    // per level the ov_ptr entries are [count + 1] long so that a level's slice starts at its own offset
    std::vector<int> lvl_cnt(static_cast<size_t>(s.nlevels), 0);
    for (const Extra& e : extras) lvl_cnt[e.level] += 1;
    // layout: ov_dst[x], ov_ptr[x] / ov_ptr[x + 1] with one shared trailing entry (offsets are absolute)
    for (const Extra& e : extras) {
      ov_dst.push_back(e.dst);
      ov_ptr.push_back(static_cast<int>(ov_src.size()));
      for (int q : e.src) ov_src.push_back(q);
    }
    ov_ptr.push_back(static_cast<int>(ov_src.size()));
    for (int l = 0; l < s.nlevels; ++l) ov_lvl_ptr[l + 1] = ov_lvl_ptr[l] + lvl_cnt[l];
  }
  const int64_t v_rows = s.sumd;
  EIGD_REQUIRE(v_rows < (int64_t(1) << 31), "vector workspace has too many rows");

  // workgroup records per level (see WgRec).  Levels with few fronts cut their long chains into groups.
  std::vector<char> has_kids(static_cast<size_t>(nf), 0);
  for (int c = 0; c < nf; ++c)
    if (s.f_parent[c] >= 0 && s.f_bs[c] > 0) has_kids[s.f_parent[c]] = 1;
  std::vector<WgRec> fwd_wg, bwd_wg;
  std::vector<int> h_fwd_ptr(static_cast<size_t>(s.nlevels) + 1, 0), h_bwd_ptr(static_cast<size_t>(s.nlevels) + 1, 0);
  int64_t fwd_slabs = 0, bwd_slabs = 0;
  int n_tickets = 0;
  // chains longer than split_min tiles are cut into groups of about split_len tiles (at most split_maxg groups) on
  // levels of at most split_nfl fronts (swept over eight settings on the benchmark: these stay the best)
  constexpr int split_min = 3, split_len = 3, split_maxg = 12, split_nfl = 128;
  auto front_numbers = [&](WgRec& w, int fr) {
    w.f = fr;
    w.ns = s.f_ns[fr];
    w.bs = s.f_bs[fr];
    w.c0 = s.f_c0[fr];
    w.voff = s.f_voff[fr];
    w.foff = s.f_foff[fr];
    w.toff = toff[fr];
    w.ldt = s.f_ns[fr];
    w.bptr = s.f_bptr[fr];
    w.ftoff = ftoff[fr];
    w.moff = 0;
    const int par = s.f_parent[fr];
    w.pvoff = (par >= 0) ? s.f_voff[par] : -1;
    w.scratch = (child_no[fr] >= kMaxS) ? 1 : 0;
    w.slot = w.scratch ? nslot : child_no[fr];
  };
  auto push_chain = [&](std::vector<WgRec>& out, int64_t& slabs, int fr, int tile, int L, bool split, int flags,
                        bool backward) {
    const int G = (split && L > split_min) ? std::min(split_maxg, (L + split_len - 1) / split_len) : 1;
    int64_t moff = 0;  // the chain's first tile in the fragment-major copy
    if (ff_of[fr] >= 0) {
      const FragFront& r = ffr[ff_of[fr]];
      moff = backward ? r.bm + 256 * frag_bwd_steps(r.nst, r.nbt, r.nko, r.nkb, tile)
                      : r.fm + 256 * frag_fwd_steps(r.nst, r.nko, tile);
    }
    for (int g = 0; g < G; ++g) {
      WgRec w;
      front_numbers(w, fr);
      w.moff = moff;
      w.tile = tile;
      w.s0 = static_cast<int>(static_cast<int64_t>(L) * g / G);
      w.s1 = static_cast<int>(static_cast<int64_t>(L) * (g + 1) / G);
      w.slab = static_cast<int>(slabs);
      w.cnt = n_tickets;
      w.G = G;
      w.flags = flags | (g << 8);
      out.push_back(w);
    }
    if (G > 1) {
      slabs += G;
      ++n_tickets;
    }
  };
  std::vector<int> h_fwd_nsingle(static_cast<size_t>(s.nlevels), 0), h_bwd_nsingle(static_cast<size_t>(s.nlevels), 0);
  std::vector<WgRec> pre_wg;
  std::vector<int> h_pre_ptr(static_cast<size_t>(s.nlevels) + 1, 0);
  const int pre_min_wg = pre_assembly_min_workgroups();
  std::vector<WgRec> wave_wg;
  std::vector<int> h_wave_ptr(static_cast<size_t>(s.nlevels) + 1, 0);
  for (int l = 0; l < s.nlevels; ++l) {
    std::vector<WgRec> multi, bmulti, pre_lvl;
    const int nfl = s.lvl_ptr[l + 1] - s.lvl_ptr[l];
    const bool sparse_level = nfl < 256;  // few fronts: parallelism has to come from inside the fronts
    const bool split_level = nfl <= split_nfl;   // the join of split chains costs an agent-scope acquire: only where chains are long
    for (int q = s.lvl_ptr[l]; q < s.lvl_ptr[l + 1]; ++q) {
      const int fr = s.lvl_fronts[q];
      const int nst = (s.f_ns[fr] + TW - 1) / TW, nbt = (s.f_bs[fr] + TW - 1) / TW;
      const int kids = has_kids[fr] ? 2 : 0;
      if (nst == 1) {  // single column tile: the right-hand side block is loaded once per workgroup
        {
          WgRec w;  // narrow sweeps: one record per front, waves take the row tiles
          front_numbers(w, fr);
          w.tile = 0;
          w.s0 = 0;
          w.s1 = nst + nbt;
          w.slab = w.cnt = 0;
          w.G = 1;
          w.flags = 1 | kids;
          wave_wg.push_back(w);
        }
        const int per = sparse_level ? 1 : nst + nbt;
        for (int t = 0; t < nst + nbt; t += per) {
          WgRec w;
          front_numbers(w, fr);
          w.tile = 0;
          w.s0 = t;
          w.s1 = std::min(nst + nbt, t + per);
          w.slab = w.cnt = 0;
          w.G = 1;
          w.flags = 1 | kids;
          fwd_wg.push_back(w);
        }
      } else {
        for (int t = nst + nbt - 1; t >= 0; --t)  // border tiles (nst products) first, then the own tiles, longest first
          push_chain(multi, fwd_slabs, fr, t, t < nst ? t + 1 : nst, split_level, kids, false);
        for (int t = 0; t < nst; ++t) {
          WgRec w;
          front_numbers(w, fr);
          w.tile = t;
          w.s0 = w.s1 = 0;
          w.slab = w.cnt = 0;
          w.G = 1;
          w.flags = kids;
          pre_lvl.push_back(w);
        }
      }
      const int nbt_b = (s.f_parent[fr] >= 0) ? nbt : 0;
      for (int t = 0; t < nst; ++t)
        // (a single-column-tile front is never split: the wave kernels run it in one piece, and a column's solution
        // must not depend on the width of the sweep it is part of)
        push_chain(nst == 1 ? bwd_wg : bmulti, bwd_slabs, fr, t, nst - t + nbt_b, split_level && nst > 1, 0, true);
    }
    h_fwd_nsingle[l] = static_cast<int>(fwd_wg.size()) - h_fwd_ptr[l];
    h_bwd_nsingle[l] = static_cast<int>(bwd_wg.size()) - h_bwd_ptr[l];
    bwd_wg.insert(bwd_wg.end(), bmulti.begin(), bmulti.end());
    h_wave_ptr[l + 1] = static_cast<int>(wave_wg.size());
    fwd_wg.insert(fwd_wg.end(), multi.begin(), multi.end());
    h_fwd_ptr[l + 1] = static_cast<int>(fwd_wg.size());
    // v1 written once per level where the row-tile workgroups that would each gather it are more than the chip holds at
    // two waves per SIMD (512): the pre-assembled form runs three, and drops an index round from every workgroup.  The
    // shell model's levels of 4000 to 14000: -30 % per launch; the 1 M-dof column's five levels of 552 to 1984: 1.43 ->
    // 1.36 ms per 32-column sweep (four factors in one process, tools/pre_ab_probe.py).  Below that the extra launch
    // costs more than the round it saves
    if (static_cast<int>(multi.size()) >= pre_min_wg)
      pre_wg.insert(pre_wg.end(), pre_lvl.begin(), pre_lvl.end());
    h_pre_ptr[l + 1] = static_cast<int>(pre_wg.size());
    h_bwd_ptr[l + 1] = static_cast<int>(bwd_wg.size());
  }
  const int64_t n_slabs = std::max<int64_t>(1, std::max(fwd_slabs, bwd_slabs));
  const bool has_v1 = !pre_wg.empty();
  if (has_v1) nplanes += 1;
  // rows of the caller's block behind every border entry (backward sweep gathers x there)
  std::vector<int> bout(s.border.size());
  for (size_t e = 0; e < s.border.size(); ++e) bout[e] = s.perm[s.border[e]];
  size_t free_b = 0, total_b = 0;
  EIGD_HIP(hipMemGetInfo(&free_b, &total_b));
  const size_t need = sizeof(double) * (static_cast<size_t>(s.front_doubles) + s.inv_doubles + toff[nf] +
                                        2 * ftoff[nf] + fm_doubles + bm_doubles + (2 * nplanes + 1) * v_rows * KBMAX +
                                        n_slabs * TW * KBMAX) +
                      16 * s.a_src.size() + (size_t(64) << 20);
  if (need > free_b) {
    set_error("factor needs %.2f GiB of device memory, %.2f GiB free", need / 1073741824.0, free_b / 1073741824.0);
    return EIGD_E_HIP;
  }
  eigd_factor* f = new eigd_factor();
  f->ctx = ctx;
  f->sym = &h->s;
  f->h_fwd_ptr = h_fwd_ptr;
  f->h_fwd_nsingle = h_fwd_nsingle;
  f->h_pre_ptr = h_pre_ptr;
  f->has_v1 = has_v1;
  f->h_lvl_leaf.assign(static_cast<size_t>(s.nlevels), 1);
  for (int q = 0; q < nf; ++q)
    if (nchild[q] > 0) f->h_lvl_leaf[s.f_level[q]] = 0;
  f->h_lvl_two.assign(static_cast<size_t>(s.nlevels), 1);
  for (int q = 0; q < nf; ++q)
    if (nchild[q] > 2) f->h_lvl_two[s.f_level[q]] = 0;
  f->h_bwd_nsingle = h_bwd_nsingle;
  f->h_bwd_mxbs.assign(static_cast<size_t>(s.nlevels), 0);
  for (int q = 0; q < nf; ++q)
    if (s.f_ns[q] > TW && s.f_parent[q] >= 0)
      f->h_bwd_mxbs[s.f_level[q]] = std::max<int>(f->h_bwd_mxbs[s.f_level[q]], s.f_bs[q]);
  f->h_fwd_kd.assign(static_cast<size_t>(s.nlevels), 8);
  f->h_bwd_kd.assign(static_cast<size_t>(s.nlevels), 8);
  {
    // levels for the wave-per-block kernels: every single-column-tile front has at most thin_fwd / thin_bwd own
    // columns and (backward) a border of <= 320
    constexpr int thin_fwd = TW, thin_bwd = TW;
    std::vector<int> mxns(static_cast<size_t>(s.nlevels), 0), mxbs(static_cast<size_t>(s.nlevels), 0);
    for (int q = 0; q < nf; ++q) {
      if (s.f_ns[q] > TW) continue;
      mxns[s.f_level[q]] = std::max<int>(mxns[s.f_level[q]], s.f_ns[q]);
      mxbs[s.f_level[q]] = std::max<int>(mxbs[s.f_level[q]], s.f_bs[q]);
    }
    f->h_thin_fwd.assign(static_cast<size_t>(s.nlevels), 0);
    f->h_thin_bwd.assign(static_cast<size_t>(s.nlevels), 0);
    for (int l = 0; l < s.nlevels; ++l) {
      // (fronts with carry planes and 33 to 48 own columns: 12 K-steps -- the buffer-access kernels, several waves per front)
      const int nks = (mxns[l] <= 16) ? 4 : (mxns[l] <= 32) ? 8 : (mxns[l] <= 48 && !f->h_lvl_leaf[l]) ? 12 : 16;
      // leaf level: K-steps of the forward kernel cut to the widest front (12 / 14 instead of 16: fewer MFMAs on zeros)
      const int nks_leaf = (mxns[l] > 32 && mxns[l] <= 48) ? 12 : (mxns[l] > 48 && mxns[l] <= 56) ? 14 : nks;
      // (with carries to gather, the 16-step forward variant needs 244 VGPRs: those levels stay with the tile kernels)
      // levels of binary fronts with up to 48 own columns go to the buffer-access thin
      // kernels too, with two or four waves per front where the level has fewer than 2048 fronts (at C3 the two levels of
      // 42-column fronts under the multi-tile ones: 61 + 44 us in the tile kernel)
      const int kids_cap = f->h_lvl_two[l] ? 48 : 32;
      if (mxns[l] > 0 && mxns[l] <= (f->h_lvl_leaf[l] ? thin_fwd : std::min(thin_fwd, kids_cap)))
        f->h_thin_fwd[l] = f->h_lvl_leaf[l] ? nks_leaf : nks;
      // (backward: one wave per front -- with more than 32 own columns only where the level has fronts enough to
      // fill the chip that way)
      const int nfl = h_wave_ptr[l + 1] - h_wave_ptr[l];
      const int nks_b = (mxns[l] <= 16) ? 4 : (mxns[l] <= 32) ? 8 : 16;  // (the backward kernels know 4, 8 and 16 K-steps)
      if (mxns[l] > 0 && mxns[l] <= thin_bwd && mxbs[l] <= 320 && (nks_b < 16 || nfl >= 1024)) f->h_thin_bwd[l] = nks_b;
    }
  }
  for (int q = 0; q < nf; ++q) {
    if (s.f_ns[q] > TW) continue;
    const int l = s.f_level[q];
    const int bsq = (s.f_parent[q] >= 0) ? std::min<int>(TW, s.f_bs[q]) : 0;
    f->h_fwd_kd[l] = std::max(f->h_fwd_kd[l], (s.f_ns[q] + 7) & ~7);
    f->h_bwd_kd[l] = std::max(f->h_bwd_kd[l], (std::max<int>(s.f_ns[q], bsq) + 7) & ~7);
  }
  f->h_wave_ptr = h_wave_ptr;
  f->ft_doubles = ftoff[nf];
  f->fm_doubles = fm_doubles;
  f->bm_doubles = bm_doubles;
  f->n_ff = static_cast<int>(ffr.size());
  f->n_mt = mt_pref.back();
  f->n_tr = tr_pref[nf];
  f->h_bwd_ptr = h_bwd_ptr;
  f->ov_lvl_ptr = ov_lvl_ptr;
  f->t_doubles = toff[nf];
  f->v_rows = v_rows;
  f->nslot = nslot;
  f->nplanes = nplanes;
  f->n_slabs = n_slabs;
  f->n_tickets = std::max(1, n_tickets);
  f->n_tri = tri_pref[nf];
  f->n_m21 = m_pref[nf];

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
  UP(d_fwd_wg, fwd_wg)
  UP(d_bwd_wg, bwd_wg)
  UP(d_wave_wg, wave_wg)
  UP(d_pre_wg, pre_wg)
  UP(d_ftoff, ftoff)
  UP(d_tr_pref, tr_pref)
  UP(d_mt_pref, mt_pref)
  if (!ffr.empty()) {
    UP(d_ff, ffr)
  }
  UP(d_bout, bout)
  UP(d_tri_pref, tri_pref)
  UP(d_m_pref, m_pref)
  UP(d_toff, toff)
  UP(d_ov_dst, ov_dst)
  UP(d_ov_ptr, ov_ptr)
  UP(d_ov_src, ov_src)
  UP(d_cs_child, s.cs_child)
  UP(d_a_src, s.a_src)
  UP(d_a_dst, s.a_dst)
  UP(d_v_src, s.v_src)
  {
    // which carry planes hold a contribution on which row: child number q of a front writes plane q at the parent's
    // rows rel(border); the summed surplus children arrive in plane nslot - 1
    std::vector<int> cmask(static_cast<size_t>(s.sumd), 0);
    for (int c = 0; c < nf; ++c) {
      const int p = s.f_parent[c];
      if (p < 0 || s.f_bs[c] == 0) continue;
      const int bit = (child_no[c] >= kMaxS) ? nslot - 1 : child_no[c];
      const int64_t b0 = s.f_bptr[c];
      for (int i = 0; i < s.f_bs[c]; ++i) cmask[static_cast<size_t>(s.f_voff[p] + s.rel[b0 + i])] |= 1 << bit;
    }
    UP(d_cmask, cmask)
  }
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
  rc = dmalloc(&f->d_T, static_cast<size_t>(f->t_doubles));
  rc = dmalloc(&f->d_aux, 2);
  rc = dmalloc(&f->d_red, 512);
  rc = dmalloc(&f->d_Ft, static_cast<size_t>(f->ft_doubles));
  rc = dmalloc(&f->d_Fb, static_cast<size_t>(f->ft_doubles));
  rc = dmalloc(&f->d_Fm, static_cast<size_t>(std::max<int64_t>(f->fm_doubles, 1)));
  rc = dmalloc(&f->d_Bm, static_cast<size_t>(std::max<int64_t>(f->bm_doubles, 1)));
  rc = dmalloc(&f->d_P, static_cast<size_t>(n_slabs) * TW * KBMAX);
  if (rc == EIGD_OK) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->d_tickets), sizeof(int) * f->n_tickets);
    if (e != hipSuccess) {
      set_error("hipMalloc failed: %s", hipGetErrorString(e));
      rc = EIGD_E_HIP;
    } else {
      e = hipMemset(f->d_tickets, 0, sizeof(int) * f->n_tickets);
    }
  }
  rc = dmalloc(&f->d_V, static_cast<size_t>(nplanes) * v_rows * kPlaneCols);
  rc = dmalloc(&f->d_Y, static_cast<size_t>(s.sumd) * KBMAX);
  rc = dmalloc(&f->d_sgn, static_cast<size_t>(s.n));
  if (rc == EIGD_OK) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->d_flag), 3 * sizeof(int));
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
  EIGD_HIP(hipMemsetAsync(f->d_T, 0, sizeof(double) * std::max<int64_t>(f->t_doubles, 1), ctx->stream));  // upper triangles stay zero
  // carry planes: entries no child writes must read as zero, in every sweep
  EIGD_HIP(hipMemsetAsync(f->d_V, 0, sizeof(double) * std::max<int64_t>(static_cast<int64_t>(nplanes) * v_rows * kPlaneCols, 1),
                          ctx->stream));
  {
    double aux[2] = {0.0, 0.0};
    const int m1 = -1;
    std::memcpy(&aux[1], &m1, sizeof(int));
    EIGD_HIP(hipMemcpy(f->d_aux, aux, sizeof(aux), hipMemcpyHostToDevice));
  }
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

int eigd_factor_refactor_dev(eigd_factor* f, const double* dvals) {
  EIGD_REQUIRE(f && dvals, "null argument");
  return numeric(f, dvals, true);
}

int eigd_factor_solve(eigd_factor* f, double* dX, int ldx, int k, double alpha) {
  return eigd_factor_solve_to(f, dX, ldx, dX, ldx, k, alpha);
}

static int solve_blocks(eigd_factor* f, hipStream_t st, double* wV, double* wY, double* wP, int* wT, const double* dIn,
                        int ldin, double* dOut, int ldout, int k, double alpha) {
  EIGD_REQUIRE(f && dIn && dOut, "null argument");
  EIGD_REQUIRE(k >= 1 && ldin >= k && ldout >= k, "bad block shape k=%d ldin=%d ldout=%d", k, ldin, ldout);
  for (int c0 = 0; c0 < k; c0 += KBMAX) {
    const int kb = std::min(KBMAX, k - c0);
    int rc;
    // 5 to 8 columns go through the 16-column kernels, whose single-tile levels are MFMA wave kernels (1.26 -> 1.09 ms)
    if (kb <= 4)
      rc = sweep<1>(f, st, wV, wY, wP, wT, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    else if (kb <= 16)
      rc = sweep<4>(f, st, wV, wY, wP, wT, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    else
      rc = sweep<8>(f, st, wV, wY, wP, wT, dIn + c0, ldin, dOut + c0, ldout, kb, alpha);
    if (rc != EIGD_OK) return rc;
  }
  return EIGD_OK;
}

int eigd_factor_solve_to(eigd_factor* f, const double* dIn, int ldin, double* dOut, int ldout, int k, double alpha) {
  EIGD_REQUIRE(f, "null argument");
  return solve_blocks(f, f->ctx->stream, f->d_V, f->d_Y, f->d_P, f->d_tickets, dIn, ldin, dOut, ldout, k, alpha);
}

// A lane = a second set of sweep workspaces bound to another context (stream) of the same device: sweeps of
// different lanes run concurrently on the one factor (independent mode groups overlap each other's latency).
struct eigd_lane {
  eigd_factor* f = nullptr;
  eigd_ctx* ctx = nullptr;
  double *V = nullptr, *Y = nullptr, *P = nullptr;
  int* tickets = nullptr;
};

int eigd_factor_lane_create(eigd_factor* f, eigd_ctx* ctx, eigd_lane** out) {
  EIGD_REQUIRE(f && ctx && out, "null argument");
  EIGD_REQUIRE(ctx->device == f->ctx->device, "lane context must live on the factor's device");
  *out = nullptr;
  const Symbolic& s = *f->sym;
  eigd_lane* l = new eigd_lane();
  l->f = f;
  l->ctx = ctx;
  const size_t vb = sizeof(double) * std::max<size_t>(static_cast<size_t>(f->nplanes) * f->v_rows * kPlaneCols, 1);
  const size_t yb = sizeof(double) * std::max<size_t>(static_cast<size_t>(s.sumd) * KBMAX, 1);
  hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&l->V), vb);
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&l->Y), yb);
  hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&l->P), sizeof(double) * static_cast<size_t>(f->n_slabs) * TW * KBMAX);
  hipError_t e4 = hipMalloc(reinterpret_cast<void**>(&l->tickets), sizeof(int) * f->n_tickets);
  if (e4 == hipSuccess) e4 = hipMemset(l->tickets, 0, sizeof(int) * f->n_tickets);
  if (e1 == hipSuccess) e1 = hipMemset(l->V, 0, vb);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
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
  if (l->tickets) (void)hipFree(l->tickets);
  delete l;
  return EIGD_OK;
}

int eigd_factor_lane_solve_to(eigd_lane* l, const double* dIn, int ldin, double* dOut, int ldout, int k, double alpha) {
  EIGD_REQUIRE(l, "null argument");
  return solve_blocks(l->f, l->ctx->stream, l->V, l->Y, l->P, l->tickets, dIn, ldin, dOut, ldout, k, alpha);
}

int eigd_factor_stats(eigd_factor* f, double* out, int nout) {
  EIGD_REQUIRE(f && out && nout >= 1, "null argument");
  const double v[7] = {static_cast<double>(f->sym->nnzL), static_cast<double>(f->bytes), f->sym->flops,
                       static_cast<double>(f->sym->nfronts), static_cast<double>(f->n_negative),
                       static_cast<double>(f->n_perturbed), static_cast<double>(f->nplanes)};
  for (int i = 0; i < nout && i < 7; ++i) out[i] = v[i];
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
