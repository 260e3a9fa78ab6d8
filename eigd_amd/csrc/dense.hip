// Tall-skinny panel kernels for gfx950: the BLAS-1/2/3 work numpy does for the reference
// (inner products, Gram-Schmidt sweeps, oblique projections, V@Y / Z@y products).
//
// All blocks are n x k row-major (numpy C order), n ~ 1e6, k <= 64.  Every kernel is a
// single streaming pass over its operands (HBM bound); cross-workgroup sums go through a
// partial-sum slab that a second small kernel adds in a FIXED order, so results do not
// depend on scheduling (run-to-run bitwise reproducible for a given block shape).
#include <algorithm>

#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <type_traits>

#include "common.h"

namespace eigd {

constexpr int kRB = 64;        // rows staged per LDS tile in the gemm kernels
constexpr int kMaxBlocks = 1024;

static inline int grid_for_rows(int n, int rows_per_block) {
  int64_t need = (static_cast<int64_t>(n) + rows_per_block - 1) / rows_per_block;
  return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(need, kMaxBlocks)));
}

// out[o] = sum over g of partial[g * nout + o]: one wave per output, lane l adds g = l, l+64, ...
// in ascending order, then a fixed shuffle tree -- the same association for every launch shape.
// gate (optional): a device word written earlier in the stream; zero = the launch that produced the partials was
// skipped, out keeps what it holds
__global__ __launch_bounds__(kThreads) void reduce_partials_kernel(const double* __restrict__ partial, int nblocks,
                                                                  int nout, double* __restrict__ out,
                                                                  const int* __restrict__ gate = nullptr) {
  if (gate != nullptr && *gate == 0) return;
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (o >= nout) return;
  double s = 0.0;
  int g = lane;
  for (; g + 7 * 64 < nblocks; g += 8 * 64) {  // eight loads in flight, added in the same ascending order
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = partial[static_cast<int64_t>(g + 64 * q) * nout + o];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; g < nblocks; g += 64) s += partial[static_cast<int64_t>(g) * nout + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[o] = s;
}

// ---------------------------------------------------------------------------
// column-wise dot products
// ---------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(kThreads) void coldot_kernel(int n, int k, const double* __restrict__ X, int ldx,
                                                         const double* __restrict__ Y, int ldy,
                                                         double* __restrict__ partial) {
  constexpr int RP = kThreads / KP;
  __shared__ double red[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double s = 0.0;
  if (c < k)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP)
      s += X[r * ldx + c] * Y[r * ldy + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rr == 0 && c < k) {
    double t = 0.0;
    for (int q = 0; q < RP; ++q) t += red[q * KP + c];
    partial[static_cast<int64_t>(blockIdx.x) * k + c] = t;
  }
}

// ---------------------------------------------------------------------------
// column-wise dot products accumulated in twice the working precision (Ogita-Rump-Oishi "Dot2": error-free
// product by FMA, error-free sum, the errors carried in a second double).  For the handful of entries of
// G = -Phi^T Phib that belong to numerically repeated eigenvalue pairs: xi, eta divide their DIFFERENCE by the gap
// (reference eigenvector_derivatives.py:373-383), so the rounding of an n-term dot product shows up 1/gap times larger.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void dd_two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

__device__ __forceinline__ void dd_add(double& hi, double& lo, double bh, double bl) {
  double s, e;
  dd_two_sum(hi, bh, s, e);
  e += lo + bl;
  hi = s + e;
  lo = e - (hi - s);
}

template <int KP>
__global__ __launch_bounds__(kThreads) void coldot_dd_kernel(int n, int k, const double* __restrict__ X, int ldx,
                                                            const double* __restrict__ Y, int ldy,
                                                            double* __restrict__ partial /* [block][2][k] */) {
  constexpr int RP = kThreads / KP;
  __shared__ double red_hi[kThreads], red_lo[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double hi = 0.0, lo = 0.0;
  if (c < k)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
      const double x = X[r * ldx + c], y = Y[r * ldy + c];
      const double p = x * y;
      const double pe = __fma_rn(x, y, -p);     // x*y = p + pe exactly
      double s, e;
      dd_two_sum(hi, p, s, e);                  // hi + p = s + e exactly
      hi = s;
      lo += e + pe;
    }
  red_hi[threadIdx.x] = hi;
  red_lo[threadIdx.x] = lo;
  __syncthreads();
  if (rr == 0 && c < k) {
    double th = 0.0, tl = 0.0;
    for (int q = 0; q < RP; ++q) dd_add(th, tl, red_hi[q * KP + c], red_lo[q * KP + c]);
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * k + c] = th;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * k + c] = tl;
  }
}

__global__ void reduce_dd_kernel(const double* __restrict__ partial, int nblocks, int k, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  double th = 0.0, tl = 0.0;
  for (int b = 0; b < nblocks; ++b)
    dd_add(th, tl, partial[(static_cast<int64_t>(b) * 2 + 0) * k + c], partial[(static_cast<int64_t>(b) * 2 + 1) * k + c]);
  out[c] = th;
  out[k + c] = tl;
}

// ---------------------------------------------------------------------------
// stack_dot: H[j][c] = sum_r S_j[r][c] * T[r][c] for JB slabs per launch
// ---------------------------------------------------------------------------
template <int KP, int JB>
__global__ __launch_bounds__(kThreads) void stack_dot_kernel(int n, int k, int nj, const double* __restrict__ S,
                                                            int64_t slab, int lds, const double* __restrict__ T,
                                                            int ldt, double* __restrict__ partial, int kslab) {
  constexpr int RP = kThreads / KP;
  __shared__ double red[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double acc[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) acc[j] = 0.0;
  if (c < k)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
      const double t = T[r * ldt + c];
      // uniform slab base + 32-bit lane offset (n * lds < 2^31); kslab < k: T is a pair of blocks [T1 | T2] that both meet
      // the slabs' columns 0..kslab-1 (column c of T pairs with slab column c - kslab past the first block)
      const unsigned off = static_cast<unsigned>(r * lds + (c >= kslab ? c - kslab : c));
      double sv[JB];
#pragma unroll
      for (int j = 0; j < JB; ++j) sv[j] = (j < nj) ? (S + j * slab)[off] : 0.0;
#pragma unroll
      for (int j = 0; j < JB; ++j) acc[j] += sv[j] * t;
    }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    red[threadIdx.x] = acc[j];
    __syncthreads();
    if (rr == 0 && c < k && j < nj) {
      double t = 0.0;
      for (int q = 0; q < RP; ++q) t += red[q * KP + c];
      partial[(static_cast<int64_t>(blockIdx.x) * nj + j) * k + c] = t;
    }
    __syncthreads();
  }
}

// masked lanes load this word (address select) instead of branching around the load: loads in divergent branches
// make the compiler wait for ALL outstanding memory operations (vmcnt(0)) at every later use -- including every
// store, which then run one at a time
__device__ const double g_zero_word = 0.0;

// T[r][c] += alpha * sum_j S_j[r][c] * H[j][c]   (H on the device, ns x k)
template <int KP>
__global__ __launch_bounds__(kThreads) void stack_axpy_kernel(int n, int k, int ns, const double* __restrict__ S,
                                                             int64_t slab, int lds, const double* __restrict__ H,
                                                             double* __restrict__ T, int ldt, double alpha, int kslab) {
  constexpr int RP = kThreads / KP;
  extern __shared__ double Hs[];  // ns * k
  for (int q = threadIdx.x; q < ns * k; q += kThreads) Hs[q] = H[q];
  __syncthreads();
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  if (c >= k) return;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
    const double* sp = S + r * lds + (c >= kslab ? c - kslab : c);
    double s = 0.0;
    int j = 0;
    for (; j + 8 <= ns; j += 8) {
      double sv[8], h[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {  // a zero coefficient skips its slab entry: never-written columns are not read at all
        h[q] = Hs[(j + q) * k + c];
        sv[q] = *((h[q] != 0.0) ? sp + (j + q) * slab : &g_zero_word);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) s += sv[q] * h[q];
    }
    for (; j < ns; ++j) {
      const double h = Hs[j * k + c];
      s += (h != 0.0) ? sp[j * slab] * h : 0.0;
    }
    T[r * ldt + c] += alpha * s;
  }
}

// Fused middle section of CGS2:  T <- T + alpha * sum_j S_j H1[j]  and, on the updated T,
// H2[j] = S_j . T  -- one pass over the slab stack (its values stay in registers between the two
// uses) instead of an axpy pass plus a dot pass.  ns <= JMAX <= 32.
template <int KP, int JMAX>
__global__ __launch_bounds__(kThreads) void stack_axpy_dot_kernel(int n, int k, int ns, const double* __restrict__ S,
                                                                 int64_t slab, int lds, const double* __restrict__ H1,
                                                                 double* __restrict__ T, int ldt, double alpha,
                                                                 double* __restrict__ partial, int kslab) {
  constexpr int RP = kThreads / KP;
  extern __shared__ double Hs[];  // ns * k
  __shared__ double red[kThreads];
  for (int q = threadIdx.x; q < ns * k; q += kThreads) Hs[q] = H1[q];
  __syncthreads();
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double acc[JMAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) acc[j] = 0.0;
  if (c < k) {
    double hc[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) hc[j] = (j < ns) ? alpha * Hs[j * k + c] : 0.0;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
      // uniform slab base + one 32-bit lane offset (the host checks n * lds < 2^31): no 64-bit address per slab in VGPRs
      const unsigned off = static_cast<unsigned>(r * lds + (c >= kslab ? c - kslab : c));
      double sv[JMAX];
#pragma unroll
      for (int j = 0; j < JMAX; ++j) sv[j] = (j < ns) ? (S + j * slab)[off] : 0.0;
      double t = T[r * ldt + c];
#pragma unroll
      for (int j = 0; j < JMAX; ++j) t += sv[j] * hc[j];
      T[r * ldt + c] = t;
#pragma unroll
      for (int j = 0; j < JMAX; ++j) acc[j] += sv[j] * t;
    }
  }
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    if (j < ns) {  // ns is uniform: no divergence around the barriers
      red[threadIdx.x] = acc[j];
      __syncthreads();
      if (rr == 0 && c < k) {
        double t = 0.0;
        for (int q = 0; q < RP; ++q) t += red[q * KP + c];
        partial[(static_cast<int64_t>(blockIdx.x) * ns + j) * k + c] = t;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// lincomb
// ---------------------------------------------------------------------------
struct LinArgs {
  const double* x[4];
  int ld[4];
  double coef[4][kMaxK];
};

template <int KP>
__global__ __launch_bounds__(kThreads) void lincomb_kernel(int n, int k, double* out, int ldo, int nterms,
                                                          LinArgs a) {
  constexpr int RP = kThreads / KP;
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  if (c >= k) return;
  double cf[4];
  for (int t = 0; t < 4; ++t) cf[t] = (t < nterms) ? a.coef[t][c] : 0.0;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
    double s = 0.0;
    for (int t = 0; t < nterms; ++t) s += cf[t] * a.x[t][r * a.ld[t] + c];
    out[r * ldo + c] = s;
  }
}

struct SkipMask {
  unsigned char skip[kMaxK];
};

// out[r][c] = x[r][c] / sqrt(norm2[c]); columns with skip[c] != 0 or norm2[c] == 0 become zero.  norm2 lives on the
// device (the column norms never visit the host on this path).
template <int KP>
__global__ __launch_bounds__(kThreads) void scale_inv_norm_kernel(int n, int k, const double* __restrict__ x, int ldx,
                                                                 double* __restrict__ out, int ldo,
                                                                 const double* __restrict__ norm2, SkipMask m) {
  constexpr int RP = kThreads / KP;
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  if (c >= k) return;
  const double n2 = norm2[c];
  const double sc = (m.skip[c] != 0 || n2 == 0.0) ? 0.0 : 1.0 / sqrt(n2);
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP)
    out[r * ldo + c] = sc * x[r * ldx + c];
}

// Two-step Krylov cycle, the pair [T1 | T2] after its Gram-Schmidt step against the stack and the projection:
//   W1 = T1 / |T1|,   T2 <- T2 - (T1.T2 / |T1|^2) T1   (in place),
// and per column the partial sums of |T2|^2 (new) and of W1 . T2 (new; what one Gram-Schmidt pass inside the pair left).
// |T1|^2 (norm2[c]) and T1.T2 (gamma[c]) live on the device: no host synchronisation between the passes.
template <int KP>
__global__ __launch_bounds__(kThreads) void pair_apply_kernel(int n, int k, double* __restrict__ T, int ldt,
                                                             const double* __restrict__ norm2,
                                                             const double* __restrict__ gamma, double* __restrict__ W1,
                                                             int ldw1, SkipMask m, double* __restrict__ partial) {
  constexpr int RP = kThreads / KP;
  __shared__ double red[kThreads];
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  double s2 = 0.0, sd = 0.0;
  if (c < k) {
    const double n1 = norm2[c];
    const bool dead = m.skip[c] != 0 || n1 == 0.0;
    const double inv = dead ? 0.0 : 1.0 / sqrt(n1);
    const double coef = dead ? 0.0 : gamma[c] / n1;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP) {
      const double t1 = T[r * ldt + c];
      const double t2 = T[r * ldt + k + c];
      const double w1 = inv * t1;
      const double t2n = dead ? 0.0 : t2 - coef * t1;
      W1[r * ldw1 + c] = w1;
      T[r * ldt + k + c] = t2n;
      s2 += t2n * t2n;
      sd += w1 * t2n;
    }
  }
  for (int q = 0; q < 2; ++q) {
    red[threadIdx.x] = (q == 0) ? s2 : sd;
    __syncthreads();
    if (rr == 0 && c < k) {
      double t = 0.0;
      for (int z = 0; z < RP; ++z) t += red[z * KP + c];
      partial[(static_cast<int64_t>(blockIdx.x) * 2 + q) * k + c] = t;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// C = U^T X  (partials per workgroup), U(r,a) = U[r*rsu + a*csu].
// The ku x kx result is cut into 16 x 16 tiles; a wave accumulates its tiles with
// v_mfma_f64_16x16x4_f64 (K = 4 rows per instruction), lane (i = l&15, k = l>>4) feeding
// A[i][k] = U(r+k, a0+i) and B[k][j] = X(r+k, b0+j) straight from global memory -- sixteen row
// quads are in flight per wave, no LDS staging.  This is the dense projection Phi^T X / V^T Phib.
// ---------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

// rows per workgroup step of gemm_tn: few loads per wave and step keep the kernel light (many waves per CU)
constexpr int kTnRows = 16;

template <int TPW>  // tiles per wave: 1 (at most four 16 x 16 tiles in all) or 4
__global__ __launch_bounds__(kThreads) void gemm_tn_kernel(int n, int ku, int kx, const double* __restrict__ U,
                                                          int64_t rsu, int64_t csu, const double* __restrict__ X,
                                                          int ldx, double* __restrict__ partial) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int nta = (ku + 15) >> 4, ntb = (kx + 15) >> 4;
  const int ntiles = nta * ntb;  // <= 16
  // fewer than four tiles (narrow X): the waves that would idle take a share of the rows of a tile instead
  // (rsplit waves per tile, quad q of the row block goes to share q % rsplit); shares are added through LDS
  const int rsplit = (TPW == 1 && ntiles == 1) ? 4 : (TPW == 1 && ntiles == 2) ? 2 : 1;
  const int share = (rsplit > 1) ? wave / ntiles : 0;
  double4_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  auto rows_of_share = [&](auto RS, int64_t base, int tile, double4_t& c) {
    constexpr int rs = decltype(RS)::value;
    const int ta = tile / ntb, tb = tile - ta * ntb;
    const int a = ta * 16 + li, b = tb * 16 + li;
    const bool oka = a < ku, okb = b < kx;
    const double* up = U + a * csu;
    const double* xp = X + b;
    double av[kTnRows / 4 / rs], bv[kTnRows / 4 / rs];
#pragma unroll
    for (int q = 0; q < kTnRows / 4 / rs; ++q) {
      const int64_t r = base + (q * rs + share) * 4 + lk;
      const bool okr = r < n;
      av[q] = (okr && oka) ? up[r * rsu] : 0.0;
      bv[q] = (okr && okb) ? xp[r * ldx] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < kTnRows / 4 / rs; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], c, 0, 0, 0);
  };
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kTnRows; base < n; base += static_cast<int64_t>(gridDim.x) * kTnRows) {
    if constexpr (TPW == 1) {
      if (rsplit == 4)
        rows_of_share(std::integral_constant<int, 4>{}, base, 0, acc[0]);
      else if (rsplit == 2)
        rows_of_share(std::integral_constant<int, 2>{}, base, wave % 2, acc[0]);
      else if (wave < ntiles)
        rows_of_share(std::integral_constant<int, 1>{}, base, wave, acc[0]);
    } else {
#pragma unroll 1
      for (int t = 0; t < 4; ++t) {
        const int tile = wave + 4 * t;
        if (tile >= ntiles) break;
        double4_t c = (t == 0) ? acc[0] : (t == 1) ? acc[1 % TPW] : (t == 2) ? acc[2 % TPW] : acc[3 % TPW];
        rows_of_share(std::integral_constant<int, 1>{}, base, tile, c);
        if (t == 0)
          acc[0] = c;
        else if (t == 1)
          acc[1 % TPW] = c;
        else if (t == 2)
          acc[2 % TPW] = c;
        else
          acc[3 % TPW] = c;
      }
    }
  }
  double* p = partial + static_cast<int64_t>(blockIdx.x) * ku * kx;
  if (rsplit > 1) {
    __shared__ double red[4][4][64];  // [wave][reg][lane]
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[wave][reg][lane] = acc[0][reg];
    __syncthreads();
    if (share != 0) return;
    const int tile = wave;  // waves 0 .. ntiles-1 hold share 0
    const int ta = tile / ntb, tb = tile - ta * ntb;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      double sum = 0.0;
      for (int sh = 0; sh < rsplit; ++sh) sum += red[tile + sh * ntiles][reg][lane];
      const int a = ta * 16 + lk + 4 * reg, b = tb * 16 + li;
      if (a < ku && b < kx) p[a * kx + b] = sum;
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave + 4 * t;
    if (tile >= ntiles) break;
    const int ta = tile / ntb, tb = tile - ta * ntb;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int a = ta * 16 + lk + 4 * reg, b = tb * 16 + li;  // f64 MFMA result map: row = (l>>4) + 4*reg, col = l&15
      if (a < ku && b < kx) p[a * kx + b] = acc[t][reg];
    }
  }
}

// The same product for WIDE results (more than four 16 x 16 tiles: the projection coefficients Phi_D^T [T1 | T2], 63 x 64)
// with row-major U: in the direct form above every wave reads the fragments of its own tiles from global memory, so
// with sixteen tiles each operand row is requested four times per workgroup (the rows are the whole traffic here:
// 1 GB per projection).  Here a chunk of 32 rows of U and of X is staged in LDS once (coalesced, the next chunk in
// registers while this one multiplies) and the four waves take their fragments from there.
constexpr int kTsElems = 32 * kMaxK;  // doubles of one operand per chunk: 32 rows of 64 columns, or 64 rows of 32 columns

// xnorm_part (optional): the squared column norms of X ride along, one partial sum per workgroup and column
// (xnorm_part[blockIdx.x * kx + b]; fixed order: the lane's chunks, then the lanes of a column)
// KW: columns of the staging layout, 64 or -- for ku, kx <= 32 -- 32: every lane loads (with 64 columns and 32 x 32
// operands half of them would ask for nothing) and a chunk holds 64 rows, twice the bytes per pair of barriers.
template <int KW>
__global__ __launch_bounds__(kThreads) void gemm_tn_staged_kernel(int n, int ku, int kx, const double* __restrict__ U,
                                                                 int64_t rsu, const double* __restrict__ X, int ldx,
                                                                 double* __restrict__ partial,
                                                                 double* __restrict__ xnorm_part) {
  constexpr int ROWS = kTsElems / KW;        // rows per chunk
  constexpr int RPT = kThreads / KW;         // rows per trip of the lanes
  __shared__ double Us[kTsElems];
  __shared__ double Xs[kTsElems];
  double xsq = 0.0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int nta = (ku + 15) >> 4, ntb = (kx + 15) >> 4;
  const int ntiles = nta * ntb;  // <= 16
  constexpr int IT = kTsElems / kThreads;  // 8 elements of each operand per lane and chunk
  const int col = tid % KW, row0 = tid / KW;  // element it of a lane: row row0 + RPT it, column col
  double4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  double ru[IT], rx[IT];
  auto fetch = [&](int64_t base) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int64_t r = base + row0 + RPT * it;
      ru[it] = *((r < n && col < ku) ? U + r * rsu + col : &g_zero_word);
      rx[it] = *((r < n && col < kx) ? X + r * ldx + col : &g_zero_word);
    }
  };
  const int64_t stride = static_cast<int64_t>(gridDim.x) * ROWS;
  int64_t base = static_cast<int64_t>(blockIdx.x) * ROWS;
  if (base < n) fetch(base);
  for (; base < n; base += stride) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      Us[(row0 + RPT * it) * KW + col] = ru[it];
      Xs[(row0 + RPT * it) * KW + col] = rx[it];
      xsq += rx[it] * rx[it];
    }
    __syncthreads();
    if (base + stride < n) fetch(base + stride);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tile = wave + 4 * t;
      if (tile < ntiles) {
        const int ta = tile / ntb, tb = tile - ta * ntb;
        double4_t c = acc[t];
#pragma unroll
        for (int q = 0; q < ROWS / 4; ++q)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(Us[(4 * q + lk) * KW + 16 * ta + li],
                                                   Xs[(4 * q + lk) * KW + 16 * tb + li], c, 0, 0, 0);
        acc[t] = c;
      }
    }
  }
  double* p = partial + static_cast<int64_t>(blockIdx.x) * ku * kx;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tile = wave + 4 * t;
    if (tile >= ntiles) break;
    const int ta = tile / ntb, tb = tile - ta * ntb;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int a = ta * 16 + lk + 4 * reg, b = tb * 16 + li;  // f64 MFMA result map: row = (l>>4) + 4*reg, col = l&15
      if (a < ku && b < kx) p[a * kx + b] = acc[t][reg];
    }
  }
  if (xnorm_part != nullptr) {
    __syncthreads();
    Xs[row0 * KW + col] = xsq;  // (the RPT lanes of a column, in order)
    __syncthreads();
    if (tid < kx) {
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) v += Xs[q * KW + tid];
      xnorm_part[static_cast<int64_t>(blockIdx.x) * kx + tid] = v;
    }
  }
}

// gate[0] = 1 if some coefficient of a projection matters, |C[a][b]| > tol * sqrt(norm2[b]), else 0: the update X -= U C
// that follows is skipped then (see eigd_project_norm2).  norm2 = the squared column norms of X: the Euclidean ones
// from the same pass, or -- eigd_project_to with dNorm2 -- the squared B-norms x_b^T B x_b the caller formed, which makes
// the test the relative B-orthogonality of the block against the panel whatever the scale of B.
__global__ __launch_bounds__(kThreads) void project_decide_kernel(const double* __restrict__ C, int ku, int kx,
                                                                 const double* __restrict__ norm2, double tol,
                                                                 int* __restrict__ gate, int* __restrict__ stats,
                                                                 double* __restrict__ flag_out = nullptr) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  int mine = 0;
  for (int q = threadIdx.x; q < ku * kx; q += kThreads) {
    const double c = C[q], lim = tol * tol * norm2[q % kx];
    if (!(c * c <= lim)) mine = 1;  // (a NaN anywhere, or a norm that rounding made negative, keeps the update)
  }
  if (mine) any = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    gate[0] = any;
    if (stats != nullptr) {
      stats[0] += 1;  // (one workgroup, launches of a stream run in order: plain adds)
      stats[1] += any;
    }
    if (flag_out != nullptr) flag_out[0] = any ? 1.0 : 0.0;
  }
}

// SVQB orthonormalising transform of a block from its Gram matrix, on the device (one wave; the host form is
// lanczos.py:_svqb): G = X^T B X (p x p, p <= 32) is symmetrised and scaled to unit diagonal, A = D^-1 G D^-1 =
// U diag(w) U^T by cyclic Jacobi rotations (small eigenvalues of a scaled positive matrix come out with high RELATIVE
// accuracy, which is what the inverse square roots need), eigenvalues ascending as LAPACK orders them;
// Tr = D^-1 U w^-1/2 makes X Tr B-orthonormal, Cq = w^1/2 U^T D gives X = (X Tr) Cq, Ctot <- Cq Ctot (first: Ctot <- Cq).
// Directions with w <= 1e-28 max(w) carry nothing but rounding: their rows of Cq are zero and flag[0] becomes 1.0.
constexpr int kSvqbMax = 32;
__global__ __launch_bounds__(64) void svqb_kernel(const double* __restrict__ G, int p, double* __restrict__ Tr,
                                                  double* __restrict__ Ctot, int first, double* __restrict__ flag) {
  __shared__ double A[kSvqbMax][kSvqbMax + 1], U[kSvqbMax][kSvqbMax + 1], Cq[kSvqbMax][kSvqbMax + 1];
  __shared__ double d[kSvqbMax], wi[kSvqbMax];
  __shared__ int perm[kSvqbMax], bad[kSvqbMax], rotated;
  const int lane = threadIdx.x;
  for (int q = lane; q < p * p; q += 64) A[q / p][q % p] = 0.5 * (G[q] + G[(q % p) * p + q / p]);
  __syncthreads();
  if (lane < p) d[lane] = sqrt(fmax(A[lane][lane], 2.2250738585072014e-308));
  __syncthreads();
  for (int q = lane; q < p * p; q += 64) {
    const int i = q / p, j = q % p;
    A[i][j] = A[i][j] / (d[i] * d[j]);
    U[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double eps = 1.1102230246251565e-16;
  for (int sweep = 0; sweep < 40; ++sweep) {
    if (lane == 0) rotated = 0;
    __syncthreads();
    for (int i = 0; i + 1 < p; ++i)
      for (int j = i + 1; j < p; ++j) {
        const double aij = A[i][j], aii = A[i][i], ajj = A[j][j];  // (the same for every lane: the branch is uniform)
        if (fabs(aij) > eps * sqrt(fabs(aii * ajj))) {
          const double tau = (ajj - aii) / (2.0 * aij);
          const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
          const int k = lane;
          if (k < p) {
            if (k != i && k != j) {
              const double aki = A[k][i], akj = A[k][j];
              const double nki = c * aki - sn * akj, nkj = sn * aki + c * akj;
              A[k][i] = nki;
              A[i][k] = nki;
              A[k][j] = nkj;
              A[j][k] = nkj;
            }
            const double uki = U[k][i], ukj = U[k][j];
            U[k][i] = c * uki - sn * ukj;
            U[k][j] = sn * uki + c * ukj;
          }
          if (lane == 0) {
            A[i][i] = aii - t * aij;
            A[j][j] = ajj + t * aij;
            A[i][j] = 0.0;
            A[j][i] = 0.0;
            rotated = 1;
          }
        }
        __syncthreads();
      }
    if (rotated == 0) break;
    __syncthreads();
  }
  // ascending order (rank by value, ties by index), the noise floor, the two transforms
  if (lane < p) {
    const double w = A[lane][lane];
    int rank = 0;
    double wmax = 2.2250738585072014e-308;
    for (int j = 0; j < p; ++j) {
      const double wj = A[j][j];
      rank += (wj < w || (wj == w && j < lane)) ? 1 : 0;
      wmax = fmax(wmax, wj);
    }
    perm[rank] = lane;
    const int b = !(w > 1e-28 * wmax);
    bad[lane] = b;
    wi[lane] = b ? 1.0 : w;
  }
  __syncthreads();
  for (int q = lane; q < p * p; q += 64) {
    const int i = q / p, r = q % p, e = perm[r];
    const double sq = sqrt(wi[e]);
    Tr[i * p + r] = U[i][e] / sq / d[i];
    Cq[r][i] = bad[e] ? 0.0 : sq * U[i][e] * d[i];
  }
  __syncthreads();
  double out[(kSvqbMax * kSvqbMax + 63) / 64];
  int nq = 0;
  for (int q = lane; q < p * p; q += 64, ++nq) {
    const int r = q / p, j = q % p;
    double v;
    if (first) {
      v = Cq[r][j];
    } else {
      v = 0.0;
      for (int i = 0; i < p; ++i) v += Cq[r][i] * Ctot[i * p + j];
    }
    out[nq] = v;
  }
  __syncthreads();  // (every entry of the old Ctot is read before any is replaced)
  nq = 0;
  for (int q = lane; q < p * p; q += 64, ++nq) Ctot[q] = out[nq];
  if (lane == 0) {
    int any = 0;
    for (int j = 0; j < p; ++j) any |= bad[j];
    if (first)
      flag[0] = any ? 1.0 : 0.0;
    else if (any)
      flag[0] = 1.0;
  }
}

// X[r][b] = beta * X[r][b] + alpha * sum_a U(r,a) C[a][b], C on the device (ku x kx).
// A 64-row chunk of U is staged in LDS with all its loads in flight; wave w owns rows 16w..16w+15 and
// forms each 16 x 16 output tile with v_mfma_f64_16x16x4_f64 (A = U rows from LDS, B = C from LDS with a
// conflict-free row stride), so the LDS traffic per flop is an eighth of a scalar inner loop.
__device__ __forceinline__ int cs_stride(int kx) { return (kx % 32 == 0) ? kx + 16 : kx; }


// Row-major U (csu == 1): every wave works alone on 16-row groups and feeds its MFMAs with fragments of U loaded
// straight from global memory.  Lane (i, g) holds NQ CONSECUTIVE columns of row base + i, U(base + i, NQ g .. NQ g + NQ - 1):
// one contiguous piece of 8 NQ bytes per lane (16-byte loads when the rows are so aligned), every line of U is
// requested by exactly one lane -- with the MFMA's natural operand order (lane (i, g) holding U(., 4 q + g)) the sixteen
// loads of a group touched every line four times, and with thirty-odd waves per CU streaming different rows the later
// requests missed the vector cache: 3.9 TB/s at 64 x 8, against the 6.7 of the transposed product on the same panels.
// K-step q therefore multiplies columns {q, NQ + q, 2 NQ + q, 3 NQ + q}: the rows of C sit in LDS in that order (slot
// 4 q + g holds row NQ g + q, rows beyond ku are zero).  No LDS for U, no barrier after the one that publishes C, few
// registers: many waves per CU keep the stream of U going.  NQ = 8 (ku <= 32) or 16 (ku <= 64) compiled in.
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int NQ>
__global__ __launch_bounds__(kThreads) void gemm_nn_direct_kernel(int n, int ku, int kx, const double* __restrict__ U,
                                                                 int64_t rsu, const double* __restrict__ C,
                                                                 double* __restrict__ X, int ldx, double alpha,
                                                                 double beta, double* __restrict__ normpart,
                                                                 const int* __restrict__ gate) {
  // normpart != nullptr: the squared column norms of the result are left as one partial sum per workgroup and
  // column (normpart[blockIdx.x * kx + b]) -- the norm pass over X that would follow a projection is folded in
  // gate != nullptr: a device word written earlier in the stream; zero = leave X alone
  if (gate != nullptr && *gate == 0) return;
  extern __shared__ double Cs[];  // 4 NQ x cs_stride(kx)
  __shared__ double nred[kThreads / 64][64];
  double nsq[4] = {0.0, 0.0, 0.0, 0.0};
  const int tid = threadIdx.x;
  const int cld = cs_stride(kx);
  for (int q = tid; q < 4 * NQ * kx; q += kThreads) {
    const int slot = q / kx, b = q % kx;
    const int k = NQ * (slot & 3) + (slot >> 2);
    Cs[slot * cld + b] = (k < ku) ? C[k * kx + b] : 0.0;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int ntb = (kx + 15) >> 4;  // <= 4
  const int64_t ngroups = (static_cast<int64_t>(n) + 15) / 16;
  const int64_t gstride = static_cast<int64_t>(gridDim.x) * 4;
  // whole rows of 4 NQ columns, 16-byte aligned: the lane's piece as 16-byte loads (wave-uniform test)
  const bool vec = (ku == 4 * NQ) && ((rsu & 1) == 0) && ((reinterpret_cast<uintptr_t>(U) & 15) == 0);
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave; g < ngroups; g += gstride) {
    const int64_t base = g * 16;
    const int rows = static_cast<int>((n - base) < 16 ? (n - base) : 16);
    const double* up = U + (base + li) * rsu + NQ * lk;
    double a[NQ];
    if (vec && rows == 16) {
      const double2_t* vp = reinterpret_cast<const double2_t*>(up);
#pragma unroll
      for (int q = 0; q < NQ / 2; ++q) {
        const double2_t v = vp[q];
        a[2 * q] = v.x;
        a[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) a[q] = *((li < rows && NQ * lk + q < ku) ? up + q : &g_zero_word);
    }
    double xv[4][4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = lk + 4 * reg, b = tb * 16 + li;
        xv[tb][reg] = *((beta != 0.0 && tb < ntb && r < rows && b < kx) ? X + (base + r) * ldx + b : &g_zero_word);
      }
    double4_t accs[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      accs[tb] = double4_t{0.0, 0.0, 0.0, 0.0};
      if (tb >= ntb) continue;
      const int b = tb * 16 + li;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const double bv = (b < kx) ? Cs[(4 * q + lk) * cld + b] : 0.0;   // (row NQ lk + q of C)
        accs[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bv, accs[tb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one wait, then the stores back to back
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      if (tb >= ntb) break;
      const int b = tb * 16 + li;
      double outv[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        outv[reg] = (beta == 0.0) ? alpha * accs[tb][reg] : beta * xv[tb][reg] + alpha * accs[tb][reg];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = lk + 4 * reg;  // f64 MFMA result map: row = (l>>4) + 4*reg, col = l&15
        if (r < rows && b < kx) {
          X[(base + r) * ldx + b] = outv[reg];
          nsq[tb] += outv[reg] * outv[reg];
        }
      }
    }
  }
  if (normpart != nullptr) {  // fixed order: the lane's groups, its four row quarters, the four waves
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      double v = nsq[tb];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lk == 0) nred[wave][tb * 16 + li] = v;
    }
    __syncthreads();
    if (tid < kx) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; ++w) v += nred[w][tid];
      normpart[static_cast<int64_t>(blockIdx.x) * kx + tid] = v;
    }
  }
}

// Out(:, o0 .. o0 + kx) = sum_p U_p C_p for a basis kept as row-major PANELS of 64 columns (U_p at U + p * pstride, leading
// dimension 64): the restart of the block Lanczos, V <- V S with 168 columns in and 130 out, and V (T Cf) of the first
// guess.  Panel by panel through gemm_nn_direct_kernel that was np launches per 64 output columns, each re-reading and
// re-writing the output (13.5 GB per restart matrix); here the coefficient rows of ALL panels sit in LDS (up to 3 x 64 rows
// of up to 80 + 16 columns: 147 KB, one workgroup of eight waves per CU: 144 registers), a wave streams 16-row groups -- the lane's 16
// consecutive columns of a panel row as 16-byte loads, the next panel's in flight while this one multiplies -- and writes
// the finished rows once: np panels read + kx columns written per launch.  Fragment layout and K order as in
// gemm_nn_direct_kernel<16> (K-step q of a panel multiplies its columns {q, 16 + q, 32 + q, 48 + q}); the sum runs over
// the panels in order.  Output column j lies in panel j / opw of Out (panels ostride apart, rows of `old` doubles: the
// layout of U with opw = old = 64, a plain row-major block with opw >= kx).
constexpr int kPanelW = 64, kPanelsMax = 3, kPanelOutMax = 80, kPanelThreads = 512;

template <int NP>
__global__ __launch_bounds__(kPanelThreads) void gemm_nn_panels_kernel(int n, int ku, int kx, const double* __restrict__ U,
                                                                       int64_t pstride, const double* __restrict__ C, int ldc,
                                                                       int c0, double* __restrict__ Out, int64_t ostride,
                                                                       int o0, int opw, int old) {
  extern __shared__ double Cs[];  // NP x 64 slots x cld
  const int tid = threadIdx.x;
  const int cld = kx + 16;
  // slot 4 q + g of panel p holds row 64 p + 16 g + q of C (rows past ku: zeros -- the columns of the last panel beyond
  // the basis may hold anything, their fragments are masked below as well)
  for (int e = tid; e < NP * kPanelW * cld; e += kPanelThreads) {
    const int slot = e / cld, b = e - slot * cld;
    const int p = slot >> 6, sl = slot & 63;
    const int k = kPanelW * p + 16 * (sl & 3) + (sl >> 2);
    Cs[e] = (k < ku && b < kx) ? C[static_cast<int64_t>(k) * ldc + c0 + b] : 0.0;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int ntb = (kx + 15) >> 4;  // <= 5
  const int64_t ngroups = (static_cast<int64_t>(n) + 15) / 16;
  const int64_t gstride = static_cast<int64_t>(gridDim.x) * (kPanelThreads / 64);
  auto fetch = [&](int p, int64_t base, int rows, double (&a)[16]) {
    const double* up = U + p * pstride + (base + li) * kPanelW + 16 * lk;
    const int left = ku - (kPanelW * p + 16 * lk);  // columns of this lane's piece that belong to the basis
    if (li < rows && left >= 16) {
      const double2_t* vp = reinterpret_cast<const double2_t*>(up);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double2_t v = vp[q];
        a[2 * q] = v.x;
        a[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = *((li < rows && q < left) ? up + q : &g_zero_word);
    }
  };
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * (kPanelThreads / 64) + wave; g < ngroups; g += gstride) {
    const int64_t base = g * 16;
    const int rows = static_cast<int>((n - base) < 16 ? (n - base) : 16);
    double4_t acc[5];
#pragma unroll
    for (int tb = 0; tb < 5; ++tb) acc[tb] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a0[16], a1[16];
    fetch(0, base, rows, a0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      double(&cur)[16] = (p & 1) ? a1 : a0;
      double(&nxt)[16] = (p & 1) ? a0 : a1;
      if (p + 1 < NP) fetch(p + 1, base, rows, nxt);
      // (K-steps outside, output tiles inside: consecutive MFMAs write different accumulators -- no wave waits for its own
      // previous product; the columns of Cs past kx are the 16 spare ones of the row stride: finite, never stored)
      const double* Cp = Cs + p * kPanelW * cld + lk * cld + li;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int tb = 0; tb < 5; ++tb)
          if (tb < ntb) acc[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[q], Cp[4 * q * cld + tb * 16], acc[tb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int tb = 0; tb < 5; ++tb) {
      if (tb < ntb) {
        const int b = tb * 16 + li;
        const int j = o0 + b, jp = j / opw;  // (output panels of opw columns, rows of `old` doubles: a plain block has one)
        double* op = Out + jp * ostride + (j - jp * opw);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r = lk + 4 * reg;  // f64 MFMA result map: row = (l>>4) + 4*reg, col = l&15
          if (r < rows && b < kx) op[(base + r) * old] = acc[tb][reg];
        }
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void gemm_nn_kernel(int n, int ku, int kx, const double* __restrict__ U,
                                                          int64_t rsu, int64_t csu, const double* __restrict__ C,
                                                          double* __restrict__ X, int ldx, double alpha, double beta) {
  extern __shared__ double Cs[];  // ku x cs_stride(kx), then the staged rows of U: kRB x (ku + 1)
  const int tid = threadIdx.x;
  const int cld = cs_stride(kx);
  const int uld = ku + 1;
  double* Us = Cs + ku * cld;
  for (int q = tid; q < ku * kx; q += kThreads) Cs[(q / kx) * cld + q % kx] = C[q];
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int ntb = (kx + 15) >> 4;  // <= 4
  const int ku4 = (ku + 3) & ~3;
  constexpr int IT = kRB * kMaxK / kThreads;  // 16 staged elements per lane
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kRB;
  // where the lane's staged elements live, relative to the row block: computed once (the divisions are by runtime ku)
  int64_t goff[IT];
  int soff[IT], srow[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int qq = tid + it * kThreads;
    int rr, a;
    if (rsu == 1) {
      rr = qq % kRB;
      a = qq / kRB;
    } else {
      a = qq % ku;
      rr = qq / ku;
    }
    const bool ok = qq < kRB * ku;
    goff[it] = rr * rsu + a * csu;
    soff[it] = rr * uld + a;
    srow[it] = ok ? rr : kRB;  // kRB: never below `rows`
  }
  double tmp[IT];
  // the next row block of U is fetched into registers while the current one multiplies
  auto fetch = [&](int64_t base) {
    const int rows = static_cast<int>((n - base) < kRB ? (n - base) : kRB);
    const double* ub = U + base * rsu;
#pragma unroll
    for (int it = 0; it < IT; ++it) tmp[it] = *((srow[it] < rows) ? ub + goff[it] : &g_zero_word);
  };
  int64_t base = static_cast<int64_t>(blockIdx.x) * kRB;
  if (base < n) fetch(base);
  for (; base < n; base += stride) {
    const int rows = static_cast<int>((n - base) < kRB ? (n - base) : kRB);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it)
      if (srow[it] < kRB) Us[soff[it]] = tmp[it];
    __syncthreads();
    if (base + stride < n) fetch(base + stride);
    if (16 * wave >= rows) continue;  // wave-uniform
    // the block of X this wave updates: loaded before the products, so its latency hides behind them
    double xv[4][4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * wave + lk + 4 * reg, b = tb * 16 + li;
        xv[tb][reg] = *((beta != 0.0 && tb < ntb && r < rows && b < kx) ? X + (base + r) * ldx + b : &g_zero_word);
      }
    // products of all column tiles first ...
    double4_t accs[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      accs[tb] = double4_t{0.0, 0.0, 0.0, 0.0};
      if (tb >= ntb) continue;
      const int b = tb * 16 + li;
      const bool okb = b < kx;
      for (int k0 = 0; k0 < ku4; k0 += 4) {
        const int k = k0 + lk;
        const double av = (k < ku) ? Us[(16 * wave + li) * uld + k] : 0.0;
        const double bv = (k < ku && okb) ? Cs[k * cld + b] : 0.0;
        accs[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, accs[tb], 0, 0, 0);
      }
    }
    // ... then ONE explicit wait for everything in flight (the block of X, the prefetched rows of U) ahead of the
    // stores: otherwise every masked store block below gets its own vmcnt(0), which also waits for the store before it
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt / expcnt untouched
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      if (tb >= ntb) break;
      const int b = tb * 16 + li;
      const bool okb = b < kx;
      // values first, stores after: a store's data register must not be reused before the store has completed
      double outv[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        outv[reg] = (beta == 0.0) ? alpha * accs[tb][reg] : beta * xv[tb][reg] + alpha * accs[tb][reg];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * wave + lk + 4 * reg;  // f64 MFMA result map: row = (l>>4) + 4*reg, col = l&15
        if (r < rows && okb) X[(base + r) * ldx + b] = outv[reg];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// block copies / column gather-scatter
// ---------------------------------------------------------------------------
struct ColMap {
  int32_t cols[kMaxK];
};

// mode 0: Dst[r][j] = Src[r][j] ; mode 1: Dst[r][j] = Src[r][map[j]] ; mode 2: Dst[r][map[j]] = Src[r][j]
template <int KP>
__global__ __launch_bounds__(kThreads) void copy_cols_kernel(int n, int k, const double* __restrict__ src, int lds,
                                                            double* __restrict__ dst, int ldd, int mode, ColMap m) {
  constexpr int RP = kThreads / KP;
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  if (c >= k) return;
  const int cs = (mode == 1) ? m.cols[c] : c;
  const int cd = (mode == 2) ? m.cols[c] : c;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * RP + rr; r < n; r += static_cast<int64_t>(gridDim.x) * RP)
    dst[r * ldd + cd] = src[r * lds + cs];
}

template <typename F>
static int dispatch_kp(int k, F&& f) {
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

static int reduce_to_host(eigd_ctx* ctx, const double* partial, int nblocks, int nout, double* dres, double* hout,
                          const int* gate = nullptr) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 3) / 4), dim3(kThreads), 0, ctx->stream, partial, nblocks,
                     nout, dres, gate);
  EIGD_LAUNCH_CHECK();
  if (hout) return eigd_d2h(ctx, hout, dres, sizeof(double) * nout);  // (small results through the page-locked bounce buffer)
  return EIGD_OK;
}

// partial-sum layout in ctx->scratch: [result nout][partials nblocks*nout]
// xnorm2 (optional, row-major U only): the squared column norms of X from the same pass, left in xnorm2[0..kx) on the device
// dest (optional, device): where the ku x kx result is left (default: the head of ctx->scratch)
static int gemm_tn_device(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t rsu, int64_t csu,
                          const double* dX, int ldx, double** dres, double* hC, double* xnorm2 = nullptr,
                          double* dest = nullptr) {
  const int nb = grid_for_rows(n, kRB);
  const int nout = ku * kx;
  int rc = ctx->ensure_scratch(sizeof(double) * ((static_cast<size_t>(nb) + 1) * nout + static_cast<size_t>(nb) * kx));
  if (rc) return rc;
  double* res = dest != nullptr ? dest : ctx->scratch;
  double* partial = ctx->scratch + nout;
  double* xpart = partial + static_cast<size_t>(nb) * nout;
  if (xnorm2 != nullptr) {
    if (csu != 1) {
      set_error("internal: column norms ride along with a row-major U only");
      return EIGD_E_INTERNAL;
    }
    if (ku <= 32 && kx <= 32)
      hipLaunchKernelGGL(gemm_tn_staged_kernel<32>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, ku, kx, dU, rsu, dX, ldx,
                         partial, xpart);
    else
      hipLaunchKernelGGL(gemm_tn_staged_kernel<64>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, ku, kx, dU, rsu, dX, ldx,
                         partial, xpart);
    EIGD_LAUNCH_CHECK();
    rc = reduce_to_host(ctx, partial, nb, nout, res, hC);
    if (rc) return rc;
    rc = reduce_to_host(ctx, xpart, nb, kx, xnorm2, nullptr);
    if (rc) return rc;
    if (dres) *dres = res;
    return EIGD_OK;
  }
  // few tiles (narrow X): the direct form, whose waves share the rows of a tile -- measured against the staged form at
  // 64 x 8: 137 against 165 us per call; wide results from a row-major U: staged through LDS
  constexpr bool staged_tn = true;
  const bool few = ((ku + 15) / 16) * ((kx + 15) / 16) <= 4;
  if (few)
    hipLaunchKernelGGL(gemm_tn_kernel<1>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, ku, kx, dU, rsu, csu, dX, ldx,
                       partial);
  else if (csu == 1 && staged_tn)
    hipLaunchKernelGGL(gemm_tn_staged_kernel<64>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, ku, kx, dU, rsu, dX, ldx,
                       partial, static_cast<double*>(nullptr));
  else
    hipLaunchKernelGGL(gemm_tn_kernel<4>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, ku, kx, dU, rsu, csu, dX, ldx,
                       partial);
  EIGD_LAUNCH_CHECK();
  rc = reduce_to_host(ctx, partial, nb, nout, res, hC);
  if (rc) return rc;
  if (dres) *dres = res;
  return EIGD_OK;
}

// normpart / nparts: see gemm_nn_direct_kernel (row-major U only); *nparts = number of partial sums per column
static int gemm_nn_device(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t rsu, int64_t csu,
                          const double* dC, double* dX, int ldx, double alpha, double beta, double* normpart = nullptr,
                          int* nparts = nullptr, const int* gate = nullptr) {
  const size_t cs_bytes = sizeof(double) * (ku * ((kx % 32 == 0) ? kx + 16 : kx) + kRB * (ku + 1));
  if (csu == 1 && ku <= 64) {  // row-major U: wave-private streaming without LDS staging
    const size_t cbytes = sizeof(double) * (ku <= 32 ? 32 : 64) * ((kx % 32 == 0) ? kx + 16 : kx);
    const int nbd = grid_for_rows(n, 64);
    if (nparts) *nparts = nbd;
    if (ku <= 32)
      hipLaunchKernelGGL(gemm_nn_direct_kernel<8>, dim3(nbd), dim3(kThreads), cbytes, ctx->stream, n, ku, kx, dU, rsu, dC,
                         dX, ldx, alpha, beta, normpart, gate);
    else
      hipLaunchKernelGGL(gemm_nn_direct_kernel<16>, dim3(nbd), dim3(kThreads), cbytes, ctx->stream, n, ku, kx, dU, rsu, dC,
                         dX, ldx, alpha, beta, normpart, gate);
    EIGD_LAUNCH_CHECK();
    return EIGD_OK;
  }
  if (normpart != nullptr || gate != nullptr) {
    set_error("internal: fused column norms need a row-major U");
    return EIGD_E_INTERNAL;
  }
  // a grid-stride kernel whose workgroups do not all fit the chip at once runs a second, thinly occupied round:
  // launch exactly what is resident
  int per_cu = 0;
  EIGD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_nn_kernel, kThreads, cs_bytes));
  const int resident = std::max(1, per_cu) * ctx->n_cu;
  const int nb = std::min(grid_for_rows(n, kRB), resident);
  hipLaunchKernelGGL(gemm_nn_kernel, dim3(nb), dim3(kThreads), cs_bytes, ctx->stream, n, ku, kx, dU, rsu, csu, dC, dX,
                     ldx, alpha, beta);
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

}  // namespace eigd

using namespace eigd;

extern "C" {

int eigd_gemm_tn(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t rsu, int64_t csu, const double* dX,
                 int ldx, double* hC) {
  EIGD_REQUIRE(ctx && dU && dX && hC, "null argument");
  EIGD_REQUIRE(n > 0 && ku >= 1 && ku <= kMaxK && kx >= 1 && kx <= kMaxK && ldx >= kx, "bad shape n=%d ku=%d kx=%d", n,
               ku, kx);
  return gemm_tn_device(ctx, n, ku, kx, dU, rsu, csu, dX, ldx, nullptr, hC);
}

int eigd_gemm_nn(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t rsu, int64_t csu, const double* hC,
                 double* dX, int ldx, double alpha, double beta) {
  EIGD_REQUIRE(ctx && dU && dX && hC, "null argument");
  EIGD_REQUIRE(n > 0 && ku >= 1 && ku <= kMaxK && kx >= 1 && kx <= kMaxK && ldx >= kx, "bad shape n=%d ku=%d kx=%d", n,
               ku, kx);
  int rc = ctx->ensure_coef(sizeof(double) * ku * kx);
  if (rc) return rc;
  rc = eigd_h2d(ctx, ctx->coef, hC, sizeof(double) * ku * kx);
  if (rc) return rc;
  return gemm_nn_device(ctx, n, ku, kx, dU, rsu, csu, ctx->coef, dX, ldx, alpha, beta);
}

int eigd_panels_times(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t pstride, const double* hC,
                      double* dOut, int64_t ostride, int opw, int ldo) {
  EIGD_REQUIRE(ctx && dU && hC && dOut, "null argument");
  EIGD_REQUIRE(n > 0 && ku >= 1 && ku <= kPanelsMax * kPanelW && kx >= 1 && pstride >= static_cast<int64_t>(n) * kPanelW &&
                   opw >= 1 && ldo >= std::min(opw, kx) && (kx <= opw || ostride >= static_cast<int64_t>(n) * ldo),
               "bad shape n=%d ku=%d kx=%d opw=%d ldo=%d (at most %d basis columns per call)", n, ku, kx, opw, ldo,
               kPanelsMax * kPanelW);
  {
    const int nop = (kx + opw - 1) / opw;
    const double* oend = dOut + static_cast<int64_t>(nop - 1) * ostride + static_cast<int64_t>(n) * ldo;
    EIGD_REQUIRE(oend <= dU || dU + ((ku + kPanelW - 1) / kPanelW) * pstride <= dOut, "the result must not overlap the basis");
  }
  int rc = ctx->ensure_coef(sizeof(double) * static_cast<size_t>(ku) * kx);
  if (rc) return rc;
  rc = eigd_h2d(ctx, ctx->coef, hC, sizeof(double) * static_cast<size_t>(ku) * kx);  // (hC is the caller's pageable memory)
  if (rc) return rc;
  const int np = (ku + kPanelW - 1) / kPanelW;
  // output columns in nearly equal chunks of at most kPanelOutMax, multiples of 2 wide (every chunk re-reads the basis)
  const int nchunk = (kx + kPanelOutMax - 1) / kPanelOutMax;
  const int wch = (((kx + nchunk - 1) / nchunk) + 1) & ~1;
  static bool attr_set = false;
  if (!attr_set) {
    const int cap = static_cast<int>(sizeof(double) * kPanelsMax * kPanelW * (kPanelOutMax + 16));
    EIGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_panels_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    EIGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_panels_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    EIGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_panels_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr_set = true;
  }
  const int64_t ngroups = (static_cast<int64_t>(n) + 15) / 16;
  const int nb = static_cast<int>(std::min<int64_t>(ctx->n_cu, (ngroups + kPanelThreads / 64 - 1) / (kPanelThreads / 64)));
  for (int o0 = 0; o0 < kx; o0 += wch) {
    const int kc = std::min(wch, kx - o0);
    const size_t lds = sizeof(double) * np * kPanelW * (kc + 16);
    if (np == 1)
      hipLaunchKernelGGL(gemm_nn_panels_kernel<1>, dim3(nb), dim3(kPanelThreads), lds, ctx->stream, n, ku, kc, dU, pstride,
                         ctx->coef, kx, o0, dOut, ostride, o0, opw, ldo);
    else if (np == 2)
      hipLaunchKernelGGL(gemm_nn_panels_kernel<2>, dim3(nb), dim3(kPanelThreads), lds, ctx->stream, n, ku, kc, dU, pstride,
                         ctx->coef, kx, o0, dOut, ostride, o0, opw, ldo);
    else
      hipLaunchKernelGGL(gemm_nn_panels_kernel<3>, dim3(nb), dim3(kPanelThreads), lds, ctx->stream, n, ku, kc, dU, pstride,
                         ctx->coef, kx, o0, dOut, ostride, o0, opw, ldo);
    EIGD_LAUNCH_CHECK();
  }
  return EIGD_OK;
}

// C = V^T X for up to 2 * kMaxK columns of V, left on the device in ctx->coef (ku x kx): the product kernel forms at
// most kMaxK x kMaxK entries per launch, a wider V goes in two column halves (no host round trip in between)
static int project_coefficients(eigd_ctx* ctx, int n, int ku, int kx, const double* dV, int ldv, const double* dX, int ldx,
                                double* xnorm2 = nullptr, double* cdest = nullptr) {
  // (behind the coefficients: the gate word of the measured updates and kMaxK column norms)
  int rc = ctx->ensure_coef(sizeof(double) * (static_cast<size_t>(ku) * kx + 8 + kMaxK));
  if (rc) return rc;
  for (int a0 = 0; a0 < ku; a0 += kMaxK) {
    const int ka = std::min(kMaxK, ku - a0);
    // (the reduction of the partial sums writes the coefficients where they are wanted: no copy kernel in between)
    rc = gemm_tn_device(ctx, n, ka, kx, dV + a0, ldv, 1, dX, ldx, nullptr, nullptr, a0 == 0 ? xnorm2 : nullptr,
                        (cdest != nullptr ? cdest : ctx->coef) + static_cast<size_t>(a0) * kx);
    if (rc) return rc;
  }
  return EIGD_OK;
}

int eigd_project(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int ldu, const double* dV, int ldv, double* dX,
                 int ldx) {
  EIGD_REQUIRE(ctx && dU && dV && dX, "null argument");
  EIGD_REQUIRE(n > 0 && ku >= 1 && ku <= 2 * kMaxK && kx >= 1 && kx <= kMaxK && ldx >= kx && ldu >= ku && ldv >= ku,
               "bad shape n=%d ku=%d kx=%d", n, ku, kx);
  int rc = project_coefficients(ctx, n, ku, kx, dV, ldv, dX, ldx);
  if (rc) return rc;
  for (int a0 = 0; a0 < ku; a0 += kMaxK) {
    const int ka = std::min(kMaxK, ku - a0);
    rc = gemm_nn_device(ctx, n, ka, kx, dU + a0, ldu, 1, ctx->coef + static_cast<size_t>(a0) * kx, dX, ldx, -1.0, 1.0);
    if (rc) return rc;
  }
  return EIGD_OK;
}


int eigd_project_norm2(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int ldu, const double* dV, int ldv,
                       double* dX, int ldx, double* dOut, double uscale, double tol) {
  EIGD_REQUIRE(ctx && dU && dV && dX && dOut, "null argument");
  EIGD_REQUIRE(uscale >= 0.0, "uscale = largest Euclidean column norm of U (0: unknown, taken as 1)");
  EIGD_REQUIRE(n > 0 && ku >= 1 && ku <= 2 * kMaxK && kx >= 1 && kx <= kMaxK && ldx >= kx && ldu >= ku && ldv >= ku,
               "bad shape n=%d ku=%d kx=%d", n, ku, kx);
  // The projection behind a Gram-Schmidt step (reference 1257) meets a block that is already B-orthogonal to Phi up to
  // rounding: the vectors it was built from were projected before.  Whether the update X -= U (V^T X) matters is
  // MEASURED: the coefficient pass delivers the squared column norms of X as well, a one-workgroup kernel compares
  // every coefficient with tol * |x_b| / uscale (tol: 1e-13 unless the caller says otherwise; uscale = the
  // largest Euclidean column norm of U, so that what is compared is the size of the update |u_a| |c_ab| against |x_b|
  // whatever the scale of the inner product), and the update pass -- the stream of U and two passes over X -- returns
  // at once when none does; the norms of the pass already made are then the result.
  EIGD_REQUIRE(tol >= 0.0, "tol = relative size of an update that matters (0: the default, 1e-13)");
  const double skip_tol = tol > 0.0 ? tol : 1e-13;
  constexpr bool measured = true;
  // scratch: [C tile (<= kMaxK x kx)] [partials of C, later the partial squared norms]
  const int nbd = grid_for_rows(n, 64);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(kMaxK) * kx + static_cast<size_t>(nbd) * kx));
  if (rc) return rc;
  rc = project_coefficients(ctx, n, ku, kx, dV, ldv, dX, ldx, measured ? dOut : nullptr);
  if (rc) return rc;
  int* gate = nullptr;
  if (measured) {
    gate = reinterpret_cast<int*>(ctx->coef + static_cast<size_t>(ku) * kx);
    if (!ctx->proj_stats) {
      EIGD_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->proj_stats), 2 * sizeof(int)));
      EIGD_HIP(hipMemsetAsync(ctx->proj_stats, 0, 2 * sizeof(int), ctx->stream));
    }
    hipLaunchKernelGGL(project_decide_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, ctx->coef, ku, kx, dOut,
                       uscale > 0.0 ? skip_tol / uscale : skip_tol, gate, ctx->proj_stats);
    EIGD_LAUNCH_CHECK();
  }
  double* normpart = ctx->scratch + static_cast<size_t>(kMaxK) * kx;  // (the partials of C are spent by now)
  int nparts = 0;
  for (int a0 = 0; a0 < ku; a0 += kMaxK) {
    const int ka = std::min(kMaxK, ku - a0);
    const bool last = a0 + kMaxK >= ku;                               // the norms belong to the finished block
    rc = gemm_nn_device(ctx, n, ka, kx, dU + a0, ldu, 1, ctx->coef + static_cast<size_t>(a0) * kx, dX, ldx, -1.0, 1.0,
                        last ? normpart : nullptr, last ? &nparts : nullptr, gate);
    if (rc) return rc;
  }
  rc = reduce_to_host(ctx, normpart, nparts, kx, dOut, nullptr, gate);
  if (rc) return rc;
  return publish_norm2(ctx, dOut, kx);
}

int eigd_project_to(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int ldu, const double* dV, int ldv, double* dX,
                    int ldx, double* dC, int ldc, double tol, double* dFlag, const double* dNorm2) {
  EIGD_REQUIRE(ctx && dU && dV && dX && dC, "null argument");
  EIGD_REQUIRE(n > 0 && ku >= 1 && ku <= kMaxK && kx >= 1 && kx <= kMaxK && ldx >= kx && ldu >= ku && ldv >= ku && ldc >= kx,
               "bad shape n=%d ku=%d kx=%d", n, ku, kx);
  EIGD_REQUIRE(tol >= 0.0 && (tol == 0.0 || dFlag != nullptr), "a measured update needs somewhere to report to");
  const bool measured = tol > 0.0;
  int rc = ctx->ensure_coef(sizeof(double) * (static_cast<size_t>(ku) * kx + 8 + kMaxK));
  if (rc) return rc;
  double* norms = ctx->coef + static_cast<size_t>(ku) * kx + 8;
  // a contiguous coefficient block receives the coefficients directly and serves the update as well
  double* coef = (ldc == kx) ? dC : ctx->coef;
  rc = project_coefficients(ctx, n, ku, kx, dV, ldv, dX, ldx, (measured && dNorm2 == nullptr) ? norms : nullptr, coef);
  if (rc) return rc;
  if (coef != dC)
    EIGD_HIP(hipMemcpy2DAsync(dC, sizeof(double) * ldc, ctx->coef, sizeof(double) * kx, sizeof(double) * kx, ku,
                              hipMemcpyDeviceToDevice, ctx->stream));
  int* gate = nullptr;
  if (measured) {
    gate = reinterpret_cast<int*>(ctx->coef + static_cast<size_t>(ku) * kx);
    hipLaunchKernelGGL(project_decide_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, coef, ku, kx,
                       dNorm2 != nullptr ? dNorm2 : norms, tol, gate, static_cast<int*>(nullptr), dFlag);
    EIGD_LAUNCH_CHECK();
  }
  return gemm_nn_device(ctx, n, ku, kx, dU, ldu, 1, coef, dX, ldx, -1.0, 1.0, nullptr, nullptr, gate);
}

int eigd_svqb_step(eigd_ctx* ctx, int n, int p, double* dX, int ldx, double* dBX, int ldbx, double* dC, int first,
                   double* dFlag, int update_bx) {
  EIGD_REQUIRE(ctx && dX && dBX && dC && dFlag, "null argument");
  EIGD_REQUIRE(n > 0 && p >= 1 && p <= kSvqbMax && ldx >= p && ldbx >= p, "bad shape n=%d p=%d (blocks of at most %d)", n, p,
               kSvqbMax);
  int rc = ctx->ensure_coef(sizeof(double) * p * p);
  if (rc) return rc;
  double* gram = nullptr;
  rc = gemm_tn_device(ctx, n, p, p, dX, ldx, 1, dBX, ldbx, &gram, nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(svqb_kernel, dim3(1), dim3(64), 0, ctx->stream, gram, p, ctx->coef, dC, first, dFlag);
  EIGD_LAUNCH_CHECK();
  // in place: a wave of the row-major product reads the sixteen rows of its group before it stores them
  rc = gemm_nn_device(ctx, n, p, p, dX, ldx, 1, ctx->coef, dX, ldx, 1.0, 0.0);
  if (rc || !update_bx) return rc;
  return gemm_nn_device(ctx, n, p, p, dBX, ldbx, 1, ctx->coef, dBX, ldbx, 1.0, 0.0);
}

int eigd_project_stats(eigd_ctx* ctx, int* out) {
  EIGD_REQUIRE(ctx && out, "null argument");
  out[0] = out[1] = 0;
  if (!ctx->proj_stats) return EIGD_OK;
  int rc = eigd_d2h(ctx, out, ctx->proj_stats, 2 * sizeof(int));
  if (rc) return rc;
  EIGD_HIP(hipMemsetAsync(ctx->proj_stats, 0, 2 * sizeof(int), ctx->stream));
  return EIGD_OK;
}

int eigd_coldot_dev(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, const double* dY, int ldy, double* dOut) {
  EIGD_REQUIRE(ctx && dX && dY && dOut, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldx >= k && ldy >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * k);
  if (rc) return rc;
  double* partial = ctx->scratch + k;
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(coldot_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dX, ldx, dY,
                       ldy, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return reduce_to_host(ctx, partial, nb, k, dOut, nullptr);
}

int eigd_coldot(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, const double* dY, int ldy, double* hout) {
  EIGD_REQUIRE(ctx && dX && dY && hout, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldx >= k && ldy >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * k);
  if (rc) return rc;
  double* res = ctx->scratch;
  double* partial = ctx->scratch + k;
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(coldot_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dX, ldx, dY,
                       ldy, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return reduce_to_host(ctx, partial, nb, k, res, hout);
}

int eigd_coldot_dd(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, const double* dY, int ldy, double* hout) {
  EIGD_REQUIRE(ctx && dX && dY && hout, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldx >= k && ldy >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = std::min(grid_for_rows(n, (kThreads / kp) * 8), 2048);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * 2 * k);
  if (rc) return rc;
  double* res = ctx->scratch;
  double* partial = ctx->scratch + 2 * k;
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(coldot_dd_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dX, ldx, dY,
                       ldy, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_dd_kernel, dim3((k + 63) / 64), dim3(64), 0, ctx->stream, partial, nb, k, res);
  EIGD_LAUNCH_CHECK();
  return eigd_d2h(ctx, hout, res, sizeof(double) * 2 * k);
}

int eigd_lincomb(eigd_ctx* ctx, int n, int k, double* dOut, int ldo, int nterms, const double* const* dXs,
                 const int* ldxs, const double* hcoef) {
  EIGD_REQUIRE(ctx && dOut && dXs && ldxs && hcoef, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && nterms >= 1 && nterms <= 4 && ldo >= k, "bad shape n=%d k=%d nterms=%d",
               n, k, nterms);
  LinArgs a;
  for (int t = 0; t < 4; ++t) {
    a.x[t] = (t < nterms) ? dXs[t] : nullptr;
    a.ld[t] = (t < nterms) ? ldxs[t] : 0;
    for (int c = 0; c < kMaxK; ++c) a.coef[t][c] = (t < nterms && c < k) ? hcoef[t * k + c] : 0.0;
    if (t < nterms) EIGD_REQUIRE(dXs[t] != nullptr && ldxs[t] >= k, "bad term %d", t);
  }
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 4);
  int rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(lincomb_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dOut, ldo,
                       nterms, a);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_stack_dot(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* dT,
                   int ldt, double* hH) {
  EIGD_REQUIRE(ctx && dS && dT && hH, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ns >= 1 && ldt >= k && lds >= k &&
                   slab >= static_cast<int64_t>(n - 1) * lds + k && static_cast<int64_t>(n) * lds < (int64_t(1) << 31),
               "bad shape n=%d k=%d ns=%d (a slab must hold fewer than 2^31 doubles)", n, k, ns);
  constexpr int JB = 16;
  const int kslab = k;
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) * JB * k + static_cast<size_t>(ns) * k));
  if (rc) return rc;
  double* res = ctx->scratch;                               // ns * k results
  double* partial = ctx->scratch + static_cast<size_t>(ns) * k;  // nb * JB * k
  for (int j0 = 0; j0 < ns; j0 += JB) {
    const int nj = std::min(JB, ns - j0);
    rc = dispatch_kp(k, [&](auto KP) {
      hipLaunchKernelGGL((stack_dot_kernel<decltype(KP)::value, JB>), dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, nj,
                         dS + j0 * slab, slab, lds, dT, ldt, partial, kslab);
    });
    if (rc) return rc;
    EIGD_LAUNCH_CHECK();
    rc = reduce_to_host(ctx, partial, nb, nj * k, res + static_cast<size_t>(j0) * k, nullptr);
    if (rc) return rc;
  }
  return eigd_d2h(ctx, hH, res, sizeof(double) * ns * k);
}

static int stack_dot_device(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* dT,
                            int ldt, double** dres, int kslab) {
  constexpr int JB = 16;
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) * JB * k + static_cast<size_t>(ns) * k));
  if (rc) return rc;
  double* res = ctx->scratch;                                    // ns * k results
  double* partial = ctx->scratch + static_cast<size_t>(ns) * k;  // nb * JB * k
  for (int j0 = 0; j0 < ns; j0 += JB) {
    const int nj = std::min(JB, ns - j0);
    rc = dispatch_kp(k, [&](auto KP) {
      hipLaunchKernelGGL((stack_dot_kernel<decltype(KP)::value, JB>), dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, nj,
                         dS + j0 * slab, slab, lds, dT, ldt, partial, kslab);
    });
    if (rc) return rc;
    EIGD_LAUNCH_CHECK();
    rc = reduce_to_host(ctx, partial, nb, nj * k, res + static_cast<size_t>(j0) * k, nullptr);
    if (rc) return rc;
  }
  *dres = res;
  return EIGD_OK;
}

static int stack_axpy_device(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* dH,
                             double* dT, int ldt, double alpha, int kslab) {
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(stack_axpy_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), sizeof(double) * ns * k,
                       ctx->stream, n, k, ns, dS, slab, lds, dH, dT, ldt, alpha, kslab);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

static int stack_axpy_dot_device(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds,
                                 const double* dH1, double* dT, int ldt, double alpha, double** dres, int kslab) {
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * ns * k);
  if (rc) return rc;
  double* res = ctx->scratch;
  double* partial = ctx->scratch + static_cast<size_t>(ns) * k;
  const size_t shm = sizeof(double) * ns * k;
  rc = dispatch_kp(k, [&](auto KP) {
    constexpr int kpv = decltype(KP)::value;
    if (ns <= 8)
      hipLaunchKernelGGL((stack_axpy_dot_kernel<kpv, 8>), dim3(nb), dim3(kThreads), shm, ctx->stream, n, k, ns, dS, slab,
                         lds, dH1, dT, ldt, alpha, partial, kslab);
    else if (ns <= 16)
      hipLaunchKernelGGL((stack_axpy_dot_kernel<kpv, 16>), dim3(nb), dim3(kThreads), shm, ctx->stream, n, k, ns, dS, slab,
                         lds, dH1, dT, ldt, alpha, partial, kslab);
    else
      hipLaunchKernelGGL((stack_axpy_dot_kernel<kpv, 32>), dim3(nb), dim3(kThreads), shm, ctx->stream, n, k, ns, dS, slab,
                         lds, dH1, dT, ldt, alpha, partial, kslab);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  rc = reduce_to_host(ctx, partial, nb, ns * k, res, nullptr);
  if (rc) return rc;
  *dres = res;
  return EIGD_OK;
}

static int stack_cgs2_impl(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, double* dT, int ldt,
                           double tol, double* hH, int* hpasses, int kslab) {
  EIGD_REQUIRE(ctx && dS && dT && hH, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ns >= 1 && ldt >= k && lds >= kslab &&
                   slab >= static_cast<int64_t>(n - 1) * lds + kslab && static_cast<int64_t>(n) * lds < (int64_t(1) << 31),
               "bad shape n=%d k=%d ns=%d (a slab must hold fewer than 2^31 doubles)", n, k, ns);
  EIGD_REQUIRE(static_cast<size_t>(ns) * k * sizeof(double) <= 60 * 1024, "stack too deep for one pass: ns*k=%d", ns * k);
  const size_t nh = static_cast<size_t>(ns) * k;
  int rc = ctx->ensure_coef(sizeof(double) * 2 * nh);  // [h1][h2]: device copies of the coefficients
  if (rc) return rc;
  double* res = nullptr;
  // pass 1: h1 = S^T T.  It stays on the device for pass 2; the host sees it together with h2.
  rc = stack_dot_device(ctx, n, k, ns, dS, slab, lds, dT, ldt, &res, kslab);
  if (rc) return rc;
  EIGD_HIP(hipMemcpyAsync(ctx->coef, res, sizeof(double) * nh, hipMemcpyDeviceToDevice, ctx->stream));
  int passes = 2;
  if (ns <= 32) {  // pass 2, fused: T -= S h1 and h2 = S^T T in one pass over the stack
    rc = stack_axpy_dot_device(ctx, n, k, ns, dS, slab, lds, ctx->coef, dT, ldt, -1.0, &res, kslab);
    if (rc) return rc;
  } else {
    // more than 32 slabs: subtract the tail slabs' part first, then the fused pass over the first 32 (T is final
    // there, so it measures h2 of those slabs), then one dot for the tail slabs: the tail is read twice, the
    // first 32 slabs once
    const int nb = ns - 32;
    const size_t na_k = static_cast<size_t>(32) * k;
    rc = stack_axpy_device(ctx, n, k, nb, dS + 32 * slab, slab, lds, ctx->coef + na_k, dT, ldt, -1.0, kslab);
    if (rc) return rc;
    rc = stack_axpy_dot_device(ctx, n, k, 32, dS, slab, lds, ctx->coef, dT, ldt, -1.0, &res, kslab);
    if (rc) return rc;
    EIGD_HIP(hipMemcpyAsync(ctx->coef + nh, res, sizeof(double) * na_k, hipMemcpyDeviceToDevice, ctx->stream));
    rc = stack_dot_device(ctx, n, k, nb, dS + 32 * slab, slab, lds, dT, ldt, &res, kslab);
    if (rc) return rc;
    EIGD_HIP(hipMemcpyAsync(ctx->coef + nh + na_k, res, sizeof(double) * nb * k, hipMemcpyDeviceToDevice, ctx->stream));
    res = ctx->coef + nh;  // h2 assembled in place
    passes = 3;
  }
  if (ctx->pinned_h_bytes < sizeof(double) * 2 * nh) {
    if (ctx->pinned_h) {
      EIGD_HIP(hipStreamSynchronize(ctx->stream));
      EIGD_HIP(hipHostFree(ctx->pinned_h));
      ctx->pinned_h = nullptr;
      ctx->pinned_h_bytes = 0;
    }
    const size_t want = sizeof(double) * 2 * nh + (size_t(1) << 16);
    EIGD_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned_h), want, hipHostMallocDefault));
    ctx->pinned_h_bytes = want;
  }
  double* h1 = ctx->pinned_h;
  double* h2 = ctx->pinned_h + nh;
  EIGD_HIP(hipMemcpyAsync(h1, ctx->coef, sizeof(double) * nh, hipMemcpyDeviceToHost, ctx->stream));
  EIGD_HIP(hipMemcpyAsync(h2, res, sizeof(double) * nh, hipMemcpyDeviceToHost, ctx->stream));
  if (res != ctx->coef + nh)
    EIGD_HIP(hipMemcpyAsync(ctx->coef + nh, res, sizeof(double) * nh, hipMemcpyDeviceToDevice, ctx->stream));
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  // what one-pass Gram-Schmidt left along S, column by column: subtract it only where it matters
  bool again = false;
  for (int c = 0; c < k && !again; ++c) {
    double n1 = 0.0, n2 = 0.0;
    for (int j = 0; j < ns; ++j) {
      n1 += h1[static_cast<size_t>(j) * k + c] * h1[static_cast<size_t>(j) * k + c];
      n2 += h2[static_cast<size_t>(j) * k + c] * h2[static_cast<size_t>(j) * k + c];
    }
    if (std::sqrt(n2) > tol * std::sqrt(n1)) again = true;
  }
  if (again) {
    rc = stack_axpy_device(ctx, n, k, ns, dS, slab, lds, ctx->coef + nh, dT, ldt, -1.0, kslab);
    if (rc) return rc;
    for (size_t q = 0; q < nh; ++q) h1[q] += h2[q];
    passes += 1;
  }
  std::memcpy(hH, h1, sizeof(double) * nh);
  if (hpasses) *hpasses = passes;
  return EIGD_OK;
}

int eigd_stack_cgs2(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, double* dT, int ldt,
                    double tol, double* hH, int* hpasses) {
  return stack_cgs2_impl(ctx, n, k, ns, dS, slab, lds, dT, ldt, tol, hH, hpasses, k);
}

int eigd_stack_cgs2_pair(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, double* dT, int ldt,
                         double tol, double* hH, int* hpasses) {
  EIGD_REQUIRE(k >= 1 && 2 * k <= kMaxK, "a pair of blocks holds at most %d columns each", kMaxK / 2);
  return stack_cgs2_impl(ctx, n, 2 * k, ns, dS, slab, lds, dT, ldt, tol, hH, hpasses, k);
}

int eigd_colnorm2_dev(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, double* dOut) {
  EIGD_REQUIRE(ctx && dX && dOut, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldx >= k, "bad shape n=%d k=%d", n, k);
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * k);
  if (rc) return rc;
  double* partial = ctx->scratch + k;
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(coldot_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dX, ldx, dX,
                       ldx, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  rc = reduce_to_host(ctx, partial, nb, k, dOut, nullptr);
  if (rc) return rc;
  return publish_norm2(ctx, dOut, k);
}

// copy for the host, in stream order right behind the reduction; eigd_colnorm2_fetch collects it
}  // extern "C"

int eigd::publish_norm2(eigd_ctx* ctx, const double* dOut, int k) {
  if (!ctx->pinned) {
    EIGD_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), sizeof(double) * 2 * kMaxK, hipHostMallocDefault));
    EIGD_HIP(hipEventCreateWithFlags(&ctx->ev_pinned, hipEventDisableTiming));
  }
  EIGD_HIP(hipMemcpyAsync(ctx->pinned, dOut, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
  EIGD_HIP(hipEventRecord(ctx->ev_pinned, ctx->stream));
  ctx->pinned_count = k;
  return EIGD_OK;
}

extern "C" {

int eigd_colnorm2_publish(eigd_ctx* ctx, const double* dNorm2, int k) {
  EIGD_REQUIRE(ctx && dNorm2 && k >= 1 && k <= 2 * kMaxK, "bad argument");
  return publish_norm2(ctx, dNorm2, k);
}

int eigd_colnorm2_fetch(eigd_ctx* ctx, double* hout, int k) {
  EIGD_REQUIRE(ctx && hout, "null argument");
  EIGD_REQUIRE(ctx->pinned && k == ctx->pinned_count, "no pending column norms of width %d", k);
  EIGD_HIP(hipEventSynchronize(ctx->ev_pinned));
  std::memcpy(hout, ctx->pinned, sizeof(double) * k);
  ctx->pinned_count = 0;
  return EIGD_OK;
}

int eigd_scale_inv_norm(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, double* dOut, int ldo,
                        const double* dNorm2, const unsigned char* hskip) {
  EIGD_REQUIRE(ctx && dX && dOut && dNorm2, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ldx >= k && ldo >= k, "bad shape n=%d k=%d", n, k);
  SkipMask m;
  for (int c = 0; c < kMaxK; ++c) m.skip[c] = (hskip && c < k) ? hskip[c] : 0;
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 4);
  int rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(scale_inv_norm_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dX, ldx,
                       dOut, ldo, dNorm2, m);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_pair_orthonormalise(eigd_ctx* ctx, int n, int k, double* dT, int ldt, const double* dNorm2, double* dW1, int ldw1,
                             double* dW2, int ldw2, const unsigned char* hskip, double* dOut) {
  EIGD_REQUIRE(ctx && dT && dNorm2 && dW1 && dW2 && dOut, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && 4 * k <= 2 * kMaxK && ldt >= 2 * k && ldw1 >= k && ldw2 >= k, "bad shape n=%d k=%d", n, k);
  SkipMask m;
  for (int c = 0; c < kMaxK; ++c) m.skip[c] = (hskip && c < k) ? hskip[c] : 0;
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * 2 * k);
  if (rc) return rc;
  double* partial = ctx->scratch + 2 * k;
  // dOut = [ |T1|^2 (k) | T1.T2 (k) | |T2 new|^2 (k) | W1.T2 new (k) ]
  EIGD_HIP(hipMemcpyAsync(dOut, dNorm2, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(coldot_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dT, ldt, dT + k,
                       ldt, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  rc = reduce_to_host(ctx, partial, nb, k, dOut + k, nullptr);
  if (rc) return rc;
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(pair_apply_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dT, ldt, dOut,
                       dOut + k, dW1, ldw1, m, partial);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  rc = reduce_to_host(ctx, partial, nb, 2 * k, dOut + 2 * k, nullptr);
  if (rc) return rc;
  rc = eigd_scale_inv_norm(ctx, n, k, dT + k, ldt, dW2, ldw2, dOut + 2 * k, hskip);
  if (rc) return rc;
  return publish_norm2(ctx, dOut, 4 * k);
}

int eigd_stack_axpy(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* hH,
                    double* dT, int ldt, double alpha) {
  EIGD_REQUIRE(ctx && dS && dT && hH, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ns >= 1 && ldt >= k && lds >= k &&
                   slab >= static_cast<int64_t>(n - 1) * lds + k && static_cast<int64_t>(n) * lds < (int64_t(1) << 31),
               "bad shape n=%d k=%d ns=%d (a slab must hold fewer than 2^31 doubles)", n, k, ns);
  EIGD_REQUIRE(static_cast<size_t>(ns) * k * sizeof(double) <= 60 * 1024, "stack too deep for one pass: ns*k=%d", ns * k);
  int rc = ctx->ensure_coef(sizeof(double) * ns * k);
  if (rc) return rc;
  EIGD_HIP(hipMemcpyAsync(ctx->coef, hH, sizeof(double) * ns * k, hipMemcpyHostToDevice, ctx->stream));
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 4);
  rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(stack_axpy_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), sizeof(double) * ns * k,
                       ctx->stream, n, k, ns, dS, slab, lds, ctx->coef, dT, ldt, alpha, k);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

// the same with the coefficients on the device (ns x k, contiguous): no host round trip
int eigd_stack_axpy_dev(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* dH,
                        double* dT, int ldt, double alpha) {
  EIGD_REQUIRE(ctx && dS && dT && dH, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ns >= 1 && ldt >= k && lds >= k &&
                   slab >= static_cast<int64_t>(n - 1) * lds + k && static_cast<int64_t>(n) * lds < (int64_t(1) << 31),
               "bad shape n=%d k=%d ns=%d (a slab must hold fewer than 2^31 doubles)", n, k, ns);
  EIGD_REQUIRE(static_cast<size_t>(ns) * k * sizeof(double) <= 60 * 1024, "stack too deep for one pass: ns*k=%d", ns * k);
  return stack_axpy_device(ctx, n, k, ns, dS, slab, lds, dH, dT, ldt, alpha, k);
}

int eigd_stack_axpy_dot(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* hH1,
                        double* dT, int ldt, double alpha, double* hH2) {
  EIGD_REQUIRE(ctx && dS && dT && hH1 && hH2, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK && ns >= 1 && ns <= 32 && ldt >= k && lds >= k &&
                   slab >= static_cast<int64_t>(n - 1) * lds + k && static_cast<int64_t>(n) * lds < (int64_t(1) << 31),
               "bad shape n=%d k=%d ns=%d (a slab must hold fewer than 2^31 doubles)", n, k, ns);
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 8);
  int rc = ctx->ensure_coef(sizeof(double) * ns * k);
  if (rc) return rc;
  rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(nb) + 1) * ns * k);
  if (rc) return rc;
  EIGD_HIP(hipMemcpyAsync(ctx->coef, hH1, sizeof(double) * ns * k, hipMemcpyHostToDevice, ctx->stream));
  EIGD_HIP(hipStreamSynchronize(ctx->stream));
  double* res = ctx->scratch;
  double* partial = ctx->scratch + static_cast<size_t>(ns) * k;
  const size_t shm = sizeof(double) * ns * k;
  rc = dispatch_kp(k, [&](auto KP) {
    constexpr int kpv = decltype(KP)::value;
    if (ns <= 8)
      hipLaunchKernelGGL((stack_axpy_dot_kernel<kpv, 8>), dim3(nb), dim3(kThreads), shm, ctx->stream, n, k, ns, dS, slab,
                         lds, ctx->coef, dT, ldt, alpha, partial, k);
    else if (ns <= 16)
      hipLaunchKernelGGL((stack_axpy_dot_kernel<kpv, 16>), dim3(nb), dim3(kThreads), shm, ctx->stream, n, k, ns, dS, slab,
                         lds, ctx->coef, dT, ldt, alpha, partial, k);
    else
      hipLaunchKernelGGL((stack_axpy_dot_kernel<kpv, 32>), dim3(nb), dim3(kThreads), shm, ctx->stream, n, k, ns, dS, slab,
                         lds, ctx->coef, dT, ldt, alpha, partial, k);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return reduce_to_host(ctx, partial, nb, ns * k, res, hH2);
}

static int copy_cols(eigd_ctx* ctx, int n, int k, const double* dSrc, int lds, double* dDst, int ldd, int mode,
                     const int32_t* hcols, int maxcol_src, int maxcol_dst) {
  EIGD_REQUIRE(ctx && dSrc && dDst, "null argument");
  EIGD_REQUIRE(n > 0 && k >= 1 && k <= kMaxK, "bad shape n=%d k=%d", n, k);
  ColMap m;
  for (int c = 0; c < kMaxK; ++c) m.cols[c] = c;
  if (mode != 0) {
    EIGD_REQUIRE(hcols != nullptr, "column map is null");
    for (int c = 0; c < k; ++c) {
      EIGD_REQUIRE(hcols[c] >= 0 && hcols[c] < (mode == 1 ? maxcol_src : maxcol_dst), "column index %d out of range",
                   hcols[c]);
      m.cols[c] = hcols[c];
    }
  }
  const int kp = next_pow2(k);
  const int nb = grid_for_rows(n, (kThreads / kp) * 4);
  int rc = dispatch_kp(k, [&](auto KP) {
    hipLaunchKernelGGL(copy_cols_kernel<decltype(KP)::value>, dim3(nb), dim3(kThreads), 0, ctx->stream, n, k, dSrc, lds,
                       dDst, ldd, mode, m);
  });
  if (rc) return rc;
  EIGD_LAUNCH_CHECK();
  return EIGD_OK;
}

int eigd_copy_block(eigd_ctx* ctx, int n, int k, const double* dSrc, int lds, double* dDst, int ldd) {
  EIGD_REQUIRE(lds >= k && ldd >= k, "leading dimensions too small");
  return copy_cols(ctx, n, k, dSrc, lds, dDst, ldd, 0, nullptr, lds, ldd);
}

int eigd_gather_cols(eigd_ctx* ctx, int n, int kdst, const double* dSrc, int lds, const int32_t* hcols, double* dDst,
                     int ldd) {
  EIGD_REQUIRE(ldd >= kdst, "leading dimension too small");
  return copy_cols(ctx, n, kdst, dSrc, lds, dDst, ldd, 1, hcols, lds, ldd);
}

int eigd_scatter_cols(eigd_ctx* ctx, int n, int ksrc, const double* dSrc, int lds, const int32_t* hcols, double* dDst,
                      int ldd) {
  EIGD_REQUIRE(lds >= ksrc, "leading dimension too small");
  return copy_cols(ctx, n, ksrc, dSrc, lds, dDst, ldd, 2, hcols, lds, ldd);
}

}  // extern "C"
