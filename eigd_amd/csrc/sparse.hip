// CSR SpMV / SpMM for gfx950.
//
//   k == 1 : CSR-stream SpMV.  A workgroup owns a contiguous block of rows whose
//            non-zeros fit one LDS tile; the 256 lanes stream val/col coalesced,
//            stage the products val*x[col] in LDS, then one lane per row adds its
//            segment in CSR order -- the same order scipy's csr_matvec uses, with
//            separately rounded multiply and add, so y is bit-identical to the
//            reference `A @ x`.  Workgroups are dealt to XCDs in contiguous row
//            ranges so each XCD's L2 holds one slice of x.
//   k >= 2 : row-major multi-vector product; a wave covers 64/kp rows x kp columns.
//            Tiled form: the rows are cut into blocks of kTileRows; at upload every block gets the sorted
//            list of the distinct columns it touches and each non-zero a 16-bit position in that list.
//            A workgroup stages those rows of X in LDS once (neighbouring matrix rows share most of
//            them: 5.6x fewer gathered bytes on the Q4 meshes) and the products read LDS, eight lanes
//            per matrix row with k/8 neighbouring columns each; the adds stay in CSR order.  Blocks
//            whose column lists do not fit LDS use the direct-gather kernel.
//
// algorithmic bytes: SpMV 12 nnz + 20 n ; SpMM 12 nnz + 4 n + 16 n k.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "common.h"

struct eigd_mat {
  eigd_ctx* ctx = nullptr;
  int n = 0;      // rows
  int ncols = 0;  // columns (== n for the operators A, B; rectangular for gather / averaging maps)
  int64_t nnz = 0;
  int32_t* indptr = nullptr;
  int32_t* indices = nullptr;
  double* data = nullptr;
  int32_t* rowblocks = nullptr;  // nblocks + 1
  int nblocks = 0;
  // tiled SpMM
  int ntiles = 0, umax = 0;       // row tiles; longest distinct-column list
  int tnz_cap = 0;                // non-zeros of a tile staged through LDS: the fullest tile's, at most kTileNnz
  int32_t* tile_ptr = nullptr;    // ntiles + 1 : offsets into ucols
  int32_t* ucols = nullptr;       // distinct columns of each tile, ascending
  uint16_t* lidx = nullptr;       // per non-zero: position of its column in the tile's list
};

namespace eigd {

constexpr int kTileRows = 32;    // matrix rows per SpMM tile
constexpr int kTileLds = 40 * 1024;  // LDS budget of the staged X rows (bytes)
constexpr int kTileNnz = 1024;       // most non-zeros of a tile staged through LDS (more: read from global memory)
constexpr int kNnzTile = 2048;   // products staged per workgroup (16 KiB of LDS)
constexpr int kMaxRowsTile = 256;


__device__ __forceinline__ int xcd_remap(int b, int nblocks_padded) {
  // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range
  const int per = nblocks_padded >> 3;
  return (b & 7) * per + (b >> 3);
}

__global__ __launch_bounds__(kThreads) void spmv_stream_kernel(const int32_t* __restrict__ rowblocks, int nblocks,
                                                              int nblocks_padded,
                                                              const int32_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices,
                                                              const double* __restrict__ vals,
                                                              const double* __restrict__ x, double* __restrict__ y,
                                                              double alpha, double beta) {
  __shared__ double prod[kNnzTile];
  __shared__ double red[kThreads];
  const int b = xcd_remap(blockIdx.x, nblocks_padded);
  if (b >= nblocks) return;
  const int tid = threadIdx.x;
  const int r0 = rowblocks[b], r1 = rowblocks[b + 1];
  const int e0 = indptr[r0], e1 = indptr[r1];
  if (e1 - e0 <= kNnzTile - 2) {
    // Lane p owns the non-zero PAIR (eb + 2p, eb + 2p + 1), eb = e0 rounded down to even: values come in as
    // one 16-byte load and indices as one 8-byte load per lane (non-temporal: streamed once, they should not
    // evict x from L2).  All loads of the tile are issued first, then the gathers of x, then the LDS stores:
    // three memory latencies per tile.  Row blocks hold at most kNnzTile - 2 non-zeros so the pairs fit.
    constexpr int IT = kNnzTile / (2 * kThreads);
    const int eb = e0 & ~1;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef int int2v __attribute__((ext_vector_type(2)));
    dbl2 v[IT];
    int2v col[IT];
    double xa[IT], xb[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = eb + 2 * (tid + it * kThreads);
      if (e < e1) {  // e + 1 may equal e1: the pair load stays inside the allocation (padded by one entry)
        v[it] = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(vals + e));
        col[it] = __builtin_nontemporal_load(reinterpret_cast<const int2v*>(indices + e));
      } else {
        v[it] = dbl2{0.0, 0.0};
        col[it] = int2v{0, 0};
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = eb + 2 * (tid + it * kThreads);
      xa[it] = (e >= e0 && e < e1) ? x[col[it][0]] : 0.0;
      xb[it] = (e + 1 < e1) ? x[col[it][1]] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = eb + 2 * (tid + it * kThreads);
      if (e >= e0 && e < e1) prod[e - e0] = __dmul_rn(v[it][0], xa[it]);
      if (e + 1 < e1) prod[e + 1 - e0] = __dmul_rn(v[it][1], xb[it]);
    }
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += kThreads) {
      const int a = indptr[r] - e0, z = indptr[r + 1] - e0;
      double s = 0.0;
      for (int q = a; q < z; ++q) s = __dadd_rn(s, prod[q]);
      y[r] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[r];
    }
  } else {
    // a single long row: strided partial sums, tree reduction
    double s = 0.0;
    for (int e = e0 + tid; e < e1; e += kThreads) s += vals[e] * x[indices[e]];
    red[tid] = s;
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (tid == 0) y[r0] = (beta == 0.0) ? alpha * red[0] : alpha * red[0] + beta * y[r0];
  }
}

template <int KP>
__global__ __launch_bounds__(kThreads) void spmm_rows_kernel(int n, int k, const int32_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const double* __restrict__ vals,
                                                            const double* __restrict__ X, int ldx,
                                                            double* __restrict__ Y, int ldy, double alpha,
                                                            double beta) {
  constexpr int RP = kThreads / KP;
  const int c = threadIdx.x % KP;
  const int r = blockIdx.x * RP + threadIdx.x / KP;
  if (r >= n || c >= k) return;
  const int a = indptr[r], z = indptr[r + 1];
  double s = 0.0;
  int e = a;
  // eight non-zeros per trip: index / value / gathered-row loads are all issued before the first add;
  // the adds stay in CSR order (bit-identical to scipy's csr_matvecs)
  for (; e + 8 <= z; e += 8) {
    int col[8];
    double v[8], x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      col[q] = indices[e + q];
      v[q] = vals[e + q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = X[static_cast<int64_t>(col[q]) * ldx + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) s = __dadd_rn(s, __dmul_rn(v[q], x[q]));
  }
  for (; e < z; ++e) s = __dadd_rn(s, __dmul_rn(vals[e], X[static_cast<int64_t>(indices[e]) * ldx + c]));
  double* yp = Y + static_cast<int64_t>(r) * ldy + c;
  *yp = (beta == 0.0) ? alpha * s : alpha * s + beta * (*yp);
}

// Tiled SpMM: see the file header.
// Product phase: a row of the tile belongs to LPR = KP / CPL lanes, each with CPL neighbouring columns in registers
// (KP >= 8: eight lanes per row, so the 256 lanes take the 32 rows of the tile in one pass).  The value and the local
// column number of a non-zero are read from LDS once per lane and serve CPL products, the staged row of X comes as
// 16-byte pieces: at 32 columns 5.5 LDS cycles per 64 products instead of 10 (the kernel is bound by the LDS pipe).
// The sums stay in CSR order per (row, column): the mapping of lanes to columns does not enter the result.
template <int KP, int LANES = 8>
struct SpmmTile {
  static constexpr int CPL = (KP >= LANES) ? KP / LANES : 1;  // columns per lane
  static constexpr int LPR = KP / CPL;                         // lanes per row
  static constexpr int LD = (CPL >= 2) ? KP + 4 : KP + 1;      // Xs row stride in doubles (CPL >= 2: rows 16-byte aligned)
};

// DOTS (KP = 8, 16, 32; the short-recurrence sibk): the tile also leaves its share of the column sums x.r and x.y --
// x = X, y = the result, r = a third block -- in dots[tile][2 k] (fixed order: the rows of a wave by shuffles, the four waves
// in order), so that conjugate gradients need no pass of its own over z, r and y for its two inner products.  The
// products of Y are untouched: the result stays bit-identical to the kernel without DOTS.
template <int KP, int LANES, bool DOTS = false>
__global__ __launch_bounds__(kThreads) void spmm_tiled_kernel(int n, int k, int ntiles, int tiles_per_xcd,
                                                             const int32_t* __restrict__ tile_ptr,
                                                             const int32_t* __restrict__ ucols,
                                                             const int32_t* __restrict__ indptr,
                                                             const uint16_t* __restrict__ lidx,
                                                             const double* __restrict__ vals,
                                                             const double* __restrict__ X, int ldx,
                                                             double* __restrict__ Y, int ldy, double alpha, double beta,
                                                             int umax, int tnz_cap,
                                                             const double* __restrict__ R = nullptr, int ldr = 0,
                                                             double* __restrict__ dots = nullptr) {
  extern __shared__ __align__(16) double Xs[];
  constexpr int RP = kThreads / KP;  // staging: rows of X per trip, lane (rr, c) moves one double
  constexpr int CPL = SpmmTile<KP, LANES>::CPL, LPR = SpmmTile<KP, LANES>::LPR, LD = SpmmTile<KP, LANES>::LD;
  constexpr int RPP = kThreads / LPR;                    // product phase: matrix rows per pass
  constexpr int NR = (kTileRows + RPP - 1) / RPP;        // passes (1 for KP >= 8)
  // the tile's non-zeros (values, local column numbers) go through LDS too: requested with the X rows, coalesced,
  // instead of a dependent round trip per 8 non-zeros of a row in the product loop
  double* const Vs = Xs + static_cast<size_t>(umax) * LD;
  uint16_t* const Ls = reinterpret_cast<uint16_t*>(Vs + tnz_cap);  // (tnz_cap: multiple of 8)
  // contiguous tile ranges per XCD (workgroups are dealt round-robin over the 8 XCDs): neighbouring tiles share
  // most of their X rows, this keeps that reuse inside one L2
  const int tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int c = threadIdx.x % KP, rr = threadIdx.x / KP;
  const int pl = threadIdx.x % LPR, prow = threadIdx.x / LPR;  // product phase: lane pl of row prow, columns CPL pl ..
  const int u0 = tile_ptr[tile], nu = tile_ptr[tile + 1] - u0;
  const int rend = min(n, (tile + 1) * kTileRows);
  const int e0 = indptr[tile * kTileRows], tnz = indptr[rend] - e0;
  const bool staged = tnz <= tnz_cap;  // (a tile with very long rows reads its non-zeros from global memory)
  constexpr int NQ = kTileNnz / kThreads;
  int ra[NR], rz[NR];  // the extents of the lane's rows are requested with everything else
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = tile * kTileRows + prow + i * RPP;
    ra[i] = (r < rend) ? indptr[r] - e0 : 0;
    rz[i] = (r < rend) ? indptr[r + 1] - e0 : 0;
  }
  double sv[NQ];
  uint16_t sl[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = threadIdx.x + q * kThreads;
    const bool ok = staged && e < tnz;
    sv[q] = ok ? vals[e0 + e] : 0.0;
    sl[q] = ok ? lidx[e0 + e] : uint16_t(0);
  }
  // DOTS: the lane's own entries of x and r (its row, its columns) are requested with everything else
  [[maybe_unused]] double xo[CPL], ro[CPL];
  if constexpr (DOTS) {
    const int r = tile * kTileRows + prow;
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
      const bool ok = r < rend && CPL * pl + t < k;
      xo[t] = ok ? X[static_cast<int64_t>(r) * ldx + CPL * pl + t] : 0.0;
      ro[t] = ok ? R[static_cast<int64_t>(r) * ldr + CPL * pl + t] : 0.0;
    }
  }
  constexpr int SU = 16;  // staged rows per lane and trip: all their loads are in flight together
  for (int j0 = 0; j0 < nu; j0 += SU * RP) {
    int col[SU];
    double x[SU];
#pragma unroll
    for (int q = 0; q < SU; ++q) {
      const int j = j0 + q * RP + rr;
      col[q] = (j < nu) ? ucols[u0 + j] : -1;
    }
#pragma unroll
    for (int q = 0; q < SU; ++q) x[q] = (col[q] >= 0 && c < k) ? X[static_cast<int64_t>(col[q]) * ldx + c] : 0.0;
#pragma unroll
    for (int q = 0; q < SU; ++q) {
      const int j = j0 + q * RP + rr;
      if (j < nu) Xs[j * LD + c] = x[q];
    }
  }
  if (staged) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = threadIdx.x + q * kThreads;
      if (e < tnz) {
        Vs[e] = sv[q];
        Ls[e] = sl[q];
      }
    }
  }
  __syncthreads();
  const int cb = CPL * pl;  // the lane's first column
  if (!DOTS && cb >= k) return;
  [[maybe_unused]] double dzr[CPL], dzy[CPL];
  [[maybe_unused]] __shared__ double dred[DOTS ? (kThreads / 64) * 2 * KP : 1];
  if constexpr (DOTS) {
#pragma unroll
    for (int t = 0; t < CPL; ++t) dzr[t] = dzy[t] = 0.0;
  }
  auto finish_dots = [&]() {  // every lane of the workgroup comes here exactly once (DOTS)
    if constexpr (DOTS) {
      static_assert(KP >= LANES && NR == 1, "one pass over the tile's rows, eight lanes per row");
#pragma unroll
      for (int t = 0; t < CPL; ++t) {  // the eight rows of a wave: lanes 8 q + pl
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
          dzr[t] += __shfl_xor(dzr[t], off, 64);
          dzy[t] += __shfl_xor(dzy[t], off, 64);
        }
      }
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      if (lane < LPR) {
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
          dred[(wave * 2 + 0) * KP + cb + t] = dzr[t];
          dred[(wave * 2 + 1) * KP + cb + t] = dzy[t];
        }
      }
      __syncthreads();
      if (threadIdx.x < 2 * k) {
        const int q = threadIdx.x / k, c2 = threadIdx.x % k;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) v += dred[(w * 2 + q) * KP + c2];
        dots[static_cast<int64_t>(tile) * 2 * k + threadIdx.x] = v;
      }
    }
  };
  auto store_row = [&](int r, const double (&s)[CPL]) {
    double* yp = Y + static_cast<int64_t>(r) * ldy + cb;
#pragma unroll
    for (int t = 0; t < CPL; ++t)
      if (cb + t < k) {
        const double yv = (beta == 0.0) ? alpha * s[t] : alpha * s[t] + beta * yp[t];
        yp[t] = yv;
        if constexpr (DOTS) {  // (NR == 1: the row is the one whose entries were loaded at the start)
          dzr[t] += xo[t] * ro[t];
          dzy[t] += xo[t] * yv;
        }
      }
  };
  const bool active = cb < k;  // (DOTS: the lanes past the block's columns come along to the one barrier of finish_dots)
  if (active && staged) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = tile * kTileRows + prow + i * RPP;
      if (r >= rend) break;
      const int a = ra[i], z = rz[i];
      double s[CPL];
#pragma unroll
      for (int t = 0; t < CPL; ++t) s[t] = 0.0;
      int e = a;
      for (; e + 8 <= z; e += 8) {  // eight non-zeros per trip: their LDS reads are in flight together
        int li[8];
        double v[8], x[8][CPL];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          li[q] = Ls[e + q];
          v[q] = Vs[e + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double* xp = Xs + li[q] * LD + cb;
          if constexpr (CPL >= 2) {
            const double2* xp2 = reinterpret_cast<const double2*>(xp);
#pragma unroll
            for (int t = 0; t < CPL / 2; ++t) {
              const double2 w = xp2[t];
              x[q][2 * t] = w.x;
              x[q][2 * t + 1] = w.y;
            }
          } else {
            x[q][0] = xp[0];
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int t = 0; t < CPL; ++t) s[t] = __dadd_rn(s[t], __dmul_rn(v[q], x[q][t]));
      }
      for (; e < z; ++e) {
        const double v = Vs[e];
        const double* xp = Xs + Ls[e] * LD + cb;
#pragma unroll
        for (int t = 0; t < CPL; ++t) s[t] = __dadd_rn(s[t], __dmul_rn(v, xp[t]));
      }
      store_row(r, s);
    }
  } else if (active) {
  for (int r = tile * kTileRows + prow; r < rend; r += RPP) {
    const int a = indptr[r], z = indptr[r + 1];
    double s[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) s[t] = 0.0;
    for (int e = a; e < z; ++e) {
      const double v = vals[e];
      const double* xp = Xs + lidx[e] * LD + cb;
#pragma unroll
      for (int t = 0; t < CPL; ++t) s[t] = __dadd_rn(s[t], __dmul_rn(v, xp[t]));
    }
    store_row(r, s);
  }
  }
  finish_dots();
}

// first level of the sum over the tiles' shares (spmm_tiled_kernel, DOTS): workgroup g adds tiles g, g + G, ... for all
// nout = 2 k sums; the coefficient kernel of krylov.hip adds the G rows
__global__ __launch_bounds__(kThreads) void tile_dots_reduce_kernel(const double* __restrict__ dots, int ntiles, int nout,
                                                                   double* __restrict__ out) {
  __shared__ double red[kThreads];
  const int per = kThreads / nout;                 // tiles in flight per trip (nout <= 64: at least four)
  const int o = threadIdx.x % nout, sub = threadIdx.x / nout;
  double v = 0.0;
  if (sub < per)
    for (int t = blockIdx.x * per + sub; t < ntiles; t += gridDim.x * per) v += dots[static_cast<int64_t>(t) * nout + o];
  red[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < nout) {
    double t = 0.0;
    for (int q = 0; q < per; ++q) t += red[q * nout + threadIdx.x];
    out[static_cast<int64_t>(blockIdx.x) * nout + threadIdx.x] = t;
  }
}

}  // namespace eigd

using namespace eigd;

extern "C" {

int eigd_csr_upload(eigd_ctx* ctx, int n, int64_t nnz, const int32_t* hindptr, const int32_t* hindices,
                    const double* hdata, eigd_mat** out) {
  return eigd_csr_upload_rect(ctx, n, n, nnz, hindptr, hindices, hdata, out);
}

int eigd_csr_upload_rect(eigd_ctx* ctx, int n, int ncols, int64_t nnz, const int32_t* hindptr, const int32_t* hindices,
                         const double* hdata, eigd_mat** out) {
  EIGD_REQUIRE(ctx && hindptr && hindices && hdata && out, "null argument");
  EIGD_REQUIRE(n > 0 && ncols > 0 && nnz >= 0 && nnz < (int64_t(1) << 31), "bad matrix size %d x %d nnz=%lld", n, ncols,
               (long long)nnz);
  EIGD_REQUIRE(hindptr[0] == 0 && hindptr[n] == nnz, "indptr does not match nnz");
  for (int i = 0; i < n; ++i) EIGD_REQUIRE(hindptr[i + 1] >= hindptr[i], "indptr not monotone at row %d", i);
  for (int64_t e = 0; e < nnz; ++e)
    EIGD_REQUIRE(hindices[e] >= 0 && hindices[e] < ncols, "column index out of range at entry %lld", (long long)e);
  *out = nullptr;
  // row blocks for the CSR-stream kernel
  std::vector<int32_t> rb;
  rb.push_back(0);
  {
    int r = 0;
    while (r < n) {
      int start = r;
      int64_t cnt = 0;
      while (r < n && (r - start) < kMaxRowsTile) {
        int64_t len = hindptr[r + 1] - hindptr[r];
        if (cnt + len > kNnzTile - 2) break;
        cnt += len;
        ++r;
      }
      if (r == start) ++r;  // one long row on its own
      rb.push_back(r);
    }
  }
  // SpMM tiles: distinct columns of every block of kTileRows rows and the position of each non-zero's column in it
  const int ntiles = (n + kTileRows - 1) / kTileRows;
  std::vector<int32_t> tile_ptr(static_cast<size_t>(ntiles) + 1, 0), ucols;
  std::vector<uint16_t> lidx(static_cast<size_t>(nnz) + 8, 0);
  int umax = 0;
  int64_t tnz_max = 0;
  bool tiles_ok = true;
  {
    std::vector<int32_t> cols, stamp(static_cast<size_t>(ncols), -1), pos(static_cast<size_t>(ncols), 0);
    ucols.reserve(static_cast<size_t>(nnz) / 4 + 16);
    for (int t = 0; t < ntiles && tiles_ok; ++t) {
      const int r0 = t * kTileRows, r1 = std::min(n, r0 + kTileRows);
      cols.clear();
      for (int64_t e = hindptr[r0]; e < hindptr[r1]; ++e) {
        const int32_t cidx = hindices[e];
        if (stamp[cidx] != t) {
          stamp[cidx] = t;
          cols.push_back(cidx);
        }
      }
      std::sort(cols.begin(), cols.end());
      if (cols.size() > 65535) tiles_ok = false;
      umax = std::max<int>(umax, static_cast<int>(cols.size()));
      tnz_max = std::max<int64_t>(tnz_max, hindptr[r1] - hindptr[r0]);
      for (size_t q = 0; q < cols.size(); ++q) pos[cols[q]] = static_cast<int32_t>(q);
      for (int64_t e = hindptr[r0]; e < hindptr[r1]; ++e) lidx[e] = static_cast<uint16_t>(pos[hindices[e]]);
      ucols.insert(ucols.end(), cols.begin(), cols.end());
      tile_ptr[t + 1] = static_cast<int32_t>(ucols.size());
    }
  }
  EIGD_HIP(hipSetDevice(ctx->device));
  eigd_mat* A = new eigd_mat();
  A->ctx = ctx;
  A->n = n;
  A->ncols = ncols;
  A->nnz = nnz;
  A->nblocks = static_cast<int>(rb.size()) - 1;
  if (tiles_ok) {
    A->ntiles = ntiles;
    A->umax = umax;
    A->tnz_cap = static_cast<int>(std::min<int64_t>(kTileNnz, (tnz_max + 7) & ~int64_t(7)));
    hipError_t t1 = hipMalloc(reinterpret_cast<void**>(&A->tile_ptr), sizeof(int32_t) * tile_ptr.size());
    hipError_t t2 = hipMalloc(reinterpret_cast<void**>(&A->ucols), sizeof(int32_t) * std::max<size_t>(ucols.size(), 1));
    hipError_t t3 = hipMalloc(reinterpret_cast<void**>(&A->lidx), sizeof(uint16_t) * lidx.size());
    if (t1 != hipSuccess || t2 != hipSuccess || t3 != hipSuccess) {
      eigd_mat_free(A);
      set_error("hipMalloc failed for the SpMM tiles (n=%d nnz=%lld)", n, (long long)nnz);
      return EIGD_E_HIP;
    }
    EIGD_HIP(hipMemcpy(A->tile_ptr, tile_ptr.data(), sizeof(int32_t) * tile_ptr.size(), hipMemcpyHostToDevice));
    if (!ucols.empty()) EIGD_HIP(hipMemcpy(A->ucols, ucols.data(), sizeof(int32_t) * ucols.size(), hipMemcpyHostToDevice));
    EIGD_HIP(hipMemcpy(A->lidx, lidx.data(), sizeof(uint16_t) * lidx.size(), hipMemcpyHostToDevice));
  }
  hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&A->indptr), sizeof(int32_t) * (n + 1));
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&A->indices), sizeof(int32_t) * (nnz + 4));
  hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&A->data), sizeof(double) * (nnz + 4));
  hipError_t e4 = hipMalloc(reinterpret_cast<void**>(&A->rowblocks), sizeof(int32_t) * rb.size());
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    eigd_mat_free(A);
    set_error("hipMalloc failed for CSR matrix (n=%d nnz=%lld)", n, (long long)nnz);
    return EIGD_E_HIP;
  }
  EIGD_HIP(hipMemcpy(A->indptr, hindptr, sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  if (nnz > 0) {
    EIGD_HIP(hipMemcpy(A->indices, hindices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
    EIGD_HIP(hipMemcpy(A->data, hdata, sizeof(double) * nnz, hipMemcpyHostToDevice));
  }
  EIGD_HIP(hipMemcpy(A->rowblocks, rb.data(), sizeof(int32_t) * rb.size(), hipMemcpyHostToDevice));
  *out = A;
  return EIGD_OK;
}

int eigd_csr_update_values(eigd_mat* A, const double* hdata) {
  EIGD_REQUIRE(A && hdata, "null argument");
  EIGD_HIP(hipStreamSynchronize(A->ctx->stream));
  if (A->nnz > 0) EIGD_HIP(hipMemcpy(A->data, hdata, sizeof(double) * A->nnz, hipMemcpyHostToDevice));
  return EIGD_OK;
}

int eigd_csr_update_values_dev(eigd_mat* A, const double* dvals) {
  EIGD_REQUIRE(A && dvals, "null argument");
  if (A->nnz > 0)
    EIGD_HIP(hipMemcpyAsync(A->data, dvals, sizeof(double) * A->nnz, hipMemcpyDeviceToDevice, A->ctx->stream));
  return EIGD_OK;
}

int eigd_mat_free(eigd_mat* A) {
  if (!A) return EIGD_OK;
  if (A->ctx && A->ctx->stream) (void)hipStreamSynchronize(A->ctx->stream);
  if (A->indptr) (void)hipFree(A->indptr);
  if (A->indices) (void)hipFree(A->indices);
  if (A->data) (void)hipFree(A->data);
  if (A->rowblocks) (void)hipFree(A->rowblocks);
  if (A->tile_ptr) (void)hipFree(A->tile_ptr);
  if (A->ucols) (void)hipFree(A->ucols);
  if (A->lidx) (void)hipFree(A->lidx);
  delete A;
  return EIGD_OK;
}

int eigd_spmm(eigd_mat* A, const double* dX, int ldx, double* dY, int ldy, int k, double alpha, double beta) {
  EIGD_REQUIRE(A, "null argument");
  return eigd_spmm_on(A->ctx, A, dX, ldx, dY, ldy, k, alpha, beta);
}

int eigd_spmm_on(eigd_ctx* ctx, eigd_mat* A, const double* dX, int ldx, double* dY, int ldy, int k, double alpha,
                 double beta) {
  EIGD_REQUIRE(ctx && A && dX && dY, "null argument");
  EIGD_REQUIRE(ctx->device == A->ctx->device, "context and matrix live on different devices");
  EIGD_REQUIRE(k >= 1 && ldx >= k && ldy >= k, "bad block shape k=%d ldx=%d ldy=%d", k, ldx, ldy);
  EIGD_REQUIRE(dX != dY, "spmm cannot run in place");
  hipStream_t st = ctx->stream;
  if (k == 1 && ldx == 1 && ldy == 1) {
    const int nbp = (A->nblocks + 7) & ~7;
    hipLaunchKernelGGL(spmv_stream_kernel, dim3(nbp), dim3(kThreads), 0, st, A->rowblocks, A->nblocks, nbp, A->indptr,
                       A->indices, A->data, dX, dY, alpha, beta);
    EIGD_LAUNCH_CHECK();
    return EIGD_OK;
  }
  // widest block per launch: 64 columns, or 32 where the tile's rows of X fit the LDS at 32 columns but not at 64 (the
  // direct-gather kernel that would take a 64-column block is four times slower than two tiled launches of 32)
  auto tile_fits = [&](int kpv) {
    const int cplv = (kpv >= 8) ? kpv / 8 : 1;
    return A->ntiles > 0 && sizeof(double) * static_cast<size_t>(A->umax) * (cplv >= 2 ? kpv + 4 : kpv + 1) <= static_cast<size_t>(kTileLds);
  };
  const int chunk = (k > 32 && !tile_fits(64) && tile_fits(32)) ? 32 : kMaxK;
  for (int c0 = 0; c0 < k; c0 += chunk) {
    const int kb = std::min(chunk, k - c0);
    const int kp = std::max(2, next_pow2(kb));
    const int rp = kThreads / kp;
    const dim3 grid((A->n + rp - 1) / rp);
    // eight lanes per matrix row in the product phase, k/8 neighbouring columns each
    const int cpl = (kp >= 8) ? kp / 8 : 1;
    const size_t tile_lds = sizeof(double) * static_cast<size_t>(A->umax) * (cpl >= 2 ? kp + 4 : kp + 1);  // SpmmTile::LD
    const size_t tile_lds_all = tile_lds + static_cast<size_t>(A->tnz_cap) * (sizeof(double) + sizeof(uint16_t));
    if (A->ntiles > 0 && tile_lds <= static_cast<size_t>(kTileLds)) {
      const int per_xcd = (A->ntiles + 7) / 8;
      const dim3 tgrid(per_xcd * 8);
#define EIGD_SPMM_TILED(KP)                                                                                            \
  case KP:                                                                                                             \
    hipLaunchKernelGGL((spmm_tiled_kernel<KP, 8>), tgrid, dim3(kThreads), tile_lds_all, st, A->n, kb, A->ntiles,       \
                       per_xcd, A->tile_ptr, A->ucols, A->indptr, A->lidx, A->data, dX + c0, ldx, dY + c0, ldy,        \
                       alpha, beta, A->umax, A->tnz_cap);                                                              \
    break;
      switch (kp) {
        EIGD_SPMM_TILED(2)
        EIGD_SPMM_TILED(4)
        EIGD_SPMM_TILED(8)
        EIGD_SPMM_TILED(16)
        EIGD_SPMM_TILED(32)
        EIGD_SPMM_TILED(64)
        default:
          set_error("internal: unexpected kp=%d", kp);
          return EIGD_E_INTERNAL;
      }
#undef EIGD_SPMM_TILED
      EIGD_LAUNCH_CHECK();
      continue;
    }
#define EIGD_SPMM_CASE(KP)                                                                                          \
  case KP:                                                                                                          \
    hipLaunchKernelGGL(spmm_rows_kernel<KP>, grid, dim3(kThreads), 0, st, A->n, kb, A->indptr, A->indices, A->data, \
                       dX + c0, ldx, dY + c0, ldy, alpha, beta);                                                    \
    break;
    switch (kp) {
      EIGD_SPMM_CASE(2)
      EIGD_SPMM_CASE(4)
      EIGD_SPMM_CASE(8)
      EIGD_SPMM_CASE(16)
      EIGD_SPMM_CASE(32)
      EIGD_SPMM_CASE(64)
      default:
        set_error("internal: unexpected kp=%d", kp);
        return EIGD_E_INTERNAL;
    }
#undef EIGD_SPMM_CASE
    EIGD_LAUNCH_CHECK();
  }
  return EIGD_OK;
}

// y = A z together with the coefficients of a conjugate-gradient step (krylov.hip): the inner products z.r and z.y come
// out of the product's own pass (5 <= k <= 32 on a tiled matrix; other shapes: the product, then eigd_cg_coefficients)
int eigd_spmm_cg(eigd_ctx* ctx, eigd_mat* A, int k, const double* dZ, int ldz, double* dY, int ldy, const double* dR, int ldr,
                 const double* dNorm2, double* dState, int step, int first, double* dLog) {
  EIGD_REQUIRE(ctx && A && dZ && dY && dR && dState, "null argument");
  EIGD_REQUIRE(ctx->device == A->ctx->device, "context and matrix live on different devices");
  EIGD_REQUIRE(k >= 1 && k <= kMaxK && ldz >= k && ldy >= k && ldr >= k && A->n == A->ncols, "bad block shape k=%d", k);
  EIGD_REQUIRE(dZ != dY, "spmm cannot run in place");
  const int kp = std::max(2, next_pow2(k));
  const int cpl = (kp >= 8) ? kp / 8 : 1;
  const size_t tile_lds = sizeof(double) * static_cast<size_t>(A->umax) * (cpl >= 2 ? kp + 4 : kp + 1);
  const bool fused = kp >= 8 && kp <= 32 && A->ntiles > 0 && tile_lds <= static_cast<size_t>(kTileLds);
  if (!fused) {
    int rc = eigd_spmm_on(ctx, A, dZ, ldz, dY, ldy, k, 1.0, 0.0);
    if (rc) return rc;
    return eigd_cg_coefficients(ctx, A->n, k, dZ, ldz, dR, ldr, dY, ldy, dNorm2, dState, step, first, dLog);
  }
  const int groups = std::min(A->ntiles, 1024);
  int rc = ctx->ensure_scratch(sizeof(double) * (static_cast<size_t>(A->ntiles) + groups) * 2 * k);
  if (rc) return rc;
  double* gsum = ctx->scratch;                                    // groups x 2k
  double* dots = ctx->scratch + static_cast<size_t>(groups) * 2 * k;  // ntiles x 2k
  const size_t tile_lds_all = tile_lds + static_cast<size_t>(A->tnz_cap) * (sizeof(double) + sizeof(uint16_t));
  const int per_xcd = (A->ntiles + 7) / 8;
  const dim3 tgrid(per_xcd * 8);
  hipStream_t st = ctx->stream;
#define EIGD_SPMM_DOTS(KP)                                                                                             \
  case KP:                                                                                                             \
    hipLaunchKernelGGL((spmm_tiled_kernel<KP, 8, true>), tgrid, dim3(kThreads), tile_lds_all, st, A->n, k, A->ntiles,  \
                       per_xcd, A->tile_ptr, A->ucols, A->indptr, A->lidx, A->data, dZ, ldz, dY, ldy, 1.0, 0.0,        \
                       A->umax, A->tnz_cap, dR, ldr, dots);                                                            \
    break;
  switch (kp) {
    EIGD_SPMM_DOTS(8)
    EIGD_SPMM_DOTS(16)
    EIGD_SPMM_DOTS(32)
    default:
      set_error("internal: unexpected kp=%d", kp);
      return EIGD_E_INTERNAL;
  }
#undef EIGD_SPMM_DOTS
  EIGD_LAUNCH_CHECK();
  hipLaunchKernelGGL(tile_dots_reduce_kernel, dim3(groups), dim3(kThreads), 0, st, dots, A->ntiles, 2 * k, gsum);
  EIGD_LAUNCH_CHECK();
  return cg_coefficients_from_partials(ctx, gsum, groups, k, dNorm2, dState, step, first, dLog);
}

}  // extern "C"
