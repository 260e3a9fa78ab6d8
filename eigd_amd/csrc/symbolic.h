// Host-side ordering + symbolic multifrontal analysis for the shift-invert factor.
// Replaces the COLAMD/etree/symbolic phase SuperLU runs inside scipy's splu
// (reference eigd/eigenvector_derivatives.py:13).  Pure C++, no device code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace eigd {

struct Symbolic {
  int n = 0;
  int leaf_size = 0;  // max dofs in a leaf subdomain
  int W = 64;         // panel width (columns factored / solved per step)
  int CH = 64;        // row-chunk height of the panel kernels

  std::vector<int> perm, iperm;  // perm[new] = old, iperm[old] = new

  // fronts, numbered in postorder (children before parents; columns consecutive)
  int nfronts = 0;
  std::vector<int> f_c0, f_ns, f_bs, f_parent, f_level, f_slot, f_npanels;
  std::vector<int64_t> f_bptr;   // border list pointers, nfronts + 1
  std::vector<int> border;       // sorted permuted row indices below each front
  std::vector<int> rel;          // position of border[i] in the parent's [cols ; border] numbering
  std::vector<int64_t> f_foff;   // front square (d x d, column major) offset, doubles
  std::vector<int64_t> f_voff;   // front vector offset, rows (sum of d over earlier fronts)
  std::vector<int64_t> f_ioff;   // inverse diagonal blocks offset, doubles (npanels * W * W per front)

  // level schedule: fronts grouped by level, inside a level sorted by npanels descending,
  // so the fronts active in panel step s are a prefix of the level's list
  int nlevels = 0;
  std::vector<int> lvl_ptr, lvl_fronts;
  std::vector<int> lvl_nsteps;            // panel steps of each level
  // (level, step) records
  std::vector<int> ls_ptr;                // nlevels + 1 : index of (level, 0)
  std::vector<int> ls_nactive;            // per (level, step)
  std::vector<int64_t> ls_pref_ptr;       // per (level, step): offset into pref arrays (nactive + 1 entries each)
  std::vector<int> pref_chunks;           // prefix sums of row chunks below the panel
  std::vector<int> pref_tiles;            // prefix sums of chunks^2 (trailing update tiles)
  // extend-add lists: children of the fronts of a level, grouped by child slot
  int maxslots = 0;
  std::vector<int> cs_ptr;                // nlevels * maxslots + 1
  std::vector<int> cs_child;

  // scatter of A (values given in the caller's CSR order) into the front buffer
  int64_t nlower = 0;
  std::vector<int64_t> a_src, a_dst;
  // source row (original numbering) of every front-vector row, -1 for border rows
  std::vector<int> v_src;

  int64_t nnzL = 0, front_doubles = 0, sumd = 0, inv_doubles = 0;
  double flops = 0;
  int maxd = 0, maxns = 0;
  int ncompressed = 0;

  std::string error;
};

// indptr/indices: full symmetric pattern, rows sorted, diagonal present.
// Returns false and sets s.error on failure.
// dof_coords (optional, n x dim, dim <= 3): geometric nested dissection instead of level structures.
bool analyze(int n, const int32_t* indptr, const int32_t* indices, int leaf_size, int panel_width, Symbolic& s,
             int dim = 0, const double* dof_coords = nullptr);

}  // namespace eigd
