// Ordering (nested dissection on the supervariable-compressed graph) and symbolic
// multifrontal analysis.  See symbolic.h.
#include "symbolic.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>

namespace eigd {
namespace {

// ---------------------------------------------------------------------------
// compressed (supervariable) graph
// ---------------------------------------------------------------------------
struct Graph {
  int nv = 0;
  std::vector<int64_t> xadj;
  std::vector<int> adj;
  std::vector<int> vw;                 // dofs per compressed vertex
  std::vector<int> cv_of;              // original dof -> compressed vertex
  std::vector<int> cv_ptr, cv_dofs;    // compressed vertex -> original dofs
};

// rows with identical sorted patterns that are consecutive in the numbering (the dofs of
// one FE node) are indistinguishable vertices of the elimination graph: merge them.
void compress(int n, const int32_t* ip, const int32_t* ix, Graph& g) {
  g.cv_of.assign(n, 0);
  int nv = 0;
  for (int i = 0; i < n; ++i) {
    bool same = false;
    if (i > 0) {
      int64_t la = ip[i] - ip[i - 1], lb = ip[i + 1] - ip[i];
      if (la == lb && std::memcmp(ix + ip[i - 1], ix + ip[i], sizeof(int32_t) * la) == 0) {
        // identical closed neighbourhoods need both diagonals present
        bool hasd = std::binary_search(ix + ip[i], ix + ip[i + 1], i) &&
                    std::binary_search(ix + ip[i], ix + ip[i + 1], i - 1);
        same = hasd;
      }
    }
    if (!same) ++nv;
    g.cv_of[i] = nv - 1;
  }
  g.nv = nv;
  g.vw.assign(nv, 0);
  for (int i = 0; i < n; ++i) g.vw[g.cv_of[i]]++;
  g.cv_ptr.assign(nv + 1, 0);
  for (int v = 0; v < nv; ++v) g.cv_ptr[v + 1] = g.cv_ptr[v] + g.vw[v];
  g.cv_dofs.resize(n);
  {
    std::vector<int> pos(g.cv_ptr.begin(), g.cv_ptr.end() - 1);
    for (int i = 0; i < n; ++i) g.cv_dofs[pos[g.cv_of[i]]++] = i;
  }
  // adjacency of the compressed graph = union over member rows, symmetrised
  std::vector<int> stamp(nv, -1);
  // collect unique neighbours per compressed vertex (from all member rows)
  std::vector<int64_t> xa(nv + 1, 0);
  std::vector<int> ad;
  ad.reserve(static_cast<size_t>(ip[n]) / 2 + 16);
  for (int v = 0; v < nv; ++v) {
    for (int q = g.cv_ptr[v]; q < g.cv_ptr[v + 1]; ++q) {
      int i = g.cv_dofs[q];
      for (int64_t e = ip[i]; e < ip[i + 1]; ++e) {
        int u = g.cv_of[ix[e]];
        if (u != v && stamp[u] != v) {
          stamp[u] = v;
          ad.push_back(u);
        }
      }
    }
    xa[v + 1] = static_cast<int64_t>(ad.size());
  }
  // symmetrise: count reverse edges that are missing
  bool symmetric = true;
  {
    // check u in adj(v) => v in adj(u) using sorted lists
    for (int v = 0; v < nv; ++v) std::sort(ad.begin() + xa[v], ad.begin() + xa[v + 1]);
    for (int v = 0; v < nv && symmetric; ++v)
      for (int64_t e = xa[v]; e < xa[v + 1]; ++e) {
        int u = ad[e];
        if (!std::binary_search(ad.begin() + xa[u], ad.begin() + xa[u + 1], v)) {
          symmetric = false;
          break;
        }
      }
  }
  if (symmetric) {
    g.xadj.swap(xa);
    g.adj.swap(ad);
    return;
  }
  std::vector<int64_t> deg(nv, 0);
  for (int v = 0; v < nv; ++v)
    for (int64_t e = xa[v]; e < xa[v + 1]; ++e) {
      deg[v]++;
      deg[ad[e]]++;
    }
  std::vector<int64_t> xb(nv + 1, 0);
  for (int v = 0; v < nv; ++v) xb[v + 1] = xb[v] + deg[v];
  std::vector<int> bd(xb[nv]);
  std::vector<int64_t> pos(xb.begin(), xb.end() - 1);
  for (int v = 0; v < nv; ++v)
    for (int64_t e = xa[v]; e < xa[v + 1]; ++e) {
      bd[pos[v]++] = ad[e];
      bd[pos[ad[e]]++] = v;
    }
  g.xadj.assign(nv + 1, 0);
  g.adj.clear();
  for (int v = 0; v < nv; ++v) {
    std::sort(bd.begin() + xb[v], bd.begin() + xb[v + 1]);
    auto last = std::unique(bd.begin() + xb[v], bd.begin() + xb[v + 1]);
    g.adj.insert(g.adj.end(), bd.begin() + xb[v], last);
    g.xadj[v + 1] = static_cast<int64_t>(g.adj.size());
  }
}

// ---------------------------------------------------------------------------
// nested dissection by level structures
// ---------------------------------------------------------------------------
struct TreeNode {
  int parent;
  int64_t v0, v1;  // range in node_verts
};

struct ND {
  const Graph& g;
  int leaf_size;
  std::vector<int> region;      // region id of each vertex (-1 = already placed in a separator/leaf)
  std::vector<int> dist;        // BFS level
  std::vector<int> queue;
  std::vector<TreeNode> nodes;
  std::vector<int> node_verts;
  int next_region = 1;
  const double* coords = nullptr;  // optional: dim coordinates per compressed vertex
  int dim = 0;

  ND(const Graph& gg, int leaf) : g(gg), leaf_size(leaf) {
    region.assign(g.nv, 0);
    dist.assign(g.nv, -1);
    queue.reserve(g.nv);
  }

  int add_node(int parent, const int* v, int64_t cnt) {
    TreeNode t;
    t.parent = parent;
    t.v0 = static_cast<int64_t>(node_verts.size());
    node_verts.insert(node_verts.end(), v, v + cnt);
    t.v1 = static_cast<int64_t>(node_verts.size());
    nodes.push_back(t);
    return static_cast<int>(nodes.size()) - 1;
  }

  // BFS inside region rid from root; fills queue (order) and dist; returns number of levels
  int bfs(int root, int rid, std::vector<int>& order) {
    order.clear();
    order.push_back(root);
    dist[root] = 0;
    size_t head = 0;
    int maxd = 0;
    while (head < order.size()) {
      int v = order[head++];
      int dv = dist[v];
      maxd = dv;
      for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
        int u = g.adj[e];
        if (region[u] == rid && dist[u] < 0) {
          dist[u] = dv + 1;
          order.push_back(u);
        }
      }
    }
    return maxd + 1;
  }
  void clear_dist(const std::vector<int>& order) {
    for (int v : order) dist[v] = -1;
  }

  // vertex separator from a rooted level structure (George & Liu): pseudo-peripheral root, lightest
  // level of the middle band, trimmed to the vertices that touch the far side
  bool level_split(const std::vector<int>& comp, int crid, int64_t wsum, std::vector<int>& order,
                   std::vector<int>& sep, std::vector<int>& p1, std::vector<int>& p2) {
    // pseudo-peripheral root
    int root = comp[0];
    int nlev = 0;
    for (int it = 0; it < 6; ++it) {
      int nl = bfs(root, crid, order);
      if (nl <= nlev && it > 0) {
        clear_dist(order);
        break;
      }
      nlev = nl;
      // min-degree vertex of the last level
      int best = order.back();
      int64_t bdeg = g.xadj[best + 1] - g.xadj[best];
      for (size_t q = order.size(); q-- > 0;) {
        int v = order[q];
        if (dist[v] != nl - 1) break;
        int64_t dg = g.xadj[v + 1] - g.xadj[v];
        if (dg < bdeg) {
          bdeg = dg;
          best = v;
        }
      }
      clear_dist(order);
      if (best == root) break;
      root = best;
    }
    nlev = bfs(root, crid, order);
    if (nlev < 3) {  // clique-like: no vertex separator from a level structure
      clear_dist(order);
      return false;
    }
    std::vector<int64_t> lw(nlev, 0);
    for (int v : order) lw[dist[v]] += g.vw[v];
    std::vector<int64_t> cum(nlev + 1, 0);
    for (int l = 0; l < nlev; ++l) cum[l + 1] = cum[l] + lw[l];
    // lightest level inside the middle band, else the level nearest to the midpoint
    int ksep = -1;
    int64_t bestw = -1;
    for (int l = 1; l < nlev - 1; ++l) {
      double mid = (cum[l] + 0.5 * lw[l]) / static_cast<double>(wsum);
      if (mid >= 0.38 && mid <= 0.62 && (ksep < 0 || lw[l] < bestw)) {
        ksep = l;
        bestw = lw[l];
      }
    }
    if (ksep < 0) {
      double bestd = 2.0;
      for (int l = 1; l < nlev - 1; ++l) {
        double mid = std::fabs((cum[l] + 0.5 * lw[l]) / static_cast<double>(wsum) - 0.5);
        if (mid < bestd) {
          bestd = mid;
          ksep = l;
        }
      }
    }
    for (int v : order) {
      int d = dist[v];
      if (d < ksep)
        p1.push_back(v);
      else if (d > ksep)
        p2.push_back(v);
      else {
        bool touches = false;
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1] && !touches; ++e) {
          int u = g.adj[e];
          touches = (region[u] == crid && dist[u] == ksep + 1);
        }
        (touches ? sep : p1).push_back(v);
      }
    }
    clear_dist(order);
    return true;
  }

  // geometric split: cut the longest axis of the bounding box at the weighted median coordinate; the
  // separator is the layer of the low side that touches the high side (a straight mesh line on a
  // structured grid, the optimum for 9-point stencils)
  bool geometric_split(const std::vector<int>& comp, int crid, int64_t wsum, std::vector<int>& sep,
                       std::vector<int>& p1, std::vector<int>& p2) {
    if (coords == nullptr) return false;
    double lo[3], hi[3];
    for (int a = 0; a < dim; ++a) {
      lo[a] = 1e300;
      hi[a] = -1e300;
    }
    for (int v : comp)
      for (int a = 0; a < dim; ++a) {
        const double x = coords[static_cast<int64_t>(v) * dim + a];
        lo[a] = std::min(lo[a], x);
        hi[a] = std::max(hi[a], x);
      }
    int axes[3] = {0, 1, 2};
    std::sort(axes, axes + dim, [&](int a, int b) { return (hi[a] - lo[a]) > (hi[b] - lo[b]); });
    std::vector<std::pair<double, int>> key(comp.size());
    for (int t = 0; t < dim; ++t) {
      const int ax = axes[t];
      if (!(hi[ax] > lo[ax])) break;
      for (size_t q = 0; q < comp.size(); ++q) key[q] = {coords[static_cast<int64_t>(comp[q]) * dim + ax], comp[q]};
      std::sort(key.begin(), key.end());
      // weighted median, then move the cut to the end of the run of equal coordinates (a whole grid line)
      int64_t acc = 0;
      size_t cut = 0;
      while (cut < key.size() && 2 * acc < wsum) acc += g.vw[key[cut++].second];
      const double tol = 1e-9 * (hi[ax] - lo[ax]);
      while (cut < key.size() && key[cut].first <= key[cut - 1].first + tol) ++cut;
      if (cut == 0 || cut >= key.size()) continue;
      // mark side: dist = 0 low side, 1 high side
      for (size_t q = 0; q < key.size(); ++q) dist[key[q].second] = (q < cut) ? 0 : 1;
      sep.clear();
      p1.clear();
      p2.clear();
      for (size_t q = 0; q < key.size(); ++q) {
        const int v = key[q].second;
        if (q >= cut) {
          p2.push_back(v);
          continue;
        }
        bool touches = false;
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1] && !touches; ++e) {
          const int u = g.adj[e];
          touches = (region[u] == crid && dist[u] == 1);
        }
        (touches ? sep : p1).push_back(v);
      }
      for (size_t q = 0; q < key.size(); ++q) dist[key[q].second] = -1;
      if (!sep.empty() && !p1.empty() && !p2.empty()) return true;
    }
    return false;
  }

  void run() {
    struct Task {
      std::vector<int> verts;
      int parent;
      int depth = 0;  // merged dissection rounds above this task
    };
    std::vector<Task> stack;
    {
      Task t;
      t.verts.resize(g.nv);
      std::iota(t.verts.begin(), t.verts.end(), 0);
      t.parent = -1;
      stack.push_back(std::move(t));
    }
    std::vector<int> order, comp;
    while (!stack.empty()) {
      Task task = std::move(stack.back());
      stack.pop_back();
      int rid = next_region++;
      for (int v : task.verts) region[v] = rid;
      // split into connected components
      for (size_t s = 0; s < task.verts.size(); ++s) {
        int seed = task.verts[s];
        if (region[seed] != rid) continue;  // already consumed by an earlier component
        bfs(seed, rid, comp);
        clear_dist(comp);
        int crid = next_region++;
        int64_t wsum = 0;
        for (int v : comp) {
          region[v] = crid;
          wsum += g.vw[v];
        }
        if (wsum <= leaf_size || comp.size() < 3) {
          add_node(task.parent, comp.data(), static_cast<int64_t>(comp.size()));
          for (int v : comp) region[v] = -1;
          continue;
        }
        std::vector<int> sep, p1, p2;
        if (!geometric_split(comp, crid, wsum, sep, p1, p2)) {
          sep.clear();
          p1.clear();
          p2.clear();
          if (!level_split(comp, crid, wsum, order, sep, p1, p2)) {
            add_node(task.parent, comp.data(), static_cast<int64_t>(comp.size()));
            for (int v : comp) region[v] = -1;
            continue;
          }
        }
        if (sep.empty() || p1.empty() || p2.empty()) {
          add_node(task.parent, comp.data(), static_cast<int64_t>(comp.size()));
          for (int v : comp) region[v] = -1;
          continue;
        }
        for (int v : sep) region[v] = -1;
        // give the sub-regions private ids now so the component scan above skips them
        int r1 = next_region++, r2 = next_region++;
        for (int v : p1) region[v] = r1;
        for (int v : p2) region[v] = r2;
        std::vector<std::vector<int>> parts;
        parts.push_back(std::move(p1));
        parts.push_back(std::move(p2));
        int snode = add_node(task.parent, sep.data(), static_cast<int64_t>(sep.size()));
        for (auto& pv : parts) {
          Task t;
          t.parent = snode;
          t.depth = task.depth + 1;
          t.verts.swap(pv);
          stack.push_back(std::move(t));
        }
      }
    }
  }
};

}  // namespace

// ---------------------------------------------------------------------------
bool analyze(int n, const int32_t* ip, const int32_t* ix, int leaf_size, int panel_width, Symbolic& s, int dim,
             const double* dof_coords) {
  s = Symbolic();
  s.n = n;
  s.leaf_size = leaf_size > 0 ? leaf_size : 64;
  s.W = panel_width > 0 ? panel_width : 64;
  if (s.W > 64 || s.W < 8) {
    s.error = "panel width must be in [8, 64]";
    return false;
  }
  if (n <= 0) {
    s.error = "empty matrix";
    return false;
  }
  for (int i = 0; i < n; ++i) {
    if (ip[i + 1] < ip[i]) {
      s.error = "indptr not monotone";
      return false;
    }
    for (int64_t e = ip[i]; e < ip[i + 1]; ++e) {
      if (ix[e] < 0 || ix[e] >= n || (e > ip[i] && ix[e] <= ix[e - 1])) {
        s.error = "indices must be sorted, unique and in range";
        return false;
      }
    }
    if (!std::binary_search(ix + ip[i], ix + ip[i + 1], i)) {
      s.error = "structural diagonal entry missing";
      return false;
    }
  }

  Graph g;
  compress(n, ip, ix, g);
  s.ncompressed = g.nv;
  ND nd(g, s.leaf_size);
  std::vector<double> vcoords;
  if (dof_coords != nullptr && dim >= 1 && dim <= 3) {  // a compressed vertex sits where its first dof sits
    vcoords.resize(static_cast<size_t>(g.nv) * dim);
    for (int v = 0; v < g.nv; ++v)
      for (int a = 0; a < dim; ++a) vcoords[static_cast<size_t>(v) * dim + a] = dof_coords[static_cast<size_t>(g.cv_dofs[g.cv_ptr[v]]) * dim + a];
    nd.coords = vcoords.data();
    nd.dim = dim;
  }
  nd.run();

  // ---- postorder of the separator tree -> fronts and permutation ----------
  const int nn = static_cast<int>(nd.nodes.size());
  std::vector<int> nchild(nn, 0), cptr(nn + 1, 0);
  for (int t = 0; t < nn; ++t)
    if (nd.nodes[t].parent >= 0) nchild[nd.nodes[t].parent]++;
  for (int t = 0; t < nn; ++t) cptr[t + 1] = cptr[t] + nchild[t];
  std::vector<int> clist(cptr[nn]), cpos(cptr.begin(), cptr.end() - 1);
  std::vector<int> roots;
  for (int t = 0; t < nn; ++t) {
    int p = nd.nodes[t].parent;
    if (p >= 0)
      clist[cpos[p]++] = t;
    else
      roots.push_back(t);
  }
  std::vector<int> post;  // tree nodes in postorder
  post.reserve(nn);
  {
    std::vector<std::pair<int, int>> st;
    for (int r : roots) {
      st.push_back({r, 0});
      while (!st.empty()) {
        auto& top = st.back();
        int t = top.first;
        if (top.second < nchild[t]) {
          int c = clist[cptr[t] + top.second++];
          st.push_back({c, 0});
        } else {
          post.push_back(t);
          st.pop_back();
        }
      }
    }
  }
  s.nfronts = nn;
  std::vector<int> fid(nn);
  for (int q = 0; q < nn; ++q) fid[post[q]] = q;
  s.f_c0.resize(nn);
  s.f_ns.resize(nn);
  s.f_parent.resize(nn);
  s.perm.resize(n);
  s.iperm.resize(n);
  {
    int next = 0;
    for (int q = 0; q < nn; ++q) {
      const TreeNode& t = nd.nodes[post[q]];
      s.f_c0[q] = next;
      for (int64_t k = t.v0; k < t.v1; ++k) {
        int cv = nd.node_verts[k];
        for (int z = g.cv_ptr[cv]; z < g.cv_ptr[cv + 1]; ++z) s.perm[next++] = g.cv_dofs[z];
      }
      s.f_ns[q] = next - s.f_c0[q];
      s.f_parent[q] = t.parent >= 0 ? fid[t.parent] : -1;
    }
    if (next != n) {
      s.error = "internal: ordering does not cover all rows";
      return false;
    }
    for (int i = 0; i < n; ++i) s.iperm[s.perm[i]] = i;
  }
  std::vector<int> colfront(n);
  for (int f = 0; f < nn; ++f)
    for (int c = s.f_c0[f]; c < s.f_c0[f] + s.f_ns[f]; ++c) colfront[c] = f;

  // children lists in front numbering (ordered by front id)
  std::vector<int> fch_ptr(nn + 1, 0);
  for (int f = 0; f < nn; ++f)
    if (s.f_parent[f] >= 0) fch_ptr[s.f_parent[f] + 1]++;
  for (int f = 0; f < nn; ++f) fch_ptr[f + 1] += fch_ptr[f];
  std::vector<int> fch(fch_ptr[nn]);
  s.f_slot.assign(nn, 0);
  {
    std::vector<int> pos(fch_ptr.begin(), fch_ptr.end() - 1);
    for (int f = 0; f < nn; ++f) {
      int p = s.f_parent[f];
      if (p >= 0) {
        s.f_slot[f] = pos[p] - fch_ptr[p];
        fch[pos[p]++] = f;
      }
    }
  }

  // ---- borders (row structure below each front) ----------------------------
  s.f_bptr.assign(nn + 1, 0);
  s.f_bs.assign(nn, 0);
  s.border.clear();
  {
    std::vector<int> stamp(n, -1), tmp;
    for (int f = 0; f < nn; ++f) {
      tmp.clear();
      const int c0 = s.f_c0[f], c1 = c0 + s.f_ns[f];
      for (int c = c0; c < c1; ++c) {
        int io = s.perm[c];
        for (int64_t e = ip[io]; e < ip[io + 1]; ++e) {
          int r = s.iperm[ix[e]];
          if (r >= c1 && stamp[r] != f) {
            stamp[r] = f;
            tmp.push_back(r);
          }
        }
      }
      for (int q = fch_ptr[f]; q < fch_ptr[f + 1]; ++q) {
        int c = fch[q];
        for (int64_t e = s.f_bptr[c]; e < s.f_bptr[c + 1]; ++e) {
          int r = s.border[e];
          if (r >= c1 && stamp[r] != f) {
            stamp[r] = f;
            tmp.push_back(r);
          } else if (r < c0) {
            s.error = "internal: child border reaches below its parent (separator property violated)";
            return false;
          }
        }
      }
      std::sort(tmp.begin(), tmp.end());
      s.border.insert(s.border.end(), tmp.begin(), tmp.end());
      s.f_bptr[f + 1] = static_cast<int64_t>(s.border.size());
      s.f_bs[f] = static_cast<int>(tmp.size());
      if (s.f_parent[f] < 0 && !tmp.empty()) {
        s.error = "internal: root front with a border";
        return false;
      }
    }
  }
  // relative indices into the parent's [cols ; border] numbering
  s.rel.assign(s.border.size(), -1);
  for (int f = 0; f < nn; ++f) {
    int p = s.f_parent[f];
    if (p < 0) continue;
    const int pc0 = s.f_c0[p], pns = s.f_ns[p];
    const int* pb = s.border.data() + s.f_bptr[p];
    const int pbs = s.f_bs[p];
    int cur = 0;
    for (int64_t e = s.f_bptr[f]; e < s.f_bptr[f + 1]; ++e) {
      int r = s.border[e];
      if (r < pc0) {
        s.error = "internal: border row below the parent's columns";
        return false;
      }
      if (r < pc0 + pns) {
        s.rel[e] = r - pc0;
      } else {
        while (cur < pbs && pb[cur] < r) ++cur;
        if (cur >= pbs || pb[cur] != r) {
          s.error = "internal: child border not contained in the parent front";
          return false;
        }
        s.rel[e] = pns + cur;
      }
    }
  }

  // ---- levels, panels, offsets -----------------------------------------------
  s.f_level.assign(nn, 0);
  for (int f = 0; f < nn; ++f) {  // postorder: children first
    int p = s.f_parent[f];
    if (p >= 0) s.f_level[p] = std::max(s.f_level[p], s.f_level[f] + 1);
  }
  s.nlevels = 0;
  for (int f = 0; f < nn; ++f) s.nlevels = std::max(s.nlevels, s.f_level[f] + 1);
  s.f_npanels.resize(nn);
  s.f_foff.resize(nn);
  s.f_voff.resize(nn);
  s.f_ioff.resize(nn);
  s.maxslots = 0;
  {
    int64_t fo = 0, vo = 0, io = 0;
    for (int f = 0; f < nn; ++f) {
      const int64_t ns = s.f_ns[f], bs = s.f_bs[f], d = ns + bs;
      s.f_npanels[f] = static_cast<int>((ns + s.W - 1) / s.W);
      s.f_foff[f] = fo;
      s.f_voff[f] = vo;
      s.f_ioff[f] = io;
      fo += d * d;
      vo += d;
      io += static_cast<int64_t>(s.f_npanels[f]) * s.W * s.W;
      s.nnzL += ns * (ns + 1) / 2 + ns * bs;
      s.flops += static_cast<double>(ns) * ns * ns / 3.0 + static_cast<double>(ns) * ns * bs +
                 static_cast<double>(ns) * bs * bs;
      s.maxd = std::max<int>(s.maxd, static_cast<int>(d));
      s.maxns = std::max<int>(s.maxns, static_cast<int>(ns));
      s.maxslots = std::max(s.maxslots, fch_ptr[f + 1] - fch_ptr[f]);
    }
    s.front_doubles = fo;
    s.sumd = vo;
    s.inv_doubles = io;
  }
  s.lvl_ptr.assign(s.nlevels + 1, 0);
  for (int f = 0; f < nn; ++f) s.lvl_ptr[s.f_level[f] + 1]++;
  for (int l = 0; l < s.nlevels; ++l) s.lvl_ptr[l + 1] += s.lvl_ptr[l];
  s.lvl_fronts.resize(nn);
  {
    std::vector<int> pos(s.lvl_ptr.begin(), s.lvl_ptr.end() - 1);
    for (int f = 0; f < nn; ++f) s.lvl_fronts[pos[s.f_level[f]]++] = f;
    for (int l = 0; l < s.nlevels; ++l)
      std::stable_sort(s.lvl_fronts.begin() + s.lvl_ptr[l], s.lvl_fronts.begin() + s.lvl_ptr[l + 1],
                       [&](int a, int b) { return s.f_npanels[a] > s.f_npanels[b]; });
  }
  s.lvl_nsteps.assign(s.nlevels, 0);
  s.ls_ptr.assign(s.nlevels + 1, 0);
  for (int l = 0; l < s.nlevels; ++l) {
    s.lvl_nsteps[l] = s.f_npanels[s.lvl_fronts[s.lvl_ptr[l]]];
    s.ls_ptr[l + 1] = s.ls_ptr[l] + s.lvl_nsteps[l];
  }
  const int nls = s.ls_ptr[s.nlevels];
  s.ls_nactive.assign(nls, 0);
  s.ls_pref_ptr.assign(nls + 1, 0);
  s.pref_chunks.clear();
  s.pref_tiles.clear();
  for (int l = 0; l < s.nlevels; ++l) {
    for (int st = 0; st < s.lvl_nsteps[l]; ++st) {
      const int rec = s.ls_ptr[l] + st;
      int na = 0;
      s.pref_chunks.push_back(0);
      s.pref_tiles.push_back(0);
      for (int q = s.lvl_ptr[l]; q < s.lvl_ptr[l + 1]; ++q) {
        int f = s.lvl_fronts[q];
        if (s.f_npanels[f] <= st) break;
        ++na;
        const int ns = s.f_ns[f], d = ns + s.f_bs[f];
        const int j1 = std::min(ns, (st + 1) * s.W);
        const int nch = (d - j1 + s.CH - 1) / s.CH;
        int64_t pc = static_cast<int64_t>(s.pref_chunks.back()) + nch;
        int64_t pt = static_cast<int64_t>(s.pref_tiles.back()) + static_cast<int64_t>(nch) * nch;
        if (pc > INT32_MAX || pt > INT32_MAX) {
          s.error = "level too large for 32-bit tile indices";
          return false;
        }
        s.pref_chunks.push_back(static_cast<int>(pc));
        s.pref_tiles.push_back(static_cast<int>(pt));
      }
      s.ls_nactive[rec] = na;
      s.ls_pref_ptr[rec + 1] = static_cast<int64_t>(s.pref_chunks.size());
    }
  }
  // extend-add lists by (level of the parent, slot of the child)
  if (s.maxslots < 1) s.maxslots = 1;
  s.cs_ptr.assign(static_cast<size_t>(s.nlevels) * s.maxslots + 1, 0);
  for (int f = 0; f < nn; ++f) {
    int p = s.f_parent[f];
    if (p >= 0) s.cs_ptr[static_cast<size_t>(s.f_level[p]) * s.maxslots + s.f_slot[f] + 1]++;
  }
  for (size_t q = 0; q + 1 < s.cs_ptr.size(); ++q) s.cs_ptr[q + 1] += s.cs_ptr[q];
  s.cs_child.resize(s.cs_ptr.back());
  {
    std::vector<int> pos(s.cs_ptr.begin(), s.cs_ptr.end() - 1);
    for (int f = 0; f < nn; ++f) {
      int p = s.f_parent[f];
      if (p >= 0) s.cs_child[pos[static_cast<size_t>(s.f_level[p]) * s.maxslots + s.f_slot[f]]++] = f;
    }
  }

  // ---- scatter map of A into the fronts ----------------------------------------
  s.a_src.clear();
  s.a_dst.clear();
  s.a_src.reserve(static_cast<size_t>(ip[n]) / 2 + n);
  s.a_dst.reserve(static_cast<size_t>(ip[n]) / 2 + n);
  for (int io = 0; io < n; ++io) {
    const int ni = s.iperm[io];
    for (int64_t e = ip[io]; e < ip[io + 1]; ++e) {
      const int nj = s.iperm[ix[e]];
      if (ni < nj) continue;  // keep the entry whose permuted position is in the lower triangle
      const int f = colfront[nj];
      const int c0 = s.f_c0[f], ns = s.f_ns[f];
      const int64_t d = ns + s.f_bs[f];
      int64_t lr;
      if (ni < c0 + ns) {
        lr = ni - c0;
      } else {
        const int* b0 = s.border.data() + s.f_bptr[f];
        const int* b1 = s.border.data() + s.f_bptr[f + 1];
        const int* it = std::lower_bound(b0, b1, ni);
        if (it == b1 || *it != ni) {
          s.error = "internal: matrix entry outside the front structure";
          return false;
        }
        lr = ns + (it - b0);
      }
      s.a_src.push_back(e);
      s.a_dst.push_back(s.f_foff[f] + static_cast<int64_t>(nj - c0) * d + lr);
    }
  }
  s.nlower = static_cast<int64_t>(s.a_src.size());

  s.v_src.assign(s.sumd, -1);
  for (int f = 0; f < nn; ++f)
    for (int r = 0; r < s.f_ns[f]; ++r) s.v_src[s.f_voff[f] + r] = s.perm[s.f_c0[f] + r];
  return true;
}

}  // namespace eigd
