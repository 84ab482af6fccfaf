// direct.cpp -- plan of the exact sparse block Cholesky (see direct.hpp).  Host only.
#include "direct.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>

namespace sim3opt {
namespace {

// ---- nested dissection on the block graph: separators are BFS level sets ----
struct Dissector {
  const std::vector<int32_t>& aptr;
  const std::vector<int32_t>& adj;
  std::vector<int32_t> tag;    // subset membership stamp
  std::vector<int32_t> lev;    // BFS level inside the current subset
  std::vector<int32_t> order;  // result: elimination order
  int32_t stamp = 0;
  static constexpr int kLeaf = 4;

  Dissector(const std::vector<int32_t>& p, const std::vector<int32_t>& a, int32_t nb)
      : aptr(p), adj(a), tag(nb, 0), lev(nb, -1) {
    order.reserve(nb);
  }

  // BFS from s over vertices tagged `t`; fills lev, returns visit order
  void bfs(int32_t s, int32_t t, std::vector<int32_t>& out) {
    out.clear();
    out.push_back(s);
    lev[s] = 0;
    for (size_t h = 0; h < out.size(); ++h) {
      const int32_t u = out[h];
      for (int32_t k = aptr[u]; k < aptr[u + 1]; ++k) {
        const int32_t w = adj[k];
        if (tag[w] == t && lev[w] < 0) {
          lev[w] = lev[u] + 1;
          out.push_back(w);
        }
      }
    }
  }

  void leaf(std::vector<int32_t>& verts) {
    std::sort(verts.begin(), verts.end());
    order.insert(order.end(), verts.begin(), verts.end());
  }

  void dissect(std::vector<int32_t>& verts) {
    if (verts.empty()) return;
    if ((int)verts.size() <= kLeaf) return leaf(verts);
    const int32_t t = ++stamp;
    for (int32_t v : verts) { tag[v] = t; lev[v] = -1; }
    // connected components first
    std::vector<int32_t> comp;
    bfs(verts[0], t, comp);
    if (comp.size() < verts.size()) {
      std::vector<std::vector<int32_t>> parts;
      parts.push_back(comp);
      for (int32_t v : verts)
        if (lev[v] < 0) {
          bfs(v, t, comp);
          parts.push_back(comp);
        }
      for (auto& p : parts) dissect(p);  // (each call re-tags its own subset)
      return;
    }
    // pseudo-peripheral start: a few sweeps to the far end
    int32_t s = comp.back();
    for (int sweep = 0; sweep < 3; ++sweep) {
      for (int32_t v : verts) lev[v] = -1;
      bfs(s, t, comp);
      s = comp.back();
    }
    for (int32_t v : verts) lev[v] = -1;
    bfs(s, t, comp);
    const int32_t nl = lev[comp.back()] + 1;
    if (nl < 3) return leaf(verts);  // clique-like: nothing to separate
    std::vector<int32_t> cnt(nl, 0);
    for (int32_t v : verts) ++cnt[lev[v]];
    // separator = the smallest level whose two sides both hold >= 1/4 of the vertices
    // (ties: the better balanced one); if none qualifies, the median level
    const int64_t n = (int64_t)verts.size();
    int32_t best = -1;
    int64_t best_cnt = 0, best_bal = 0, below = cnt[0];
    for (int32_t m = 1; m + 1 < nl; ++m) {
      const int64_t above = n - below - cnt[m];
      const int64_t bal = std::min(below, above);
      if (4 * bal >= n && (best < 0 || cnt[m] < best_cnt || (cnt[m] == best_cnt && bal > best_bal))) {
        best = m;
        best_cnt = cnt[m];
        best_bal = bal;
      }
      below += cnt[m];
    }
    if (best < 0) {
      int64_t cum = 0;
      best = 1;
      for (int32_t m = 0; m < nl; ++m) {
        cum += cnt[m];
        if (2 * cum >= n) { best = std::min(std::max(m, 1), nl - 2); break; }
      }
    }
    std::vector<int32_t> A, B, S;
    for (int32_t v : verts) {
      if (lev[v] < best) A.push_back(v);
      else if (lev[v] > best) B.push_back(v);
      else {
        // a separator vertex without a neighbour on the far side belongs to the near side
        bool far = false;
        for (int32_t k = aptr[v]; k < aptr[v + 1] && !far; ++k)
          far = tag[adj[k]] == t && lev[adj[k]] > best;
        if (far) S.push_back(v); else A.push_back(v);
      }
    }
    dissect(A);
    dissect(B);
    std::sort(S.begin(), S.end());
    order.insert(order.end(), S.begin(), S.end());
  }
};

// column structures of L for a given elimination order (rows > j, ascending) + elimination tree
bool symbolic(int32_t nb, const std::vector<int32_t>& aptr, const std::vector<int32_t>& adj,
              const std::vector<int32_t>& order, int64_t max_pairs,
              std::vector<std::vector<int32_t>>& cols, std::vector<int32_t>& parent, int64_t& npairs) {
  std::vector<int32_t> pos(nb);
  for (int32_t j = 0; j < nb; ++j) pos[order[j]] = j;
  cols.assign(nb, {});
  parent.assign(nb, -1);
  std::vector<std::vector<int32_t>> kids(nb);
  std::vector<int32_t> mark(nb, -1);
  npairs = 0;
  for (int32_t j = 0; j < nb; ++j) {
    std::vector<int32_t>& s = cols[j];
    const int32_t v = order[j];
    for (int32_t k = aptr[v]; k < aptr[v + 1]; ++k) {
      const int32_t i = pos[adj[k]];
      if (i > j && mark[i] != j) { mark[i] = j; s.push_back(i); }
    }
    for (int32_t c : kids[j])
      for (int32_t i : cols[c])
        if (i > j && mark[i] != j) { mark[i] = j; s.push_back(i); }
    std::sort(s.begin(), s.end());
    if (!s.empty()) {
      parent[j] = s[0];
      kids[s[0]].push_back(j);
    }
    const int64_t c = (int64_t)s.size();
    npairs += c * (c + 1) / 2;
    if (npairs > max_pairs) return false;
  }
  return true;
}

}  // namespace

bool build_direct_plan(int32_t nb, const int32_t* rowptr, const int32_t* colidx, int64_t max_pairs,
                       int32_t subtree_cols, DirectPlan& P, std::string& why, int32_t sub_waves) {
  P = DirectPlan();
  P.sub_waves = std::max(1, std::min((int)DirectPlan::CELL_WAVES, (int)sub_waves));
  if (nb <= 0) { why = "empty system"; return false; }
  // adjacency without the diagonal and without repeated columns
  std::vector<int32_t> aptr(nb + 1, 0), adj;
  adj.reserve((size_t)rowptr[nb]);
  for (int32_t i = 0; i < nb; ++i) {
    const size_t a0 = adj.size();
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (colidx[k] != i) adj.push_back(colidx[k]);
    std::sort(adj.begin() + a0, adj.end());
    adj.erase(std::unique(adj.begin() + a0, adj.end()), adj.end());
    aptr[i + 1] = (int32_t)adj.size();
  }
  Dissector D(aptr, adj, nb);
  {
    std::vector<int32_t> all(nb);
    std::iota(all.begin(), all.end(), 0);
    D.dissect(all);
  }
  if ((int32_t)D.order.size() != nb) { why = "internal: ordering lost vertices"; return false; }
  std::vector<std::vector<int32_t>> cols;
  std::vector<int32_t> parent;
  int64_t npairs = 0;
  if (!symbolic(nb, aptr, adj, D.order, max_pairs, cols, parent, npairs)) {
    why = "fill too large for the direct solver";
    return false;
  }
  // ---- groups: bottom subtrees of <= tau columns are independent; the rest is the top ----
  // (automatic: a sixteenth of the columns, at most 64 -- a level costs a bottom group half of what it costs
  // the top group, whose early levels are wide; KITTI-00: 48 columns, 215 / 387 us per solve against 256 / 465 us
  // with 16 columns and four wavefronts per group; profiles/r3_direct_sweep.log)
  if (subtree_cols <= 0)
    if (const char* ev = std::getenv("SIM3OPT_DIRECT_SUBTREE")) subtree_cols = std::atoi(ev);  // tuning knob
  const int32_t tau = subtree_cols > 0 ? subtree_cols : std::max(4, std::min(64, nb / 16));
  std::vector<int32_t> size(nb, 1);
  for (int32_t j = 0; j < nb; ++j)
    if (parent[j] >= 0) size[parent[j]] += size[j];
  std::vector<int32_t> group(nb, -1), level(nb, 0);
  // columns (positions of the dissection order) sorted by (group, level, position) for subtrees of <= t columns
  auto schedule_order = [&](int32_t t, std::vector<int32_t>& idx) {
    group.assign(nb, -1);
    level.assign(nb, 0);
    const bool single = nb <= 2 * t;
    int32_t ng = 0;
    if (!single) {
      // roots of bottom subtrees in elimination order; consecutive ones share a group up to t columns
      int32_t fill = 0;
      for (int32_t j = 0; j < nb; ++j) {
        if (size[j] > t) continue;
        const int32_t p = parent[j];
        if (p >= 0 && size[p] <= t) continue;  // not a subtree root
        if (fill == 0 || fill + size[j] > t) { ++ng; fill = 0; }
        group[j] = ng - 1;
        fill += size[j];
      }
      // descendants inherit the group of their subtree root (parents have larger indices)
      for (int32_t j = nb - 1; j >= 0; --j)
        if (group[j] < 0 && size[j] <= t) group[j] = group[parent[j]];
    }
    const int32_t top = ng;  // the last group
    for (int32_t j = 0; j < nb; ++j)
      if (group[j] < 0) group[j] = top;
    for (int32_t j = 0; j < nb; ++j) {  // level = height among children of the same group
      const int32_t p = parent[j];
      if (p >= 0 && group[p] == group[j]) level[p] = std::max(level[p], level[j] + 1);
    }
    idx.resize(nb);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
      if (group[a] != group[b]) return group[a] < group[b];
      return level[a] < level[b];
    });
  };
  // The ORDER OF SUMMATION (products of a block, rows of a backward-solve column) is that of a reference
  // numbering -- the schedule of round 2, subtrees of nb / 48 columns -- whatever schedule runs: every
  // schedule is a topological order of the same elimination tree, so the blocks of L are the same numbers
  // and, summed in one fixed order, the same bits.  Tuning the schedule does not move a chaotic LM run.
  std::vector<int32_t> idx_ref, idx;
  schedule_order(std::max(4, nb / 48), idx_ref);
  std::vector<int32_t> refpos(nb);  // dissection position -> reference position
  for (int32_t r = 0; r < nb; ++r) refpos[idx_ref[r]] = r;
  // ---- renumber by (group, level, position): every level becomes a contiguous column range ----
  schedule_order(tau, idx);
  std::vector<int32_t> key(nb);  // final column -> reference position
  for (int32_t j = 0; j < nb; ++j) key[j] = refpos[idx[j]];
  P.nb = nb;
  P.perm.resize(nb);
  for (int32_t j = 0; j < nb; ++j) P.perm[j] = D.order[idx[j]];
  P.gptr.assign(1, 0);
  P.lcolp.clear();
  for (int32_t j = 0; j < nb; ++j) {
    const int32_t o = idx[j];
    const bool new_group = j == 0 || group[o] != group[idx[j - 1]];
    if (new_group && j > 0) P.gptr.push_back((int32_t)P.lcolp.size());
    if (new_group || level[o] != level[idx[j - 1]]) P.lcolp.push_back(j);
  }
  P.gptr.push_back((int32_t)P.lcolp.size());
  P.lcolp.push_back(nb);
  // ---- structure in the final numbering ----
  if (!symbolic(nb, aptr, adj, P.perm, max_pairs, cols, parent, npairs)) {
    why = "fill too large for the direct solver";
    return false;
  }
  P.npairs = npairs;
  P.colptr.assign(nb + 1, 0);
  for (int32_t j = 0; j < nb; ++j) P.colptr[j + 1] = P.colptr[j] + 1 + (int32_t)cols[j].size();
  P.nL = P.colptr[nb];
  P.lrow.resize(P.nL);
  P.lcol.resize(P.nL);
  for (int32_t j = 0; j < nb; ++j) {
    int32_t s = P.colptr[j];
    P.lrow[s] = j;
    P.lcol[s++] = j;
    for (int32_t i : cols[j]) { P.lrow[s] = i; P.lcol[s++] = j; }
  }
  auto slot = [&](int32_t i, int32_t j) -> int32_t {  // block (i, j), i >= j; -1 if not stored
    if (i == j) return P.colptr[j];
    const std::vector<int32_t>& c = cols[j];
    auto it = std::lower_bound(c.begin(), c.end(), i);
    if (it == c.end() || *it != i) return -1;
    return P.colptr[j] + 1 + (int32_t)(it - c.begin());
  };
  // the schedule must respect the dependencies: L(j,k) != 0  =>  k is in an earlier level of j's
  // group, or j is in the top group and k is not
  {
    std::vector<int32_t> lev_of(nb), grp_of(nb);
    for (int32_t g = 0; g < P.ngroups(); ++g)
      for (int32_t l = P.gptr[g]; l < P.gptr[g + 1]; ++l)
        for (int32_t j = P.lcolp[l]; j < P.lcolp[l + 1]; ++j) { lev_of[j] = l; grp_of[j] = g; }
    for (int32_t k = 0; k < nb; ++k)
      for (int32_t j : cols[k]) {
        const bool ok = (grp_of[j] == grp_of[k] && lev_of[k] < lev_of[j]) ||
                        (grp_of[j] == P.ngroups() - 1 && grp_of[k] != grp_of[j]);
        if (!ok) { why = "internal: schedule violates a dependency"; return false; }
      }
    std::vector<int32_t> h(nb, 0);
    for (int32_t j = 0; j < nb; ++j) {
      if (parent[j] >= 0) h[parent[j]] = std::max(h[parent[j]], h[j] + 1);
      P.height = std::max(P.height, h[j] + 1);
    }
  }
  // ---- where the entries of H go: block (a, c) of H is L-slot (pos a, pos c) when pos a >= pos c ----
  std::vector<int32_t> pos(nb);
  for (int32_t j = 0; j < nb; ++j) pos[P.perm[j]] = j;
  P.srcptr.assign(P.nL + 1, 0);
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<int32_t> cur;
    if (pass == 1) {
      for (int64_t s = 0; s < P.nL; ++s) P.srcptr[s + 1] += P.srcptr[s];
      P.src.resize(P.srcptr[P.nL]);
      cur.assign(P.srcptr.begin(), P.srcptr.end() - 1);
    }
    for (int32_t j = 0; j < nb; ++j) {  // slots of a row are visited in ascending block index
      const int32_t a = P.perm[j];
      for (int32_t k = rowptr[a]; k < rowptr[a + 1]; ++k) {
        const int32_t c = colidx[k];
        if (c == a && k != rowptr[a]) continue;  // (a self-loop would not be a pose-graph edge)
        const int32_t jc = pos[c];
        if (jc > j) continue;
        const int32_t s = slot(j, jc);
        if (s < 0) { why = "internal: entry of H outside the pattern of L"; return false; }
        if (pass == 0) ++P.srcptr[s + 1];
        else P.src[cur[s]++] = k;
      }
    }
  }
  // ---- update lists: column k contributes L(i,k) L(j,k)^T to every block (i,j), j <= i in its rows ----
  std::vector<int32_t> by_key(nb);
  for (int32_t j = 0; j < nb; ++j) by_key[key[j]] = j;
  P.pairptr.assign(P.nL + 1, 0);
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<int32_t> cur;
    if (pass == 1) {
      for (int64_t s = 0; s < P.nL; ++s) P.pairptr[s + 1] += P.pairptr[s];
      P.pa.resize(P.pairptr[P.nL]);
      P.pb.resize(P.pairptr[P.nL]);
      P.pcol.resize(P.pairptr[P.nL]);
      cur.assign(P.pairptr.begin(), P.pairptr.end() - 1);
    }
    for (int32_t kr = 0; kr < nb; ++kr) {  // (ascending reference position of the source column)
      const int32_t k = by_key[kr];
      const std::vector<int32_t>& c = cols[k];
      const int32_t base = P.colptr[k] + 1;
      for (size_t a = 0; a < c.size(); ++a)
        for (size_t b = a; b < c.size(); ++b) {
          const int32_t s = slot(c[b], c[a]);
          if (s < 0) { why = "internal: update outside the pattern of L"; return false; }
          if (pass == 0) ++P.pairptr[s + 1];
          else {
            P.pa[cur[s]] = base + (int32_t)b;
            P.pb[cur[s]] = base + (int32_t)a;
            P.pcol[cur[s]] = k;
            ++cur[s];
          }
        }
    }
  }
  // ---- what the top group reads of the bottom groups' results (its warm-up touches them once, in bulk) ----
  if (P.ngroups() > 1) {
    const int32_t c0 = P.lcolp[P.gptr[P.ngroups() - 1]];  // first column of the top group
    const int32_t s0 = P.colptr[c0];
    for (int64_t k = P.pairptr[s0]; k < P.pairptr[P.nL]; ++k) {
      if (P.pa[k] < s0) P.tpre.push_back(P.pa[k]);
      if (P.pb[k] < s0) P.tpre.push_back(P.pb[k]);
      if (P.pcol[k] < c0) P.tprey.push_back(P.pcol[k]);
    }
    std::sort(P.tpre.begin(), P.tpre.end());
    P.tpre.erase(std::unique(P.tpre.begin(), P.tpre.end()), P.tpre.end());
    std::sort(P.tprey.begin(), P.tprey.end());
    P.tprey.erase(std::unique(P.tprey.begin(), P.tprey.end()), P.tprey.end());
  }
  // ---- backward solve: the blocks of a column in ascending reference position of their rows ----
  P.bord.resize(P.nL);
  P.brow.resize(P.nL);
  for (int32_t j = 0; j < nb; ++j) {
    const int32_t s0 = P.colptr[j], s1 = P.colptr[j + 1];
    for (int32_t t = s0; t < s1; ++t) P.bord[t] = t;
    std::sort(P.bord.begin() + s0 + 1, P.bord.begin() + s1,
              [&](int32_t x, int32_t y) { return key[P.lrow[x]] < key[P.lrow[y]]; });
    for (int32_t t = s0; t < s1; ++t) P.brow[t] = P.lrow[P.bord[t]];
  }
  // ---- work split: rounds of cells per level (see direct.hpp) ----
  {
    const int NW = DirectPlan::CELL_WAVES, CS = DirectPlan::CELL_SLOTS, ST = DirectPlan::CELL_STRIDE;
    P.rptr.assign(1, 0);
    for (int32_t g = 0; g < P.ngroups(); ++g) {
      const int nw = g == P.ngroups() - 1 ? NW : P.sub_waves;
      for (int32_t l = P.gptr[g]; l < P.gptr[g + 1]; ++l) {
        const int32_t S0 = P.colptr[P.lcolp[l]], S1 = P.colptr[P.lcolp[l + 1]];
        // weight of a block = its products + 2, a diagonal block + DIAGW (its 7x7 Cholesky and
        // inverse run in a single lane: about as long as a dozen products)
        const int64_t DIAGW = 12;
        auto weight = [&](int32_t s) {
          return (int64_t)(P.pairptr[s + 1] - P.pairptr[s]) + 2 + (P.lrow[s] == P.lcol[s] ? DIAGW : 0);
        };
        int64_t W = 0;
        for (int32_t s = S0; s < S1; ++s) W += weight(s);
        const int64_t R0 = std::max<int64_t>(1, ((int64_t)(S1 - S0) + (int64_t)CS * nw - 1) / ((int64_t)CS * nw));
        const int64_t target = (W + R0 * nw - 1) / (R0 * nw);
        std::vector<int32_t> bounds(1, S0);  // cell boundaries, in order (round-major, wave-minor)
        int64_t cw = 0;
        int cnt = 0;
        for (int32_t s = S0; s < S1; ++s) {
          cw += weight(s);
          ++cnt;
          if (cw >= target || cnt == CS) { bounds.push_back(s + 1); cw = 0; cnt = 0; }
        }
        if (bounds.back() != S1) bounds.push_back(S1);
        const size_t ncell = bounds.size() - 1;
        const size_t R = std::max<size_t>(1, (ncell + nw - 1) / nw);
        for (size_t q = 0; q < R; ++q) {
          const size_t base = P.cells.size();
          P.cells.resize(base + ST, S1);
          for (int w = 0; w <= NW; ++w) {
            const size_t c = q * nw + (size_t)std::min(w, nw);
            const int32_t sb = c < bounds.size() ? bounds[c] : S1;
            P.cells[base + w] = sb;
            P.cells[base + NW + 1 + w] = P.pairptr[sb];
          }
        }
        P.rptr.push_back((int32_t)(P.cells.size() / ST));
      }
    }
  }
  return true;
}

}  // namespace sim3opt
