// amg.cpp -- pairwise-matching aggregation and Galerkin bookkeeping (see amg.hpp).
#include "amg.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace sim3opt {

namespace {

struct WGraph {  // weighted adjacency without self loops, neighbours ascending
  std::vector<int32_t> ptr, nbr;
  std::vector<int64_t> wgt;
  int32_t n() const { return (int32_t)ptr.size() - 1; }
};

// weight = number of level-0 blocks between two rows (parallel edges count); bw == nullptr: 1 each
WGraph graph_from_pattern(int32_t nb, const int32_t* rowptr, const int32_t* colidx,
                          const int64_t* bw) {
  WGraph g;
  g.ptr.assign(nb + 1, 0);
  for (int32_t i = 0; i < nb; ++i) {
    int32_t last = -1;
    for (int32_t k = rowptr[i] + 1; k < rowptr[i + 1]; ++k) {  // columns ascending after the diagonal
      const int32_t j = colidx[k];
      if (j == i) continue;
      const int64_t w = bw ? bw[k] : 1;
      if (j == last) {
        g.wgt.back() += w;
      } else {
        g.nbr.push_back(j);
        g.wgt.push_back(w);
        last = j;
      }
    }
    g.ptr[i + 1] = (int32_t)g.nbr.size();
  }
  return g;
}

// One pass of greedy heavy-edge matching in natural order; returns the number of clusters.
// owner (may be null): rank that owns every node; nodes of different ranks are never clustered together
int32_t match_pass(const WGraph& g, std::vector<int32_t>& cid, const int32_t* owner) {
  const int32_t m = g.n();
  std::vector<int32_t> match(m, -1);
  for (int32_t i = 0; i < m; ++i) {
    if (match[i] >= 0) continue;
    int32_t best = -1;
    int64_t bw = 0;
    for (int32_t k = g.ptr[i]; k < g.ptr[i + 1]; ++k) {
      const int32_t j = g.nbr[k];
      if (owner && owner[j] != owner[i]) continue;
      if (match[j] < 0 && g.wgt[k] > bw) {
        best = j;
        bw = g.wgt[k];
      }
    }
    if (best >= 0) {
      match[i] = best;
      match[best] = i;
    } else {
      match[i] = i;
    }
  }
  // rows left alone (all neighbours were taken when their turn came: leaves of hubs, odd ends of
  // chains) join the neighbouring cluster they are tied to most strongly, up to 4 rows per
  // cluster and pass -- otherwise star-like coarse graphs stop coarsening
  std::vector<int32_t> rep(m), csize(m, 0);
  for (int32_t i = 0; i < m; ++i) {
    rep[i] = std::min(i, match[i]);
    ++csize[rep[i]];
  }
  for (int32_t i = 0; i < m; ++i) {
    if (match[i] != i) continue;
    int32_t best = -1;
    int64_t bw = 0;
    for (int32_t k = g.ptr[i]; k < g.ptr[i + 1]; ++k) {
      if (owner && owner[g.nbr[k]] != owner[i]) continue;
      const int32_t c = rep[g.nbr[k]];
      if (c != i && csize[c] < 4 && g.wgt[k] > bw) {
        best = c;
        bw = g.wgt[k];
      }
    }
    if (best >= 0 && csize[i] == 1) {  // still alone (nobody joined it meanwhile)
      rep[i] = best;
      ++csize[best];
      csize[i] = 0;
    }
  }
  cid.assign(m, -1);
  int32_t nc = 0;
  for (int32_t i = 0; i < m; ++i)
    if (rep[i] == i) cid[i] = nc++;  // clusters numbered by their smallest row: locality survives
  for (int32_t i = 0; i < m; ++i) cid[i] = cid[rep[i]];
  return nc;
}

WGraph coarsen_graph(const WGraph& g, const std::vector<int32_t>& cid, int32_t nc) {
  struct E { uint64_t key; int64_t w; };
  std::vector<E> es;
  es.reserve(g.nbr.size());
  for (int32_t i = 0; i < g.n(); ++i)
    for (int32_t k = g.ptr[i]; k < g.ptr[i + 1]; ++k) {
      const int32_t a = cid[i], b = cid[g.nbr[k]];
      if (a != b) es.push_back({((uint64_t)(uint32_t)a << 32) | (uint32_t)b, g.wgt[k]});
    }
  std::sort(es.begin(), es.end(), [](const E& x, const E& y) { return x.key < y.key; });
  WGraph c;
  c.ptr.assign(nc + 1, 0);
  uint64_t last = ~0ull;
  for (const E& e : es) {
    if (e.key == last) {
      c.wgt.back() += e.w;
    } else {
      c.nbr.push_back((int32_t)(e.key & 0xffffffffu));
      c.wgt.push_back(e.w);
      ++c.ptr[(e.key >> 32) + 1];
      last = e.key;
    }
  }
  for (int32_t i = 0; i < nc; ++i) c.ptr[i + 1] += c.ptr[i];
  return c;
}

}  // namespace

bool build_amg_hierarchy(int32_t nb0, const int32_t* rowptr0, const int32_t* colidx0,
                         std::vector<AmgLevelHost>& levels, std::string& why, const AmgBuildOptions& bo) {
  levels.clear();
  levels.emplace_back();
  levels[0].nb = nb0;
  levels[0].nnzb = rowptr0[nb0];
  const bool parted = bo.world > 1 && bo.row_begin != nullptr;
  if (parted) levels[0].row_begin.assign(bo.row_begin, bo.row_begin + bo.world + 1);
  const int32_t* rowptr = rowptr0;
  const int32_t* colidx = colidx0;
  std::vector<int64_t> bw, bw_next;  // level-0 blocks behind each block of the current level
  const int max_coarsest = std::max(8, std::min(AMG_MAX_COARSEST, bo.max_coarsest > 0 ? bo.max_coarsest : AMG_MAX_COARSEST));
  for (;;) {
    AmgLevelHost& L = levels.back();
    const int32_t nb = L.nb;
    if (nb <= max_coarsest) return levels.size() > 1 || (why = "system already tiny", false);
    if ((int)levels.size() >= AMG_MAX_LEVELS) {
      why = "too many levels";
      return false;
    }
    // 3 matching passes: aggregates of up to 8 rows
    WGraph g = graph_from_pattern(nb, rowptr, colidx, bw.empty() ? nullptr : bw.data());
    std::vector<int32_t> agg(nb);
    std::iota(agg.begin(), agg.end(), 0);
    int32_t nc = nb;
    // aggregates of 8 (3 passes); graphs small enough for two levels with aggregates of 4 take those:
    // on KITTI-00 (770 rows -> 192 dense) a third of the PCG iterations (measured)
    int npass = levels.size() == 1 && nb <= 4 * max_coarsest ? 2 : 3;
    {
      const int32_t p = bo.passes[std::min<size_t>(levels.size() - 1, 2)];  // tuning knob: passes per level
      if (p >= 1 && p <= 6) npass = p;
    }
    // Owner of every node of the (repeatedly coarsened) matching graph: clusters stay inside one rank's span --
    // on the levels a multi-rank run PARTITIONS (level 0 and coarse levels above bo.shard_rows rows whose own
    // rows have owners): their Galerkin rows and restrictions then need no reduction across ranks.  A level that
    // will be replicated anyway aggregates freely; and where the constraint stalls the matching (a coarse level
    // with a few rows per rank that are not connected among themselves) the level aggregates freely too and is
    // marked: it must be replicated (respects_owner = false).
    bool constrain = parted && !L.row_begin.empty() && (levels.size() == 1 || nb > bo.shard_rows);
    std::vector<int32_t> owner;
    for (int attempt = 0; attempt < 2; ++attempt) {
      std::iota(agg.begin(), agg.end(), 0);
      nc = nb;
      WGraph gm = g;
      if (constrain) {
        owner.resize(nb);
        for (int32_t r = 0; r < bo.world; ++r)
          for (int32_t i = L.row_begin[r]; i < L.row_begin[r + 1]; ++i) owner[i] = r;
      }
      for (int pass = 0; pass < npass && nc > max_coarsest / 2; ++pass) {
        std::vector<int32_t> cid;
        const int32_t m = match_pass(gm, cid, constrain ? owner.data() : nullptr);
        for (int32_t& a : agg) a = cid[a];
        if (constrain) {
          std::vector<int32_t> oc(m);
          for (int32_t i = 0; i < (int32_t)cid.size(); ++i) oc[cid[i]] = owner[i];
          owner.swap(oc);
        }
        gm = coarsen_graph(gm, cid, m);
        nc = m;
      }
      if (nc <= nb - nb / 4 || !constrain || levels.size() == 1) break;
      constrain = false;  // (stalled under the constraint: once more without it; level 0 is never released)
    }
    L.respects_owner = constrain;
    if (nc > nb - nb / 4) {  // matching stalls (isolated rows, stars): no useful hierarchy
      why = "graph does not coarsen";
      return false;
    }
    L.agg = agg;
    L.mptr.assign(nc + 1, 0);
    for (int32_t i = 0; i < nb; ++i) ++L.mptr[agg[i] + 1];
    for (int32_t a = 0; a < nc; ++a) L.mptr[a + 1] += L.mptr[a];
    L.mem.resize(nb);
    {
      std::vector<int32_t> fill(L.mptr.begin(), L.mptr.end() - 1);
      for (int32_t i = 0; i < nb; ++i) L.mem[fill[agg[i]]++] = i;
    }
    // coarse pattern and the list of fine blocks behind every coarse block
    struct Ent { int32_t I, J, k, i; };
    std::vector<Ent> ents((size_t)L.nnzb);
    for (int32_t i = 0; i < nb; ++i)
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int32_t I = agg[i], J = agg[colidx[k]];
        ents[(size_t)k] = {I, J == I ? -1 : J, k, i};  // -1: the diagonal sorts first
      }
    std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) {
      if (x.I != y.I) return x.I < y.I;
      if (x.J != y.J) return x.J < y.J;
      return x.k < y.k;
    });
    AmgLevelHost C;
    C.nb = nc;
    if (constrain) {  // clusters are numbered by their smallest row and never straddle: contiguous spans again
      C.row_begin.assign(bo.world + 1, 0);
      for (int32_t a = 0; a < nc; ++a) ++C.row_begin[owner[a] + 1];
      for (int32_t r = 0; r < bo.world; ++r) C.row_begin[r + 1] += C.row_begin[r];
      for (int32_t a = 1; a < nc; ++a)
        if (owner[a] < owner[a - 1]) {
          why = "internal: coarse ownership is not contiguous";
          return false;
        }
    }
    C.rowptr.assign(nc + 1, 0);
    L.gblk.resize(ents.size());
    L.grow.resize(ents.size());
    int32_t lastI = -1, lastJ = -2;
    bw_next.clear();
    for (size_t e = 0; e < ents.size(); ++e) {
      const Ent& t = ents[e];
      if (t.I != lastI || t.J != lastJ) {
        C.colidx.push_back(t.J < 0 ? t.I : t.J);
        bw_next.push_back(0);
        L.gptr.push_back((int32_t)e);
        ++C.rowptr[t.I + 1];
        lastI = t.I;
        lastJ = t.J;
      }
      L.gblk[e] = t.k;
      L.grow[e] = t.i;
      bw_next.back() += bw.empty() ? 1 : bw[(size_t)t.k];
    }
    L.gptr.push_back((int32_t)ents.size());
    for (int32_t a = 0; a < nc; ++a) C.rowptr[a + 1] += C.rowptr[a];
    C.nnzb = (int64_t)C.colidx.size();
    levels.push_back(std::move(C));
    bw.swap(bw_next);
    rowptr = levels.back().rowptr.data();
    colidx = levels.back().colidx.data();
  }
}

}  // namespace sim3opt
