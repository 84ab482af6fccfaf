// graph.cpp -- block-CSR structure builder (see graph.hpp).
#include "graph.hpp"

#include <algorithm>
#include <numeric>

namespace sim3opt {

bool build_structure(const HostGraph& g, Structure& s, std::string& err) {
  const int32_t nv = g.nv(), ne = g.ne();
  s = Structure();
  s.hidx.assign(nv, -1);
  for (int32_t v = 0; v < nv; ++v)
    if (!g.fixed[v]) {
      s.hidx[v] = s.nb++;
      s.row2vertex.push_back(v);
    }
  if (s.nb == 0 || ne == 0) {
    err = "nothing to optimise: no free vertex or no edge";
    return false;
  }
  const int32_t nb = s.nb;
  // count blocks per row (1 diagonal + one per incident edge with both ends free) and incidences
  std::vector<int32_t> nblk(nb, 1), ninc(nb, 0);
  s.active.reserve(ne);
  for (int32_t k = 0; k < ne; ++k) {
    const int32_t a = s.hidx[g.ev0[k]], b = s.hidx[g.ev1[k]];
    if (a < 0 && b < 0) continue;
    if (g.ev0[k] == g.ev1[k]) {
      err = "edge with identical endpoints";
      return false;
    }
    s.active.push_back(k);
    if (a >= 0) ++ninc[a];
    if (b >= 0) ++ninc[b];
    if (a >= 0 && b >= 0) {
      ++nblk[a];
      ++nblk[b];
    }
  }
  s.rowptr.assign(nb + 1, 0);
  s.incptr.assign(nb + 1, 0);
  int64_t tot = 0;
  for (int32_t i = 0; i < nb; ++i) {
    tot += nblk[i];
    if (tot > INT32_MAX) {
      err = "block count exceeds int32";
      return false;
    }
    s.rowptr[i + 1] = (int32_t)tot;
    s.incptr[i + 1] = s.incptr[i] + ninc[i];
  }
  s.nnzb = tot;
  // off-diagonal entries: (row, col, edge, direction) sorted by (row, col, edge)
  struct Ent { int32_t row, col, edge, dir; };
  std::vector<Ent> ents;
  ents.reserve((size_t)(tot - nb));
  for (int32_t k : s.active) {
    const int32_t a = s.hidx[g.ev0[k]], b = s.hidx[g.ev1[k]];
    if (a >= 0 && b >= 0) {
      ents.push_back({a, b, k, 0});
      ents.push_back({b, a, k, 1});
    }
  }
  std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) {
    if (x.row != y.row) return x.row < y.row;
    if (x.col != y.col) return x.col < y.col;
    return x.edge < y.edge;
  });
  s.colidx.assign((size_t)tot, 0);
  s.slot01.assign(ne, -1);
  s.slot10.assign(ne, -1);
  {
    std::vector<int32_t> fill(nb, 1);  // slot 0 of each row is the diagonal
    for (int32_t i = 0; i < nb; ++i) s.colidx[s.rowptr[i]] = i;
    for (const Ent& e : ents) {
      const int32_t pos = s.rowptr[e.row] + fill[e.row]++;
      s.colidx[pos] = e.col;
      (e.dir == 0 ? s.slot01 : s.slot10)[e.edge] = pos;
    }
  }
  // incidence slots in edge order
  s.inc0.assign(ne, -1);
  s.inc1.assign(ne, -1);
  {
    std::vector<int32_t> fill(nb, 0);
    for (int32_t k : s.active) {
      const int32_t a = s.hidx[g.ev0[k]], b = s.hidx[g.ev1[k]];
      if (a >= 0) s.inc0[k] = s.incptr[a] + fill[a]++;
      if (b >= 0) s.inc1[k] = s.incptr[b] + fill[b]++;
    }
  }
  return true;
}

void partition_rows(int32_t nb, const int32_t* rowptr, int32_t world, int32_t* begin) {
  // split so that every rank streams about the same number of 7x7 blocks per SpMV
  const int64_t base = rowptr[0], total = (int64_t)rowptr[nb] - base;  // rowptr may be a sub-range
  begin[0] = 0;
  int32_t row = 0;
  for (int32_t r = 1; r < world; ++r) {
    const int64_t target = total * r / world;
    while (row < nb && (int64_t)rowptr[row] - base < target) ++row;
    begin[r] = row;
  }
  begin[world] = nb;
  for (int32_t r = 1; r <= world; ++r)
    if (begin[r] < begin[r - 1]) begin[r] = begin[r - 1];
}

void partition_rows_equal(int32_t nb, int32_t world, int32_t* begin) {
  // rank spans of equal length (the last ones may be short or empty): with 7*rows_per_rank
  // doubles per rank the search-direction exchange is ONE in-place ncclAllGather
  const int32_t rpr = (nb + world - 1) / world;
  for (int32_t r = 0; r <= world; ++r) begin[r] = (int32_t)std::min<int64_t>((int64_t)r * rpr, nb);
}

}  // namespace sim3opt
