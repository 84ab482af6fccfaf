// graph.cpp -- block-CSR structure builder (see graph.hpp).
#include "graph.hpp"

#include <algorithm>
#include <numeric>

namespace sim3opt {

bool build_structure(const HostGraph& g, Structure& s, std::string& err,
                     const std::vector<int32_t>* row_order) {
  const int32_t nv = g.nv(), ne = g.ne();
  s = Structure();
  s.hidx.assign(nv, -1);
  if (row_order) {
    for (int32_t v : *row_order) {
      if (v < 0 || v >= nv || g.fixed[v] || s.hidx[v] >= 0) {
        err = "row order: not a permutation of the free vertices";
        return false;
      }
      s.hidx[v] = s.nb++;
      s.row2vertex.push_back(v);
    }
    for (int32_t v = 0; v < nv; ++v)
      if (!g.fixed[v] && s.hidx[v] < 0) {
        err = "row order: a free vertex is missing";
        return false;
      }
  } else {
    for (int32_t v = 0; v < nv; ++v)
      if (!g.fixed[v]) {
        s.hidx[v] = s.nb++;
        s.row2vertex.push_back(v);
      }
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

void locality_order(const HostGraph& g, std::vector<int32_t>& order) {
  const int32_t nv = g.nv(), ne = g.ne();
  order.clear();
  // adjacency among free vertices (CSR, duplicates from parallel edges are harmless)
  std::vector<int32_t> ptr(nv + 1, 0);
  for (int32_t k = 0; k < ne; ++k) {
    const int32_t a = g.ev0[k], b = g.ev1[k];
    if (a == b || g.fixed[a] || g.fixed[b]) continue;
    ++ptr[a + 1];
    ++ptr[b + 1];
  }
  for (int32_t v = 0; v < nv; ++v) ptr[v + 1] += ptr[v];
  std::vector<int32_t> nbr(ptr[nv]), fill(ptr.begin(), ptr.end() - 1);
  for (int32_t k = 0; k < ne; ++k) {
    const int32_t a = g.ev0[k], b = g.ev1[k];
    if (a == b || g.fixed[a] || g.fixed[b]) continue;
    nbr[fill[a]++] = b;
    nbr[fill[b]++] = a;
  }
  auto degree = [&](int32_t v) { return ptr[v + 1] - ptr[v]; };
  std::vector<int32_t> level(nv, -1), queue;
  // breadth-first search from `root` over vertices with level < 0 or of the current sweep `mark`;
  // returns the last vertex reached (a farthest one)
  std::vector<int32_t> mark(nv, -1);
  auto bfs = [&](int32_t root, int32_t sweep, std::vector<int32_t>* out) {
    queue.clear();
    queue.push_back(root);
    mark[root] = sweep;
    std::vector<int32_t> nb_sorted;
    for (size_t head = 0; head < queue.size(); ++head) {
      const int32_t v = queue[head];
      nb_sorted.clear();
      for (int32_t e = ptr[v]; e < ptr[v + 1]; ++e) {
        const int32_t w = nbr[e];
        if (mark[w] != sweep && level[w] < 0) {
          mark[w] = sweep;
          nb_sorted.push_back(w);
        }
      }
      std::sort(nb_sorted.begin(), nb_sorted.end(), [&](int32_t x, int32_t y) {
        const int32_t dx = degree(x), dy = degree(y);
        return dx != dy ? dx < dy : x < y;
      });
      queue.insert(queue.end(), nb_sorted.begin(), nb_sorted.end());
    }
    if (out) *out = queue;
    return queue.back();
  };
  int32_t sweep = 0;
  for (int32_t v0 = 0; v0 < nv; ++v0) {
    if (g.fixed[v0] || level[v0] >= 0) continue;
    // pseudo-peripheral start of this component: twice to the far end
    int32_t root = bfs(v0, sweep++, nullptr);
    root = bfs(root, sweep++, nullptr);
    std::vector<int32_t> comp;
    bfs(root, sweep++, &comp);
    for (int32_t v : comp) {
      level[v] = 0;
      order.push_back(v);
    }
  }
}

void boundary_rows(int32_t nb, const int32_t* rowptr, const int32_t* colidx, int32_t world,
                   const int32_t* row_begin, std::vector<int32_t>& rows, std::vector<int32_t>& seg) {
  rows.clear();
  seg.assign(world + 1, 0);
  int32_t r = 0;
  for (int32_t i = 0; i < nb; ++i) {
    while (r + 1 < world && i >= row_begin[r + 1]) ++r;
    bool cut = false;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1] && !cut; ++k)
      cut = colidx[k] < row_begin[r] || colidx[k] >= row_begin[r + 1];
    if (cut) {
      rows.push_back(i);
      ++seg[r + 1];
    }
  }
  for (int32_t q = 0; q < world; ++q) seg[q + 1] += seg[q];
}

void halo_plan(int32_t nb, const int32_t* rowptr, const int32_t* colidx, int32_t world, const int32_t* row_begin,
               int32_t rank, std::vector<int32_t>& send_rows, std::vector<int32_t>& send_seg,
               std::vector<int32_t>& recv_rows, std::vector<int32_t>& recv_seg) {
  (void)nb;
  const int32_t lo = row_begin[rank], hi = row_begin[rank + 1];
  auto owner_of = [&](int32_t j) {
    return (int32_t)(std::upper_bound(row_begin, row_begin + world + 1, j) - row_begin) - 1;
  };
  std::vector<std::vector<int32_t>> snd(world), rcv(world);
  for (int32_t i = lo; i < hi; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t j = colidx[k];
      if (j >= lo && j < hi) continue;
      const int32_t p = owner_of(j);
      if (snd[p].empty() || snd[p].back() != i) snd[p].push_back(i);  // rows ascend: duplicates are adjacent
      rcv[p].push_back(j);
    }
  send_rows.clear();
  recv_rows.clear();
  send_seg.assign(world + 1, 0);
  recv_seg.assign(world + 1, 0);
  for (int32_t p = 0; p < world; ++p) {
    std::sort(rcv[p].begin(), rcv[p].end());
    rcv[p].erase(std::unique(rcv[p].begin(), rcv[p].end()), rcv[p].end());
    send_rows.insert(send_rows.end(), snd[p].begin(), snd[p].end());
    recv_rows.insert(recv_rows.end(), rcv[p].begin(), rcv[p].end());
    send_seg[p + 1] = (int32_t)send_rows.size();
    recv_seg[p + 1] = (int32_t)recv_rows.size();
  }
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
