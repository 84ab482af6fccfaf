// host_sanitizer_driver.cpp -- the library's host-side algorithms (csrc/graph.cpp, amg.cpp, direct.cpp: pattern, row
// orders, partitions, halo plans, the partition-aware aggregation hierarchy, the elimination plan) under
// AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizers.py; GPU sanitizers do not exist on this pool, the
// index arithmetic that feeds the kernels is host code).  Invariants are checked along the way; prints "ok".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include "amg.hpp"
#include "direct.hpp"
#include "graph.hpp"

using namespace sim3opt;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #c); std::exit(1); } } while (0)

static unsigned long long lcg = 0x2545F4914F6CDD1Dull;
static unsigned rnd(unsigned n) {
  lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
  return (unsigned)((lcg >> 33) % n);
}

// random walk on a lattice: odometry edges + loops to earlier vertices nearby, some parallel edges, arbitrary ids
static HostGraph lattice_graph(int V, int E, int side, int fixed_extra) {
  HostGraph g;
  std::vector<int> cell(V);
  int x = 0, y = 0, z = 0;
  for (int i = 0; i < V; ++i) {
    g.vid.push_back(1000 + 3 * i);
    g.id2idx[1000 + 3 * i] = i;
    sim3::Sim3 S{};
    S.q[3] = 1.0; S.t[0] = x; S.t[1] = y; S.t[2] = z; S.s = 1.0;
    g.states.push_back(S);
    g.fixed.push_back(i == 0 || (fixed_extra > 0 && i == fixed_extra));
    cell[i] = (x * side + y) * 4 + z;
    switch (rnd(6)) {
      case 0: x = std::min(side - 1, x + 1); break;
      case 1: x = std::max(0, x - 1); break;
      case 2: y = std::min(side - 1, y + 1); break;
      case 3: y = std::max(0, y - 1); break;
      case 4: z = std::min(3, z + 1); break;
      default: z = std::max(0, z - 1); break;
    }
  }
  sim3::Sim3 I{};
  I.q[3] = 1.0; I.s = 1.0;
  for (int i = 1; i < V; ++i) { g.ev0.push_back(i); g.ev1.push_back(i - 1); g.meas.push_back(I); }
  while ((int)g.ev0.size() < E) {
    const int a = (int)rnd(V), b = (int)rnd(V);
    if (a == b) continue;
    const int ca = cell[a] / 4, cb = cell[b] / 4;
    if (std::abs(ca / side - cb / side) + std::abs(ca % side - cb % side) > 2 && rnd(50)) continue;  // mostly local loops
    g.ev0.push_back(std::min(a, b)); g.ev1.push_back(std::max(a, b)); g.meas.push_back(I);
  }
  return g;
}

static void check_partition_and_halo(const Structure& s) {
  for (int world : {1, 2, 3, 5, 8, 64}) {
    std::vector<int32_t> rb(world + 1), rb2(world + 1);
    partition_rows_equal(s.nb, world, rb.data());
    partition_rows(s.nb, s.rowptr.data(), world, rb2.data());
    CHECK(rb[0] == 0 && rb[world] == s.nb && rb2[0] == 0 && rb2[world] == s.nb);
    for (int r = 0; r < world; ++r) CHECK(rb[r] <= rb[r + 1] && rb2[r] <= rb2[r + 1]);
    std::vector<int32_t> brow, bseg;
    boundary_rows(s.nb, s.rowptr.data(), s.colidx.data(), world, rb.data(), brow, bseg);
    CHECK((int)bseg.size() == world + 1 && bseg[world] == (int)brow.size());
    std::vector<std::vector<int32_t>> sr(world), ss(world), rr(world), rs(world);
    for (int r = 0; r < world; ++r) {
      halo_plan(s.nb, s.rowptr.data(), s.colidx.data(), world, rb.data(), r, sr[r], ss[r], rr[r], rs[r]);
      CHECK((int)ss[r].size() == world + 1 && (int)rs[r].size() == world + 1);
      for (int32_t row : sr[r]) CHECK(row >= rb[r] && row < rb[r + 1]);
      for (int32_t row : rr[r]) CHECK(row >= 0 && row < s.nb && !(row >= rb[r] && row < rb[r + 1]));
    }
    for (int p = 0; p < world; ++p)     // what p sends to q is what q receives from p, in the same order
      for (int q = 0; q < world; ++q) {
        const std::vector<int32_t> a(sr[p].begin() + ss[p][q], sr[p].begin() + ss[p][q + 1]);
        const std::vector<int32_t> b(rr[q].begin() + rs[q][p], rr[q].begin() + rs[q][p + 1]);
        CHECK(a == b);
      }
  }
}

static int n_hierarchies = 0, n_constrained = 0;

static void check_hierarchy(const Structure& s, int world, int shard, int coarsest) {
  std::vector<int32_t> rb(world + 1);
  partition_rows_equal(s.nb, world, rb.data());
  AmgBuildOptions bo;
  bo.world = world;
  bo.row_begin = world > 1 ? rb.data() : nullptr;
  bo.shard_rows = shard;
  bo.max_coarsest = coarsest;
  std::vector<AmgLevelHost> L;
  std::string why;
  if (!build_amg_hierarchy(s.nb, s.rowptr.data(), s.colidx.data(), L, why, bo)) return;  // (a graph that does not coarsen)
  CHECK(L.size() >= 2 && L[0].nb == s.nb);
  ++n_hierarchies;
  for (size_t l = 0; l + 1 < L.size(); ++l) {
    const AmgLevelHost& F = L[l];
    const AmgLevelHost& C = L[l + 1];
    CHECK((int)F.agg.size() == F.nb && (int)F.mptr.size() == C.nb + 1 && (int)F.mem.size() == F.nb);
    CHECK((int64_t)F.gptr.size() == C.nnzb + 1 && F.gblk.size() == F.grow.size() && (int64_t)F.gblk.size() == F.nnzb);
    for (int i = 0; i < F.nb; ++i) CHECK(F.agg[i] >= 0 && F.agg[i] < C.nb);
    CHECK((int)C.rowptr.size() == C.nb + 1 && (int64_t)C.colidx.size() == C.nnzb && C.rowptr[C.nb] == C.nnzb);
    for (int a = 0; a < C.nb; ++a) CHECK(C.colidx[C.rowptr[a]] == a);  // diagonal first
    if (world > 1 && F.respects_owner && !F.row_begin.empty()) {       // aggregates inside the ranks' spans
      CHECK((int)C.row_begin.size() == world + 1 && C.row_begin[world] == C.nb);
      ++n_constrained;
      for (int r = 0; r < world; ++r)
        for (int i = F.row_begin[r]; i < F.row_begin[r + 1]; ++i)
          CHECK(F.agg[i] >= C.row_begin[r] && F.agg[i] < C.row_begin[r + 1]);
    }
  }
}

static void check_direct(int nb) {
  // chain with a few loops: full-symmetric pattern, diagonal first, columns ascending after it
  std::vector<std::set<int>> adj(nb);
  for (int i = 1; i < nb; ++i) { adj[i].insert(i - 1); adj[i - 1].insert(i); }
  for (int k = 0; k < nb / 10; ++k) {
    const int a = (int)rnd(nb), b = (int)rnd(nb);
    if (a != b) { adj[a].insert(b); adj[b].insert(a); }
  }
  std::vector<int32_t> rowptr(1, 0), colidx;
  for (int i = 0; i < nb; ++i) {
    colidx.push_back(i);
    for (int j : adj[i]) colidx.push_back(j);
    rowptr.push_back((int32_t)colidx.size());
  }
  for (int sub : {0, 8, 64})
    for (int waves : {1, 8}) {
      DirectPlan P;
      std::string why;
      CHECK(build_direct_plan(nb, rowptr.data(), colidx.data(), 30000000, sub, P, why, waves));
      CHECK(P.nb == nb && (int)P.perm.size() == nb && (int)P.colptr.size() == nb + 1 && P.colptr[nb] == P.nL);
      std::vector<int32_t> sorted = P.perm;
      std::sort(sorted.begin(), sorted.end());
      for (int i = 0; i < nb; ++i) CHECK(sorted[i] == i);
      CHECK((int64_t)P.pairptr.size() == P.nL + 1 && P.pairptr[P.nL] == P.npairs && (int64_t)P.pa.size() == P.npairs);
      for (int64_t k = 0; k < P.npairs; ++k) CHECK(P.pa[k] >= 0 && P.pa[k] < P.nL && P.pb[k] >= 0 && P.pb[k] < P.nL);
      for (int32_t k : P.src) CHECK(k >= 0 && k < (int32_t)colidx.size());
    }
  DirectPlan P;
  std::string why;
  CHECK(!build_direct_plan(nb, rowptr.data(), colidx.data(), 10, 0, P, why, 8) && !why.empty());  // refused: too many products
}

int main() {
  for (int variant = 0; variant < 3; ++variant) {
    const int V = variant == 0 ? 2500 : (variant == 1 ? 300 : 40);
    HostGraph g = lattice_graph(V, 10 * V, variant == 0 ? 16 : 6, variant == 1 ? 17 : 0);
    std::string err;
    for (int order = 0; order < 2; ++order) {
      Structure s;
      std::vector<int32_t> ord;
      if (order) locality_order(g, ord);
      CHECK(build_structure(g, s, err, order ? &ord : nullptr));
      CHECK(s.nb == V - 1 - (variant == 1 ? 1 : 0) && (int)s.rowptr.size() == s.nb + 1 && s.rowptr[s.nb] == s.nnzb);
      for (int i = 0; i < s.nb; ++i) CHECK(s.colidx[s.rowptr[i]] == i);
      for (size_t e = 0; e < g.ev0.size(); ++e) {
        if (s.slot01[e] >= 0) CHECK(s.slot01[e] < s.nnzb && s.colidx[s.slot01[e]] == s.hidx[g.ev1[e]]);
        if (s.inc0[e] >= 0) CHECK(s.inc0[e] < s.incptr[s.nb]);
      }
      check_partition_and_halo(s);
      for (int world : {1, 2, 4, 8})
        for (int shard : {1, 100, 1 << 30})
          check_hierarchy(s, world, shard, variant == 0 ? 64 : 16);
    }
  }
  {  // malformed graphs: refused with a message, nothing read out of bounds
    HostGraph g;
    Structure s;
    std::string err;
    CHECK(!build_structure(g, s, err) && !err.empty());
    HostGraph h = lattice_graph(5, 6, 3, 0);
    std::fill(h.fixed.begin(), h.fixed.end(), 1);
    err.clear();
    CHECK(!build_structure(h, s, err) && !err.empty());
  }
  check_direct(300);
  check_direct(17);
  CHECK(n_hierarchies >= 40 && n_constrained >= 30);
  std::printf("%d hierarchies, %d partition-constrained levels\nok\n", n_hierarchies, n_constrained);
  return 0;
}
