// amg.hpp -- host-side set-up of the aggregation-multigrid preconditioner (structure only).
//
// The LM normal matrix of a pose graph, H = sum_e [J0 J1]^T W [J0 J1], is a connection Laplacian:
// its low-energy modes are the fields dx_v = Ad(S_v) g, g in R^7 (a global right-multiplication
// S_v -> S_v exp(g) leaves every residual log(C S_v0 S_v1^-1) unchanged; in the left perturbation
// VertexSim3Expmap::oplusImpl uses, that is dx_v = Ad(S_v) g).  Block-Jacobi cannot see them, so on
// locally connected graphs (Manhattan world: every vertex linked to ~20 neighbours a few cells
// away) PCG needs thousands of iterations.  The hierarchy here coarsens the GRAPH by pairwise
// matching (3 passes = aggregates of 8 block rows per level, 2 passes when that already reaches the
// dense level); the numbers (Galerkin products
// with the Ad-transported piecewise-constant prolongation) are formed on the GPU after every
// linearisation (engine.hip / amg_kernels.hpp).  The reference solves the system exactly
// (LinearSolverEigen, kitti_surf.cpp:553-554); this is how the PCG gets to the same answer in tens
// of iterations on config 3.  Pure host C++17, no GPU calls.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace sim3opt {

struct AmgLevelHost {
  int32_t nb = 0;   // block rows of this level
  int64_t nnzb = 0; // stored blocks of this level (level 0: the engine's own pattern)
  std::vector<int32_t> rowptr, colidx;  // pattern (diagonal first, then unique sorted columns); empty on level 0
  // transfer to the next (coarser) level; empty on the coarsest level
  std::vector<int32_t> agg;        // nb: aggregate (= coarse block row) of each row
  std::vector<int32_t> mptr, mem;  // coarse nb + 1, nb: rows of each aggregate, ascending
  std::vector<int32_t> gptr;       // coarse nnzb + 1: contributions to each coarse block ...
  std::vector<int32_t> gblk, grow; // ... as (block index, block row) of THIS level, ascending block index
  // row partition of this level over the ranks (world + 1 entries; empty: one rank).  Aggregates never
  // straddle two ranks and are numbered by their smallest row, so every level's spans are contiguous and
  // a rank's coarse rows are exactly the aggregates of its own fine rows: Galerkin products and
  // restrictions need no reduction across ranks
  std::vector<int32_t> row_begin;
  // this level's own aggregation stayed inside the ranks' spans (only then may the level be partitioned)
  bool respects_owner = false;
};

struct AmgBuildOptions {
  int32_t max_coarsest = 256;       // most rows of the dense coarsest level (8 .. AMG_MAX_COARSEST)
  int32_t passes[3] = {0, 0, 0};    // matching passes on level 0, 1, >= 2; 0 = automatic
  int32_t shard_rows = 0;           // coarse levels with more rows than this will be partitioned (and are constrained)
  int32_t world = 1;                // ranks of the row partition the aggregation must respect
  const int32_t* row_begin = nullptr;  // world + 1: level-0 spans (required when world > 1)
};

constexpr int AMG_MAX_COARSEST = 256;  // block rows of the dense coarsest level (1792 unknowns)
constexpr int AMG_MAX_LEVELS = 10;

// Builds the level patterns from the level-0 block-CSR pattern.  Returns false (with a reason)
// when the graph does not coarsen to AMG_MAX_COARSEST rows, e.g. star-like graphs.
bool build_amg_hierarchy(int32_t nb, const int32_t* rowptr, const int32_t* colidx,
                         std::vector<AmgLevelHost>& levels, std::string& why,
                         const AmgBuildOptions& bo = AmgBuildOptions());

}  // namespace sim3opt
