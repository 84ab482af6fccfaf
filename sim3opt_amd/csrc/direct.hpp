// direct.hpp -- host-side plan of the exact sparse block Cholesky (structure only).
//
// The reference solves the damped normal equations exactly: LinearSolverEigen = block -> scalar CSC,
// AMD ordering, Eigen::SimplicialLDLT, two triangular solves (kitti_surf.cpp:553-554, run at
// :674-675).  On KITTI-00 (770 free keyframes, cond(H + lambda I) ~ 1e12 in
// the reference's arithmetic) a preconditioned CG cannot stand in for that: its steps are not
// accurate enough, LM rejects trials the reference accepts and leaves its trajectory.  This is the
// device counterpart: a level-scheduled left-looking Cholesky on the 7x7 block pattern.
//
// A minimum-degree order (what Eigen's AMD produces) eliminates a chain from its ends: an
// elimination tree as tall as the chain, i.e. 770 dependent steps -- fine for one CPU thread,
// hopeless for a GPU.  The plan here orders by nested dissection (separators = BFS level sets of a
// pseudo-peripheral vertex), which keeps the fill as small (KITTI-00: 2.3-2.8 k blocks) and makes the
// tree 10-20 levels tall; columns are then renumbered by (group, level) so that every level is a
// contiguous range of columns and of stored blocks.  Groups: subtrees at the bottom of the tree are
// independent, one workgroup each; the top of the tree is one last group.
//
// Everything the numeric kernels need is precomputed here once per sim3opt_initialize: where each
// block of L takes its entries of H from, and for every target block L(i,j) the fixed-order list of
// products L(i,k) L(j,k)^T it subtracts -- no atomics, bit-reproducible.  Pure host C++17.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace sim3opt {

struct DirectPlan {
  int32_t nb = 0;                 // block columns (free vertices)
  int64_t nL = 0;                 // stored blocks of L (lower triangle incl. diagonal)
  int64_t npairs = 0;             // 7x7x7 products of one factorisation
  int32_t height = 0;             // levels of the elimination tree (block columns)
  std::vector<int32_t> perm;      // perm[j] = block row of H eliminated at position j
  std::vector<int32_t> colptr;    // nb + 1: blocks of column j, diagonal first, rows ascending
  std::vector<int32_t> lrow;      // nL: row (elimination position) of each block
  std::vector<int32_t> lcol;      // nL: column of each block
  std::vector<int32_t> srcptr;    // nL + 1 -> src: blocks of H (indices into the block-CSR values)
  std::vector<int32_t> src;       //   summed into this block of L (parallel edges: several; fill: none)
  std::vector<int32_t> pairptr;   // nL + 1 -> pa/pb: products L[pa] L[pb]^T subtracted from this
  std::vector<int32_t> pa, pb;    //   block, in ascending REFERENCE position of the source column (direct.cpp)
  std::vector<int32_t> pcol;      //   ... and that source column k (the forward solve rides along)
  std::vector<int32_t> tpre;      // blocks of the bottom groups that the top group's products read, ascending
  std::vector<int32_t> tprey;     //   ... and the bottom columns whose y they read (the top group's warm-up)
  std::vector<int32_t> bord;      // nL: backward solve of column j visits its blocks bord[colptr[j] + 1 ...]
  std::vector<int32_t> brow;      //   ... whose rows are brow[same index] (a fixed order of summation that
                                  //   does not depend on the schedule; direct.cpp)
  // schedule: group g runs levels [gptr[g], gptr[g+1]); level l is columns [lcolp[l], lcolp[l+1]).
  // The last group is the top of the tree (everything the others feed into).
  std::vector<int32_t> gptr;
  std::vector<int32_t> lcolp;
  // work split of the factorisation kernel: level l is done in rounds [rptr[l], rptr[l+1]); in round
  // q wavefront w owns the contiguous blocks cells[CELL_STRIDE q + w] .. cells[CELL_STRIDE q + w + 1]
  // (at most CELL_SLOTS of them, balanced by their product counts) and their products
  // cells[CELL_STRIDE q + CELL_WAVES + 1 + w] .. [.. + w + 1] -- one scalar read tells a wavefront
  // everything it needs to issue its index loads.  Bottom groups run `sub_waves` wavefronts, the top
  // group CELL_WAVES.
  static constexpr int CELL_WAVES = 8, CELL_SLOTS = 8, CELL_STRIDE = 2 * (CELL_WAVES + 1);
  static constexpr int STAGE_PRODUCTS = 112;  // products whose operands the kernel stages in LDS at a time (840 B each):
                                              // a round of up to that many is fetched by all wavefronts together
  int32_t sub_waves = 8;
  std::vector<int32_t> rptr;
  std::vector<int32_t> cells;
  int32_t ngroups() const { return (int32_t)gptr.size() - 1; }
};

// Builds the plan from the full-symmetric block-CSR pattern (diagonal block first in every row,
// parallel edges as repeated columns).  Returns false (with a reason) when one factorisation would
// need more than max_pairs block products -- the graph is then left to the PCG.
// subtree_cols: bottom subtrees of at most this many columns become independent groups
// (<= 0: automatic).
// sub_waves: wavefronts per workgroup of the bottom groups (1 .. CELL_WAVES).
bool build_direct_plan(int32_t nb, const int32_t* rowptr, const int32_t* colidx, int64_t max_pairs,
                       int32_t subtree_cols, DirectPlan& plan, std::string& why, int32_t sub_waves = 8);

}  // namespace sim3opt
