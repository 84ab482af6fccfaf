// graph.hpp -- host-side pose-graph container and block-CSR structure builder.
//
// Replaces the container half of g2o::SparseOptimizer / OptimizableGraph /
// HyperGraph (addVertex, addEdge, vertex(id), initializeOptimization's index
// mapping; reference call sites kitti_surf.cpp:558, 620, 634-638, 664-668, 674).
// Pure host C++17, no GPU calls: the structure it builds is what the HIP engine
// uploads to HBM.
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "sim3_math.hpp"

namespace sim3opt {

struct HostGraph {
  // vertices, insertion order
  std::vector<int32_t> vid;
  std::unordered_map<int32_t, int32_t> id2idx;
  std::vector<sim3::Sim3> states;
  std::vector<uint8_t> fixed;
  // edges, insertion order; endpoints as vertex indices
  std::vector<int32_t> ev0, ev1;
  std::vector<sim3::Sim3> meas;
  bool has_info = false;       // some edge carries a non-identity information matrix
  bool has_kernel = false;     // some edge carries a robust kernel
  std::vector<double> info;    // m x 49 once has_info, else empty
  std::vector<double> kdelta;  // m once has_kernel (0 = none, >0 = Huber delta), else empty

  int32_t nv() const { return (int32_t)vid.size(); }
  int32_t ne() const { return (int32_t)ev0.size(); }
};

// Block-CSR pattern of the LM normal equations, full-symmetric storage, 7x7 blocks.
//   block row k  = k-th free vertex in insertion order (g2o: hessianIndex; fixed -> -1)
//   row layout   = [diagonal block, then one block per incident edge whose other endpoint is
//                   free, sorted by (column, edge index)] -- parallel edges keep separate blocks,
//                   so the linearisation kernel owns every off-diagonal block exclusively
//                   (plain stores, no atomics) and SpMV sums them implicitly.
//   incidence    = for each free vertex the list of (edge, role) pairs in edge order; the
//                   linearisation kernel writes per-incidence diagonal/b contributions which a
//                   row kernel reduces in that fixed order (deterministic assembly).
struct Structure {
  int32_t nb = 0;                  // free block rows
  int64_t nnzb = 0;                // stored blocks
  std::vector<int32_t> hidx;       // vertex index -> block row or -1
  std::vector<int32_t> row2vertex; // block row -> vertex index
  std::vector<int32_t> rowptr;     // nb + 1
  std::vector<int32_t> colidx;     // nnzb
  std::vector<int32_t> slot01;     // per edge: block index of H(h0,h1), -1 if not both free
  std::vector<int32_t> slot10;     // per edge: block index of H(h1,h0)
  std::vector<int32_t> incptr;     // nb + 1
  std::vector<int32_t> inc0;       // per edge: incidence slot of endpoint 0, -1 if fixed
  std::vector<int32_t> inc1;       // per edge: incidence slot of endpoint 1, -1 if fixed
  std::vector<int32_t> active;     // edges with at least one free endpoint (linearised)
};

// Returns false and fills err on malformed graphs (no free vertex, no edge, ...).
// row_order (optional): the free vertices in the order their block rows shall take (default:
// insertion order, g2o's hessianIndex).  The row-partitioned multi-GPU path passes locality_order().
bool build_structure(const HostGraph& g, Structure& s, std::string& err,
                     const std::vector<int32_t>* row_order = nullptr);

// Free vertices in breadth-first order from a pseudo-peripheral vertex (Cuthill-McKee: neighbours by
// ascending degree; further components appended).  Contiguous spans of this order are slabs of the
// graph: on the 100k / 1M Manhattan graph 6 % / 16 % / 34 % of the rows have a neighbour on another
// rank at 2 / 4 / 8 ranks and 2 % / 5 % / 10 % of the edges are cut, against 94-99 % and 40-70 % in
// insertion order (the walk wanders through the whole lattice) -- so the per-iteration exchange of
// the partitioned PCG shrinks from the whole vector to its boundary rows.
void locality_order(const HostGraph& g, std::vector<int32_t>& order);

// Boundary rows of a contiguous row partition: rows with a stored block whose column another rank
// owns, ascending (hence grouped by owner); seg[r] .. seg[r + 1] = rank r's share (world + 1 entries).
void boundary_rows(int32_t nb, const int32_t* rowptr, const int32_t* colidx, int32_t world,
                   const int32_t* row_begin, std::vector<int32_t>& rows, std::vector<int32_t>& seg);

// Neighbour-only halo plan of rank `rank` on a level with a contiguous row partition and a structurally
// symmetric pattern: send_rows = my rows that another rank's rows reference, grouped by that rank (ascending
// row inside a group; send_seg has world + 1 entries), recv_rows = the other ranks' rows my rows reference,
// grouped by owner.  Built from the rank's own rows only; by the symmetry of the pattern rank p's receive
// group for q is exactly q's send group for p, in the same order.
void halo_plan(int32_t nb, const int32_t* rowptr, const int32_t* colidx, int32_t world, const int32_t* row_begin,
               int32_t rank, std::vector<int32_t>& send_rows, std::vector<int32_t>& send_seg,
               std::vector<int32_t>& recv_rows, std::vector<int32_t>& recv_seg);

// Contiguous row partition balanced by stored blocks (multi-GPU row split); begin has world+1 entries.
void partition_rows(int32_t nb, const int32_t* rowptr, int32_t world, int32_t* begin);

// Equal-length contiguous row spans (multi-GPU rank partition); begin has world+1 entries.
void partition_rows_equal(int32_t nb, int32_t world, int32_t* begin);

}  // namespace sim3opt
