// direct_args.hpp -- argument block and launch constants of the exact block Cholesky kernels
// (direct_kernels.hpp); included inside the namespace of the translation unit that uses them (the pose-graph
// engine, the bundle adjuster), after DevScalars and direct.hpp.
#pragma once
struct LdlArgs {
  const int32_t* perm;
  const int32_t* colptr;
  const int32_t* lrow;
  const int32_t* lcol;
  const int32_t* srcptr;
  const int32_t* src;
  const int32_t* pairptr;
  const int32_t* pa;
  const int32_t* pb;
  const int32_t* pcol;
  const int32_t* gptr;
  const int32_t* lcolp;
  const int32_t* bord;
  const int32_t* brow;
  const int32_t* tpre;   // the top group's warm-up lists (blocks of L, columns of y) and their lengths
  const int32_t* tprey;
  int32_t ntpre, ntprey;
  const int32_t* rptr;
  const int32_t* cells;
  const double* vals;  // block-CSR values of H (column-major 7x7)
  const double* b;     // right-hand side, block rows of H
  double* Aperm;       // nL x 49: blocks of H in the layout of L (no damping)
  double* bp;          // 7 nb: b in elimination order
  double* L;           // nL x 49
  double* Dinv;        // nb x 49: L(j,j)^-1 (lower triangular, column-major)
  double* y;           // 7 nb, elimination order
  double* xp;          // 7 nb, elimination order
  double* x;           // 7 nb, block rows of H (the result)
  int32_t nb, nL;
  int32_t fail_token = 1;  // what a non-positive pivot writes into sc->fail (the engine: a number per solve)
  double lambda;
  DevScalars* sc;
  long long* dbg;  // tuning aid (SIM3OPT_DIRECT_TRACE): wall_clock64 stamps of the top group's levels
};

constexpr int LDL_WG_TOP = 64 * DirectPlan::CELL_WAVES;  // the top of the tree: one workgroup of 8
// wavefronts (512 threads leave each wavefront 256 VGPRs: with 1024 the operand batches spilled)
constexpr int LDL_WG_SUB = 512;                          // bottom subtrees: 8 wavefronts each too (round 3 sweep)
constexpr int LDL_CS = DirectPlan::CELL_SLOTS;
constexpr int LDL_ST = DirectPlan::CELL_STRIDE;
constexpr int LDL_NW = DirectPlan::CELL_WAVES;
constexpr int LDL_STAGE = DirectPlan::STAGE_PRODUCTS;  // products whose operands one piece stages in LDS (92 KB)
constexpr int LDL_WCH = LDL_STAGE / LDL_NW;             // ... a wavefront's own slice of that, in wide rounds

