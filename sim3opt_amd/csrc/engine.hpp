// engine.hpp -- device-resident Levenberg-Marquardt engine (interface).
//
// Replaces g2o's OptimizationAlgorithmLevenberg + BlockSolverX + LinearSolverEigen
// (instantiated at kitti_surf.cpp:552-558) and the per-edge / per-vertex virtual calls
// behind SparseOptimizer::optimize (kitti_surf.cpp:675).  Implementation: engine.hip.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sim3opt.h"
#include "graph.hpp"

namespace sim3opt {

class Engine;  // opaque, defined in engine.hip

struct Comm;
// comm (may be null) is moved into the engine
Engine* engine_create(const HostGraph& g, const Structure& s, const sim3opt_options& opt,
                      Comm* comm, std::string& err, int& status);
void engine_local_rows(const Engine* e, int32_t* begin, int32_t* end);
void engine_destroy(Engine* e);
// hands the communicator back before the engine is destroyed (re-initialisation)
void engine_take_comm(Engine* e, Comm* out);

int engine_set_options(Engine* e, const sim3opt_options& opt);
int engine_optimize(Engine* e, int32_t max_iters, std::vector<sim3opt_iter_stats>& stats,
                    std::string& err);
int engine_chi2(Engine* e, double* chi2, std::string& err);
int engine_get_states(Engine* e, sim3::Sim3* out, std::string& err);
int engine_set_states(Engine* e, const sim3::Sim3* in, std::string& err);
int engine_edge_errors(Engine* e, double* out, std::string& err);
int engine_linearize(Engine* e, std::string& err);
int engine_get_system(Engine* e, int32_t* rowptr, int32_t* colidx, double* values, double* b,
                      std::string& err);
int engine_solve(Engine* e, double lambda, double* x, int32_t* iters, double* rel_res,
                 std::string& err);
int engine_bench_spmv(Engine* e, int32_t reps, double* ms_mean, std::string& err);
int engine_bench_stream(Engine* e, int32_t mode, int32_t reps, double* ms_mean, std::string& err);
int engine_preconditioner(const Engine* e);
int engine_linear_solver(const Engine* e);
void engine_amg_in_use(const Engine* e, int32_t* n_levels, int32_t* n_partitioned, int32_t visits[4]);
void engine_device_bytes(const Engine* e, int64_t bytes[2]);
int engine_kernel_times(Engine* e, sim3opt_kernel_times* out, bool reset);
int engine_comm_times(Engine* e, sim3opt_comm_times* out);
#ifdef SIM3OPT_BENCH_HOOKS
int engine_bench_spmv_symmetric(Engine* e, int32_t reps, double out[4], std::string& err);
int engine_bench_spmv_rowlane(Engine* e, int32_t reps, int32_t rows_per_group, double out[8], std::string& err);
#endif

}  // namespace sim3opt
