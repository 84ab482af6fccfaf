// capi.cpp -- the C-ABI of libsim3opt (include/sim3opt.h): argument checking, the host graph
// container and dispatch into the HIP engine.  No exceptions leave this file.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/sim3opt.h"
#include "../../include/sim3opt_bench.h"
#include "amg.hpp"
#include "comm.hpp"
#include "devmem.hpp"
#include "direct.hpp"
#include "engine.hpp"
#include "graph.hpp"

using namespace sim3opt;

struct sim3opt_graph {
  HostGraph host;
  Structure structure;
  sim3opt_options opt;
  Engine* engine = nullptr;
  bool initialized = false;
  bool dirty = false;  // vertices/edges added since the last initialize (g2o: re-initialize)
  std::vector<sim3opt_iter_stats> stats;
  std::string err;
  Comm comm;        // handed to the engine at initialize
  bool comm_set = false;
  ~sim3opt_graph() {
    if (engine) engine_destroy(engine);
  }
};

namespace {

int fail(sim3opt_graph* g, int code, const char* msg) {
  if (g) g->err = msg;
  return code;
}

bool state_ok(const double s[8]) {
  for (int i = 0; i < 8; ++i)
    if (!std::isfinite(s[i])) return false;
  return s[7] > 0.0;
}

sim3::Sim3 to_sim3(const double s[8]) {
  sim3::Sim3 r;
  r.q[0] = s[0]; r.q[1] = s[1]; r.q[2] = s[2]; r.q[3] = s[3];
  r.t[0] = s[4]; r.t[1] = s[5]; r.t[2] = s[6]; r.s = s[7];
  return r;
}

void from_sim3(const sim3::Sim3& r, double s[8]) {
  s[0] = r.q[0]; s[1] = r.q[1]; s[2] = r.q[2]; s[3] = r.q[3];
  s[4] = r.t[0]; s[5] = r.t[1]; s[6] = r.t[2]; s[7] = r.s;
}

bool is_identity77(const double* m) {
  for (int c = 0; c < 7; ++c)
    for (int r = 0; r < 7; ++r)
      if (m[7 * c + r] != (r == c ? 1.0 : 0.0)) return false;
  return true;
}

// pulls the current estimates back into the host container (so get/set work either side of
// initialize, like g2o's vertex objects)
int sync_host_states(sim3opt_graph* g) {
  if (!g->initialized) return SIM3OPT_OK;
  return engine_get_states(g->engine, g->host.states.data(), g->err);
}

int add_edge_impl(sim3opt_graph* g, int32_t id0, int32_t id1, const double* meas,
                  const double* info, int32_t kernel, double kdelta) {
  auto a = g->host.id2idx.find(id0), b = g->host.id2idx.find(id1);
  if (a == g->host.id2idx.end() || b == g->host.id2idx.end())
    return fail(g, SIM3OPT_ERR_ARG, "add_edge: unknown vertex id");
  if (a->second == b->second) return fail(g, SIM3OPT_ERR_ARG, "add_edge: identical endpoints");
  if (!state_ok(meas)) return fail(g, SIM3OPT_ERR_ARG, "add_edge: non-finite measurement or scale <= 0");
  if (kernel != SIM3OPT_KERNEL_NONE && kernel != SIM3OPT_KERNEL_HUBER)
    return fail(g, SIM3OPT_ERR_ARG, "add_edge: unknown robust kernel");
  if (kernel == SIM3OPT_KERNEL_HUBER && !(kdelta > 0.0))
    return fail(g, SIM3OPT_ERR_ARG, "add_edge: Huber delta must be > 0");
  HostGraph& h = g->host;
  const size_t m = h.ev0.size();
  const bool nonident = info && !is_identity77(info);
  if (nonident && !h.has_info) {  // first non-identity information: materialise I7 for earlier edges
    h.has_info = true;
    h.info.assign(49 * m, 0.0);
    for (size_t k = 0; k < m; ++k)
      for (int d = 0; d < 7; ++d) h.info[49 * k + 8 * d] = 1.0;
  }
  if (h.has_info) {
    const size_t off = h.info.size();
    h.info.resize(off + 49, 0.0);
    if (info) std::memcpy(&h.info[off], info, sizeof(double) * 49);
    else for (int d = 0; d < 7; ++d) h.info[off + 8 * d] = 1.0;
  }
  const bool has_k = kernel == SIM3OPT_KERNEL_HUBER;
  if (has_k && !h.has_kernel) {
    h.has_kernel = true;
    h.kdelta.assign(m, 0.0);
  }
  if (h.has_kernel) h.kdelta.push_back(has_k ? kdelta : 0.0);
  h.ev0.push_back(a->second);
  h.ev1.push_back(b->second);
  h.meas.push_back(to_sim3(meas));
  return SIM3OPT_OK;
}

}  // namespace

extern "C" {

int sim3opt_version(void) { return 110; }  // 1.1: multigrid preconditioner, hierarchy / BAL entry points

void sim3opt_options_default(sim3opt_options* o) {
  if (!o) return;
  o->tau = 1e-5;
  o->user_lambda_init = 0.0;
  o->good_step_lower = 1.0 / 3.0;
  o->good_step_upper = 2.0 / 3.0;
  o->max_trials = 10;
  o->fd_delta = 1e-9;
  o->exp_eps = 1e-5;
  o->small_rot_half = 0;
  o->fix_small_angle_b = 0;
  o->dof_mask = 127;
  o->pcg_max_iters = 0;
  o->pcg_rel_tol = 1e-10;
  o->pcg_check_every = 16;
  o->pcg_graph = 1;
  o->preconditioner = -1;
  o->chain_segment = 256;
  o->device = -1;
  o->verbose = 0;
  o->time_kernels = 0;
  o->linear_solver = -1;
  for (int32_t& v : o->amg_cycle) v = 0;
  for (int32_t& v : o->amg_passes) v = 0;
  o->amg_additive = 0;
  o->amg_fp32 = 1;
  o->amg_pivot = 14;
  o->amg_coarsest = 256;
  o->adaptive_prec = 1;
  o->row_order = -1;
  o->halo_exchange = 1;
  o->span_grid = 0;
  o->force_collectives = 0;
  o->amg_shard_rows = 4096;
  o->amg_virtual_ranks = 0;
  o->pcg_batch = 0;
  o->amg_omega = 0.9;
  o->amg_over[0] = 1.8;
  o->amg_over[1] = 1.6;
  o->direct_max_pairs = 0;
  o->debug_full_arrays = 0;
}

// Debug overrides: a SIM3OPT_* environment variable replaces the option field of the same name when the
// graph is initialised (tests and tuning scripts drive the library that way); the override lands in the
// handle's options, so sim3opt_get_options reports what was used.
static void apply_env_overrides(sim3opt_options& o) {
  auto digits = [](const char* ev, int32_t* dst, int n, int lo, int hi) {
    int last = 0;
    const int len = (int)std::strlen(ev);
    for (int l = 0; l < n; ++l) {
      if (l < len && ev[l] - '0' >= lo && ev[l] - '0' <= hi) last = ev[l] - '0';
      dst[l] = last;
    }
  };
  if (const char* ev = std::getenv("SIM3OPT_AMG_CYCLE")) digits(ev, o.amg_cycle, 4, 1, 3);    // "122": levels 1, 2, 3...
  if (const char* ev = std::getenv("SIM3OPT_AMG_PASSES")) digits(ev, o.amg_passes, 3, 1, 6);  // "344"
  if (const char* ev = std::getenv("SIM3OPT_AMG_ADDITIVE")) o.amg_additive = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SIM3OPT_AMG_FP32")) o.amg_fp32 = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SIM3OPT_AMG_PIVOT")) o.amg_pivot = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_AMG_COARSEST")) o.amg_coarsest = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_ADAPTIVE_PREC")) o.adaptive_prec = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("SIM3OPT_ROW_ORDER")) o.row_order = std::string(ev) == "bfs" ? 1 : 0;
  if (std::getenv("SIM3OPT_NO_HALO")) o.halo_exchange = 0;
  if (const char* ev = std::getenv("SIM3OPT_SPAN_GRID")) o.span_grid = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_FORCE_COMM")) o.force_collectives = ev[0] == '1';
  if (const char* ev = std::getenv("SIM3OPT_AMG_SHARD_ROWS")) o.amg_shard_rows = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_AMG_VIRTUAL_RANKS")) o.amg_virtual_ranks = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_PCG_BATCH")) o.pcg_batch = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_AMG_OMEGA")) o.amg_omega = std::atof(ev);
  if (const char* ev = std::getenv("SIM3OPT_AMG_OVER")) {  // "a0[,a1]": the last value repeats
    o.amg_over[0] = o.amg_over[1] = std::atof(ev);
    if (const char* c = std::strchr(ev, ',')) o.amg_over[1] = std::atof(c + 1);
  }
  if (const char* ev = std::getenv("SIM3OPT_DIRECT_MAX_PAIRS")) o.direct_max_pairs = std::atoll(ev);
  if (const char* ev = std::getenv("SIM3OPT_DEBUG_FULL_ARRAYS")) o.debug_full_arrays = std::atoi(ev) != 0;
}

sim3opt_graph* sim3opt_create(void) {
  sim3opt_graph* g = new (std::nothrow) sim3opt_graph();
  if (g) {
    sim3opt_options_default(&g->opt);
    sim3opt::handle_count(+1);
  }
  return g;
}

void sim3opt_destroy(sim3opt_graph* g) {
  if (!g) return;
  delete g;
  if (sim3opt::handle_count(-1) == 0) sim3opt::dev_cache_release();  // the last handle of the process
}

void sim3opt_release_device_cache(void) { sim3opt::dev_cache_release(); }

int sim3opt_set_options(sim3opt_graph* g, const sim3opt_options* o) {
  if (!g || !o) return fail(g, SIM3OPT_ERR_ARG, "set_options: null argument");
  if (!(o->fd_delta > 0) || !(o->exp_eps > 0) || o->max_trials < 1 || !(o->pcg_rel_tol >= 0) ||
      !(o->tau > 0) || o->pcg_check_every < 0 || o->amg_virtual_ranks < 0 || o->pcg_batch < 0 ||
      o->direct_max_pairs < 0 || !(o->amg_omega > 0) || !(o->amg_over[0] > 0) || !(o->amg_over[1] > 0))
    return fail(g, SIM3OPT_ERR_ARG, "set_options: value out of range");
  g->opt = *o;
  if (g->engine) engine_set_options(g->engine, g->opt);
  return SIM3OPT_OK;
}

int sim3opt_get_options(const sim3opt_graph* g, sim3opt_options* o) {
  if (!g || !o) return SIM3OPT_ERR_ARG;
  *o = g->opt;
  return SIM3OPT_OK;
}

const char* sim3opt_last_error(const sim3opt_graph* g) { return g ? g->err.c_str() : "null graph"; }

int sim3opt_add_vertex(sim3opt_graph* g, int32_t id, const double state[8], int32_t fixed) {
  try {
  if (!g || !state) return fail(g, SIM3OPT_ERR_ARG, "add_vertex: null argument");
  if (g->initialized) g->dirty = true;  // needs initializeOptimization() again, like g2o
  if (!state_ok(state)) return fail(g, SIM3OPT_ERR_ARG, "add_vertex: non-finite state or scale <= 0");
  HostGraph& h = g->host;
  if (!h.id2idx.emplace(id, (int32_t)h.vid.size()).second)
    return fail(g, SIM3OPT_ERR_ARG, "add_vertex: duplicate id");  // g2o addVertex returns false
  h.vid.push_back(id);
  h.states.push_back(to_sim3(state));
  h.fixed.push_back(fixed ? 1 : 0);
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "add_vertex: out of host memory or internal error");
  }
}

int sim3opt_add_vertices(sim3opt_graph* g, int32_t n, const int32_t* ids, const double* states,
                         const uint8_t* fixed) {
  try {
  if (!g || n < 0 || (n > 0 && !states)) return fail(g, SIM3OPT_ERR_ARG, "add_vertices: bad argument");
  HostGraph& h = g->host;
  h.vid.reserve(h.vid.size() + n);
  h.states.reserve(h.states.size() + n);
  h.fixed.reserve(h.fixed.size() + n);
  h.id2idx.reserve(h.id2idx.size() + n);
  const int32_t base = (int32_t)h.vid.size();
  for (int32_t k = 0; k < n; ++k) {
    const int rc = sim3opt_add_vertex(g, ids ? ids[k] : base + k, states + 8 * (size_t)k,
                                      fixed ? fixed[k] : 0);
    if (rc != SIM3OPT_OK) return rc;
  }
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "add_vertices: out of host memory or internal error");
  }
}

int sim3opt_add_edge(sim3opt_graph* g, int32_t id_v0, int32_t id_v1, const double meas[8],
                     const double* info77, int32_t kernel, double kernel_delta) {
  try {
  if (!g || !meas) return fail(g, SIM3OPT_ERR_ARG, "add_edge: null argument");
  if (g->initialized) g->dirty = true;
  return add_edge_impl(g, id_v0, id_v1, meas, info77, kernel, kernel_delta);
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "add_edge: out of host memory or internal error");
  }
}

int sim3opt_add_edges(sim3opt_graph* g, int32_t m, const int32_t* id_v0, const int32_t* id_v1,
                      const double* meas, const double* info, int32_t kernel,
                      double kernel_delta) {
  try {
  if (!g || m < 0 || (m > 0 && (!id_v0 || !id_v1 || !meas)))
    return fail(g, SIM3OPT_ERR_ARG, "add_edges: bad argument");
  if (g->initialized) g->dirty = true;
  HostGraph& h = g->host;
  h.ev0.reserve(h.ev0.size() + m);
  h.ev1.reserve(h.ev1.size() + m);
  h.meas.reserve(h.meas.size() + m);
  for (int32_t k = 0; k < m; ++k) {
    const int rc = add_edge_impl(g, id_v0[k], id_v1[k], meas + 8 * (size_t)k,
                                 info ? info + 49 * (size_t)k : nullptr, kernel, kernel_delta);
    if (rc != SIM3OPT_OK) return rc;
  }
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "add_edges: out of host memory or internal error");
  }
}

int32_t sim3opt_num_vertices(const sim3opt_graph* g) { return g ? g->host.nv() : 0; }
int32_t sim3opt_num_edges(const sim3opt_graph* g) { return g ? g->host.ne() : 0; }

int sim3opt_get_edge(const sim3opt_graph* g, int32_t k, int32_t* id_v0, int32_t* id_v1,
                     double meas[8]) {
  try {
  if (!g || k < 0 || k >= g->host.ne()) return SIM3OPT_ERR_ARG;
  if (id_v0) *id_v0 = g->host.vid[g->host.ev0[k]];
  if (id_v1) *id_v1 = g->host.vid[g->host.ev1[k]];
  if (meas) from_sim3(g->host.meas[k], meas);
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return SIM3OPT_ERR_ARG;
  }
}

int sim3opt_initialize(sim3opt_graph* g) {
  try {
  if (!g) return SIM3OPT_ERR_ARG;
  const bool trace = std::getenv("SIM3OPT_INIT_TRACE") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  double t1 = t0, t2 = t0;
  if (g->initialized) {  // g2o allows re-initialisation: rebuild from the current estimates
    int rc = sync_host_states(g);
    if (rc) return rc;
    engine_take_comm(g->engine, &g->comm);  // a multi-GPU graph stays partitioned after re-init
    g->comm_set = g->comm.world > 1 || g->comm.force;
    engine_destroy(g->engine);
    g->engine = nullptr;
    g->initialized = false;
    g->dirty = false;
  }
  t1 = now();
  {
    // a graph that is row-partitioned over several ranks gets its block rows in locality order
    // (contiguous rank spans are then slabs of the graph: few cut edges, a small halo); one rank
    // keeps g2o's insertion order (options.row_order overrides: tests, measurements).
    apply_env_overrides(g->opt);
    const bool local = g->opt.row_order >= 0 ? g->opt.row_order == 1 : (g->comm_set && g->comm.world > 1);
    std::vector<int32_t> order;
    if (local) locality_order(g->host, order);
    if (!build_structure(g->host, g->structure, g->err, local ? &order : nullptr)) return SIM3OPT_ERR_STATE;
  }
  t2 = now();
  int status = SIM3OPT_OK;
  g->engine = engine_create(g->host, g->structure, g->opt, g->comm_set ? &g->comm : nullptr,
                            g->err, status);
  if (trace)
    std::fprintf(stderr, "sim3opt_initialize: tear-down %.2f ms, structure %.2f ms, engine %.2f ms\n", t1 - t0, t2 - t1,
                 now() - t2);
  g->comm_set = false;
  if (!g->engine) return status;
  g->initialized = true;
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "initialize: out of host memory or internal error");
  }
}

int sim3opt_optimize(sim3opt_graph* g, int32_t max_iters) {
  try {
  if (!g) return 0;
  if (!g->initialized) {
    // g2o: optimize() on an uninitialised / empty problem returns -1
    if (g->host.ne() == 0 || g->host.nv() == 0) { g->err = "optimize: nothing to optimise"; return -1; }
    g->err = "optimize: call sim3opt_initialize first";
    return 0;
  }
  if (max_iters <= 0) return 0;
  if (g->dirty) { g->err = "optimize: graph changed, call sim3opt_initialize again"; return 0; }
  const int rc = engine_optimize(g->engine, max_iters, g->stats, g->err);
  return rc < 0 ? 0 : rc;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    (void)fail(g, SIM3OPT_ERR_ARG, "optimize: out of host memory or internal error"); return 0;
  }
}

int sim3opt_get_vertex(sim3opt_graph* g, int32_t id, double state[8]) {
  if (!g || !state) return fail(g, SIM3OPT_ERR_ARG, "get_vertex: null argument");
  auto it = g->host.id2idx.find(id);
  if (it == g->host.id2idx.end()) return fail(g, SIM3OPT_ERR_ARG, "get_vertex: unknown id");
  int rc = sync_host_states(g);
  if (rc) return rc;
  from_sim3(g->host.states[it->second], state);
  return SIM3OPT_OK;
}

int sim3opt_set_vertex(sim3opt_graph* g, int32_t id, const double state[8]) {
  if (!g || !state) return fail(g, SIM3OPT_ERR_ARG, "set_vertex: null argument");
  auto it = g->host.id2idx.find(id);
  if (it == g->host.id2idx.end()) return fail(g, SIM3OPT_ERR_ARG, "set_vertex: unknown id");
  if (!state_ok(state)) return fail(g, SIM3OPT_ERR_ARG, "set_vertex: non-finite state or scale <= 0");
  int rc = sync_host_states(g);
  if (rc) return rc;
  g->host.states[it->second] = to_sim3(state);
  if (g->initialized) return engine_set_states(g->engine, g->host.states.data(), g->err);
  return SIM3OPT_OK;
}

int sim3opt_get_vertices(sim3opt_graph* g, double* states) {
  if (!g || !states) return fail(g, SIM3OPT_ERR_ARG, "get_vertices: null argument");
  int rc = sync_host_states(g);
  if (rc) return rc;
  for (size_t k = 0; k < g->host.states.size(); ++k) from_sim3(g->host.states[k], states + 8 * k);
  return SIM3OPT_OK;
}

int sim3opt_set_vertices(sim3opt_graph* g, const double* states) {
  if (!g || !states) return fail(g, SIM3OPT_ERR_ARG, "set_vertices: null argument");
  for (size_t k = 0; k < g->host.states.size(); ++k)
    if (!state_ok(states + 8 * k)) return fail(g, SIM3OPT_ERR_ARG, "set_vertices: bad state");
  for (size_t k = 0; k < g->host.states.size(); ++k) g->host.states[k] = to_sim3(states + 8 * k);
  if (g->initialized) return engine_set_states(g->engine, g->host.states.data(), g->err);
  return SIM3OPT_OK;
}

int sim3opt_chi2(sim3opt_graph* g, double* chi2) {
  if (!g || !chi2) return fail(g, SIM3OPT_ERR_ARG, "chi2: null argument");
  if (!g->initialized || g->dirty) return fail(g, SIM3OPT_ERR_STATE, "chi2: call sim3opt_initialize first");
  return engine_chi2(g->engine, chi2, g->err);
}

int32_t sim3opt_num_iterations(const sim3opt_graph* g) { return g ? (int32_t)g->stats.size() : 0; }

int sim3opt_get_stats(const sim3opt_graph* g, int32_t iter, sim3opt_iter_stats* out) {
  if (!g || !out || iter < 0 || iter >= (int32_t)g->stats.size()) return SIM3OPT_ERR_ARG;
  *out = g->stats[iter];
  return SIM3OPT_OK;
}

int sim3opt_get_kernel_times(sim3opt_graph* g, sim3opt_kernel_times* out) {
  if (!g || !out) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "kernel_times: not initialized");
  return engine_kernel_times(g->engine, out, false);
}

int sim3opt_get_comm_times(sim3opt_graph* g, sim3opt_comm_times* out) {
  if (!g || !out) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "comm_times: not initialized");
  return engine_comm_times(g->engine, out);
}

int sim3opt_reset_kernel_times(sim3opt_graph* g) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "kernel_times: not initialized");
  return engine_kernel_times(g->engine, nullptr, true);
}

int sim3opt_edge_errors(sim3opt_graph* g, double* e_out) {
  if (!g || !e_out) return fail(g, SIM3OPT_ERR_ARG, "edge_errors: null argument");
  if (!g->initialized || g->dirty) return fail(g, SIM3OPT_ERR_STATE, "edge_errors: call sim3opt_initialize first");
  return engine_edge_errors(g->engine, e_out, g->err);
}

int sim3opt_linearize(sim3opt_graph* g) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized || g->dirty) return fail(g, SIM3OPT_ERR_STATE, "linearize: call sim3opt_initialize first");
  return engine_linearize(g->engine, g->err);
}

int sim3opt_system_dims(const sim3opt_graph* g, int32_t* n_block_rows, int64_t* n_blocks) {
  if (!g || !g->initialized) return SIM3OPT_ERR_STATE;
  if (n_block_rows) *n_block_rows = g->structure.nb;
  if (n_blocks) *n_blocks = g->structure.nnzb;
  return SIM3OPT_OK;
}

int sim3opt_system_pattern(sim3opt_graph* g, int32_t* n_block_rows, int64_t* n_blocks,
                           int32_t* rowptr, int32_t* colidx) {
  try {
  if (!g) return SIM3OPT_ERR_ARG;
  Structure st;
  if (!build_structure(g->host, st, g->err)) return SIM3OPT_ERR_STATE;
  if (n_block_rows) *n_block_rows = st.nb;
  if (n_blocks) *n_blocks = st.nnzb;
  if (rowptr) std::memcpy(rowptr, st.rowptr.data(), sizeof(int32_t) * (size_t)(st.nb + 1));
  if (colidx) std::memcpy(colidx, st.colidx.data(), sizeof(int32_t) * (size_t)st.nnzb);
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "system_pattern: out of host memory or internal error");
  }
}

int sim3opt_get_system(sim3opt_graph* g, int32_t* rowptr, int32_t* colidx, double* values,
                       double* b) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "get_system: call sim3opt_initialize first");
  return engine_get_system(g->engine, rowptr, colidx, values, b, g->err);
}

int sim3opt_solve(sim3opt_graph* g, double lambda, double* x, int32_t* iters, double* rel_res) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "solve: call sim3opt_initialize first");
  return engine_solve(g->engine, lambda, x, iters, rel_res, g->err);
}

int sim3opt_bench_spmv(sim3opt_graph* g, int32_t reps, double* ms_mean) {
  if (!g || !ms_mean || reps < 1) return fail(g, SIM3OPT_ERR_ARG, "bench_spmv: bad argument");
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "bench_spmv: call sim3opt_initialize first");
  return engine_bench_spmv(g->engine, reps, ms_mean, g->err);
}

int sim3opt_preconditioner_in_use(const sim3opt_graph* g) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return SIM3OPT_ERR_STATE;
  return engine_preconditioner(g->engine);
}

int sim3opt_amg_in_use(const sim3opt_graph* g, int32_t* n_levels, int32_t* n_partitioned, int32_t visits[4]) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return SIM3OPT_ERR_STATE;
  engine_amg_in_use(g->engine, n_levels, n_partitioned, visits);
  return SIM3OPT_OK;
}

int sim3opt_device_bytes(const sim3opt_graph* g, int64_t bytes[2]) {
  if (!g || !bytes) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return SIM3OPT_ERR_STATE;
  engine_device_bytes(g->engine, bytes);
  return SIM3OPT_OK;
}

int sim3opt_linear_solver_in_use(const sim3opt_graph* g) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (!g->initialized) return SIM3OPT_ERR_STATE;
  return engine_linear_solver(g->engine);
}

int sim3opt_direct_plan(sim3opt_graph* g, int64_t max_pairs, int64_t dims[8], int32_t* perm,
                        int32_t* colptr, int32_t* lrow, int32_t* srcptr, int32_t* src,
                        int32_t* pairptr, int32_t* pa, int32_t* pb, int32_t* gptr, int32_t* lcolp,
                        int32_t* rptr, int32_t* cells) {
  try {
  if (!g || !dims) return fail(g, SIM3OPT_ERR_ARG, "direct_plan: bad argument");
  Structure st;
  if (!build_structure(g->host, st, g->err)) return SIM3OPT_ERR_STATE;
  DirectPlan P;
  std::string why;
  if (!build_direct_plan(st.nb, st.rowptr.data(), st.colidx.data(), max_pairs > 0 ? max_pairs : 300000, 0,
                         P, why)) {
    g->err = "direct_plan: " + why;
    return SIM3OPT_ERR_STATE;
  }
  dims[0] = P.nb; dims[1] = P.nL; dims[2] = P.npairs; dims[3] = P.height; dims[4] = P.ngroups();
  dims[5] = (int64_t)P.lcolp.size() - 1; dims[6] = (int64_t)P.src.size();
  dims[7] = (int64_t)P.cells.size() / DirectPlan::CELL_STRIDE;
  auto out = [](int32_t* dst, const std::vector<int32_t>& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), sizeof(int32_t) * v.size());
  };
  out(perm, P.perm); out(colptr, P.colptr); out(lrow, P.lrow); out(srcptr, P.srcptr); out(src, P.src);
  out(pairptr, P.pairptr); out(pa, P.pa); out(pb, P.pb); out(gptr, P.gptr); out(lcolp, P.lcolp);
  out(rptr, P.rptr); out(cells, P.cells);
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "direct_plan: out of host memory or internal error");
  }
}

int sim3opt_amg_hierarchy(sim3opt_graph* g, int32_t capacity, int32_t* n_levels, int32_t* rows,
                          int64_t* blocks, int32_t* aggregate_of_row) {
  try {
  if (!g || !n_levels || capacity < 0) return fail(g, SIM3OPT_ERR_ARG, "amg_hierarchy: bad argument");
  // (the hierarchy sim3opt_initialize would build on ONE rank with the handle's options: row order,
  // matching passes, dense-level cap and, with amg_virtual_ranks, the aggregation of an N-rank partition)
  sim3opt_options o = g->opt;
  apply_env_overrides(o);
  Structure st;
  std::vector<int32_t> order;
  if (o.row_order == 1) locality_order(g->host, order);
  if (!build_structure(g->host, st, g->err, o.row_order == 1 ? &order : nullptr)) return SIM3OPT_ERR_STATE;
  std::vector<AmgLevelHost> levels;
  std::string why;
  AmgBuildOptions bo;
  bo.max_coarsest = o.amg_coarsest;
  for (int k = 0; k < 3; ++k) bo.passes[k] = o.amg_passes[k];
  std::vector<int32_t> vbegin;
  if (o.amg_virtual_ranks > 1) {
    bo.world = o.amg_virtual_ranks;
    bo.shard_rows = std::max(1, o.amg_shard_rows);
    vbegin.resize(bo.world + 1);
    partition_rows_equal(st.nb, bo.world, vbegin.data());
    bo.row_begin = vbegin.data();
  }
  if (!build_amg_hierarchy(st.nb, st.rowptr.data(), st.colidx.data(), levels, why, bo)) {
    g->err = "amg_hierarchy: " + why;
    return SIM3OPT_ERR_STATE;
  }
  *n_levels = (int32_t)levels.size();
  for (int32_t l = 0; l < *n_levels && l < capacity; ++l) {
    if (rows) rows[l] = levels[l].nb;
    if (blocks) blocks[l] = levels[l].nnzb;
  }
  if (aggregate_of_row) std::memcpy(aggregate_of_row, levels[0].agg.data(), sizeof(int32_t) * (size_t)st.nb);
  return SIM3OPT_OK;
  } catch (...) {  // (std::bad_alloc, std::length_error ...: nothing crosses the C boundary)
    return fail(g, SIM3OPT_ERR_ARG, "amg_hierarchy: out of host memory or internal error");
  }
}

int sim3opt_partition_plan(sim3opt_graph* g, int32_t world, int32_t locality, int32_t* vertex_of_row,
                           int32_t* row_begin, int32_t* boundary_rows_of_rank, int64_t* cut_edges) {
  try {
    if (!g || world < 1) return fail(g, SIM3OPT_ERR_ARG, "partition_plan: bad argument");
    std::vector<int32_t> order;
    if (locality) locality_order(g->host, order);
    Structure st;
    if (!build_structure(g->host, st, g->err, locality ? &order : nullptr)) return SIM3OPT_ERR_STATE;
    std::vector<int32_t> begin(world + 1), rows, seg;
    partition_rows_equal(st.nb, world, begin.data());
    boundary_rows(st.nb, st.rowptr.data(), st.colidx.data(), world, begin.data(), rows, seg);
    if (vertex_of_row) std::memcpy(vertex_of_row, st.row2vertex.data(), sizeof(int32_t) * (size_t)st.nb);
    if (row_begin) std::memcpy(row_begin, begin.data(), sizeof(int32_t) * (size_t)(world + 1));
    if (boundary_rows_of_rank)
      for (int32_t r = 0; r < world; ++r) boundary_rows_of_rank[r] = seg[r + 1] - seg[r];
    if (cut_edges) {
      auto owner = [&](int32_t row) {
        return (int32_t)(std::upper_bound(begin.begin(), begin.end(), row) - begin.begin()) - 1;
      };
      int64_t cut = 0;
      for (int32_t k : st.active) {
        const int32_t a = st.hidx[g->host.ev0[k]], b = st.hidx[g->host.ev1[k]];
        if (a >= 0 && b >= 0 && owner(a) != owner(b)) ++cut;
      }
      *cut_edges = cut;
    }
    return SIM3OPT_OK;
  } catch (...) {
    return fail(g, SIM3OPT_ERR_ARG, "partition_plan: out of host memory or internal error");
  }
}

int sim3opt_bench_stream(sim3opt_graph* g, int32_t mode, int32_t reps, double* ms_mean) {
  if (!g || !ms_mean || reps < 1 || mode < 0 || mode > 2) return fail(g, SIM3OPT_ERR_ARG, "bench_stream: bad argument");
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "bench_stream: call sim3opt_initialize first");
  return engine_bench_stream(g->engine, mode, reps, ms_mean, g->err);
}

#ifdef SIM3OPT_BENCH_HOOKS  // (measurement prototype: not in the product library, see engine_proto.hip)
int sim3opt_bench_spmv_symmetric(sim3opt_graph* g, int32_t reps, double out[4]) {
  if (!g || !out || reps < 1) return fail(g, SIM3OPT_ERR_ARG, "bench_spmv_symmetric: bad argument");
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "bench_spmv_symmetric: call sim3opt_initialize first");
  return engine_bench_spmv_symmetric(g->engine, reps, out, g->err);
}
int sim3opt_bench_spmv_rowlane(sim3opt_graph* g, int32_t reps, int32_t rows_per_group, double out[8]) {
  if (!g || !out || reps < 1) return fail(g, SIM3OPT_ERR_ARG, "bench_spmv_rowlane: bad argument");
  if (!g->initialized) return fail(g, SIM3OPT_ERR_STATE, "bench_spmv_rowlane: call sim3opt_initialize first");
  return engine_bench_spmv_rowlane(g->engine, reps, rows_per_group, out, g->err);
}
#endif

int sim3opt_partition_rows(int32_t n_block_rows, const int32_t* rowptr, int32_t world,
                           int32_t* row_begin) {
  if (n_block_rows < 0 || !rowptr || world < 1 || !row_begin) return SIM3OPT_ERR_ARG;
  partition_rows(n_block_rows, rowptr, world, row_begin);
  return SIM3OPT_OK;
}

int sim3opt_partition_rows_equal(int32_t n_block_rows, int32_t world, int32_t* row_begin) {
  if (n_block_rows < 0 || world < 1 || !row_begin) return SIM3OPT_ERR_ARG;
  partition_rows_equal(n_block_rows, world, row_begin);
  return SIM3OPT_OK;
}

int sim3opt_comm_allgather_plan(int32_t n_block_rows, int32_t world, int32_t* row_begin,
                                int64_t* count, int64_t* padded_len) {
  if (n_block_rows < 0 || world < 1) return SIM3OPT_ERR_ARG;
  std::vector<int32_t> rb(world + 1);
  partition_rows_equal(n_block_rows, world, rb.data());
  std::vector<int64_t> offs(world + 1);
  for (int r = 0; r <= world; ++r) offs[r] = 7 * (int64_t)rb[r];
  if (row_begin) std::memcpy(row_begin, rb.data(), sizeof(int32_t) * (size_t)(world + 1));
  return allgather_equal_plan(offs.data(), world, count, padded_len) ? 1 : 0;
}

int sim3opt_comm_unique_id(uint8_t id_out[128]) {
  if (!id_out) return SIM3OPT_ERR_ARG;
  std::string err;
  return comm_unique_id(id_out, err);
}

int sim3opt_comm_init(sim3opt_graph* g, int32_t rank, int32_t world, const uint8_t unique_id[128]) {
  if (!g || world < 1 || rank < 0 || rank >= world) return fail(g, SIM3OPT_ERR_ARG, "comm_init: bad rank/world");
  if (g->initialized) return fail(g, SIM3OPT_ERR_STATE, "comm_init: call before sim3opt_initialize");
  // options.force_collectives: build the communicator and run every collective even with one rank
  // (self-test of the RCCL transport on a single-GPU machine)
  apply_env_overrides(g->opt);
  const bool force = g->opt.force_collectives != 0;
  if (world == 1 && !force) return SIM3OPT_OK;
  if (!unique_id) return fail(g, SIM3OPT_ERR_ARG, "comm_init: null unique id");
  if (g->opt.device >= 0 && hipSetDevice(g->opt.device) != hipSuccess)
    return fail(g, SIM3OPT_ERR_HIP, "comm_init: hipSetDevice failed");
  g->comm.release();
  const int rc = comm_init_rccl(g->comm, rank, world, unique_id, g->err);
  g->comm.force = force;
  g->comm_set = rc == SIM3OPT_OK;
  return rc;
}

int sim3opt_comm_init_callbacks(sim3opt_graph* g, int32_t rank, int32_t world,
                                sim3opt_allreduce_fn allreduce, sim3opt_allgatherv_fn allgatherv,
                                void* ctx) {
  if (!g || world < 1 || rank < 0 || rank >= world || !allreduce || !allgatherv)
    return fail(g, SIM3OPT_ERR_ARG, "comm_init_callbacks: bad argument");
  if (g->initialized) return fail(g, SIM3OPT_ERR_STATE, "comm_init_callbacks: call before sim3opt_initialize");
  g->comm.release();
  g->comm.rank = rank;
  g->comm.world = world;
  g->comm.kind = 2;
  g->comm.cb_allreduce = allreduce;
  g->comm.cb_allgatherv = allgatherv;
  g->comm.cb_ctx = ctx;
  g->comm_set = world > 1;
  return SIM3OPT_OK;
}

int sim3opt_halo_plan(sim3opt_graph* g, int32_t world, int32_t rank, int32_t* n_send, int32_t* n_recv,
                      int32_t* send_rows, int32_t* send_seg, int32_t* recv_rows, int32_t* recv_seg) {
  try {
    if (!g || world < 1 || rank < 0 || rank >= world) return fail(g, SIM3OPT_ERR_ARG, "halo_plan: bad argument");
    std::vector<int32_t> order;
    locality_order(g->host, order);
    Structure st;
    if (!build_structure(g->host, st, g->err, &order)) return SIM3OPT_ERR_STATE;
    std::vector<int32_t> begin(world + 1), sr, ss, rr, rs;
    partition_rows_equal(st.nb, world, begin.data());
    halo_plan(st.nb, st.rowptr.data(), st.colidx.data(), world, begin.data(), rank, sr, ss, rr, rs);
    if (n_send) *n_send = (int32_t)sr.size();
    if (n_recv) *n_recv = (int32_t)rr.size();
    if (send_rows) std::memcpy(send_rows, sr.data(), sizeof(int32_t) * sr.size());
    if (recv_rows) std::memcpy(recv_rows, rr.data(), sizeof(int32_t) * rr.size());
    if (send_seg) std::memcpy(send_seg, ss.data(), sizeof(int32_t) * ss.size());
    if (recv_seg) std::memcpy(recv_seg, rs.data(), sizeof(int32_t) * rs.size());
    return SIM3OPT_OK;
  } catch (...) {
    return fail(g, SIM3OPT_ERR_ARG, "halo_plan: out of host memory or internal error");
  }
}

int sim3opt_comm_set_alltoallv(sim3opt_graph* g, sim3opt_alltoallv_fn alltoallv) {
  if (!g) return SIM3OPT_ERR_ARG;
  if (g->initialized) return fail(g, SIM3OPT_ERR_STATE, "comm_set_alltoallv: call before sim3opt_initialize");
  if (g->comm.kind != 2) return fail(g, SIM3OPT_ERR_STATE, "comm_set_alltoallv: call sim3opt_comm_init_callbacks first");
  g->comm.cb_alltoallv = alltoallv;
  return SIM3OPT_OK;
}

int sim3opt_local_rows(const sim3opt_graph* g, int32_t* begin, int32_t* end) {
  if (!g || !g->initialized) return SIM3OPT_ERR_STATE;
  engine_local_rows(g->engine, begin, end);
  return SIM3OPT_OK;
}

}  // extern "C"
