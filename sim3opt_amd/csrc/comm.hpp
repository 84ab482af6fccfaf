// comm.hpp -- collectives of the row-partitioned multi-GPU path (one process per GPU).
//
// Two transports behind one interface:
//   RCCL      in-place ncclAllReduce / ncclAllGather on the engine's HIP stream (xGMI); librccl is
//             dlopen'ed at sim3opt_comm_init so that libsim3opt.so loads on machines without it
//   callbacks host-staged: the engine copies the operands to pinned host memory and calls the
//             user's functions (MPI, torch.distributed/gloo, ...).  Used by the 2-process tests.
// The reference has no communication at all; this is new work.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sim3opt.h"

namespace sim3opt {

struct Comm {
  int32_t rank = 0, world = 1;
  int kind = 0;  // 0 none, 1 RCCL, 2 callbacks
  void* nccl = nullptr;  // ncclComm_t
  sim3opt_allreduce_fn cb_allreduce = nullptr;
  sim3opt_allgatherv_fn cb_allgatherv = nullptr;
  sim3opt_alltoallv_fn cb_alltoallv = nullptr;  // optional: neighbour exchange
  void* cb_ctx = nullptr;
  double* h_stage = nullptr;  // pinned staging buffer (callbacks transport)
  size_t h_stage_len = 0;
  double* h_stage2 = nullptr;  // ... and the receive side of an exchange
  size_t h_stage2_len = 0;

  bool force = false;  // run the collectives even with one rank (transport self-test)
  bool active() const { return world > 1 || force; }
  // optional timing (sim3opt_options.time_kernels): an event pair around every collective on the
  // engine's stream, folded into `times` by drain() after the caller's next stream synchronisation.
  // What a pair measures includes the wait for the slowest peer -- the figure a scaling run needs.
  bool timing = false;
  sim3opt_comm_times times{};
  std::vector<hipEvent_t> ev;     // pool, pairs
  std::vector<int> ev_kind;       // per pair: 0 all-reduce, 1 all-gather, 2 neighbour exchange
  size_t ev_used = 0;
  int drain(std::string& err);
  // in-place on device memory, ordered on `stream`; op: 0 = sum, 1 = max
  int allreduce(double* dptr, int n, int op, hipStream_t stream, std::string& err);
  // in-place all-gather of a vector split at offs[0..world] (in doubles); rank r contributes
  // [offs[r], offs[r+1]).  With equal spans (offs[r] = r * count, short tail allowed) the buffer must
  // hold world * count doubles: the RCCL transport then issues a single ncclAllGather
  int allgatherv(double* dvec, const std::vector<int64_t>& offs, hipStream_t stream,
                 std::string& err);
  // neighbour exchange: sbuf[soffs[p] .. soffs[p+1]) goes to rank p, rbuf[roffs[p] .. roffs[p+1]) comes from
  // rank p (doubles; both plans come from the same symmetric pattern, so the two sides agree on every
  // count).  RCCL: one group of ncclSend / ncclRecv pairs; callbacks: the optional alltoallv callback.
  bool can_exchange() const { return kind == 1 || (kind == 2 && cb_alltoallv != nullptr); }
  int exchange(const double* sbuf, const std::vector<int64_t>& soffs, double* rbuf,
               const std::vector<int64_t>& roffs, hipStream_t stream, std::string& err);
  void release();

 private:
  int exchange_impl(const double* sbuf, const std::vector<int64_t>& soffs, double* rbuf,
                    const std::vector<int64_t>& roffs, hipStream_t stream, std::string& err);
  int allreduce_impl(double* dptr, int n, int op, hipStream_t stream, std::string& err);
  int allgatherv_impl(double* dvec, const std::vector<int64_t>& offs, hipStream_t stream, std::string& err);
  int stamp(int kind_, bool begin, hipStream_t stream, std::string& err);
};

// Plan of the in-place all-gather of a vector split at offs[0..world] (in doubles): true when the
// spans are the engine's equal-count partition -- offs[r] = r * count, trailing ranks short or empty
// -- so that ONE ncclAllGather of `count` doubles per rank does the exchange; the buffer must then
// hold `padded_len` = world * count doubles (>= offs[world]: the tail beyond the vector is padding
// that no kernel reads).
inline bool allgather_equal_plan(const int64_t* offs, int32_t world, int64_t* count, int64_t* padded_len) {
  const int64_t cnt = offs[1] - offs[0];
  bool equal = cnt > 0;
  for (int r = 0; r < world && equal; ++r) equal = offs[r] == (int64_t)r * cnt || offs[r] == offs[world];
  if (count) *count = cnt;
  if (padded_len) *padded_len = equal ? (int64_t)world * cnt : offs[world];
  return equal;
}

int comm_unique_id(uint8_t id_out[128], std::string& err);
int comm_init_rccl(Comm& c, int32_t rank, int32_t world, const uint8_t id[128], std::string& err);

}  // namespace sim3opt
