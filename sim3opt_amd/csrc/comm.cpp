// comm.cpp -- RCCL (dlopen) and host-staged callback transports (see comm.hpp).
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>

namespace sim3opt {

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;
std::mutex g_mu;

bool load_rccl(std::string& err) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_api.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    err = std::string("cannot load librccl: ") + dlerror();
    return false;
  }
#define LOAD(field, sym)                                                   \
  g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(h, sym));    \
  if (!g_api.field) {                                                      \
    err = std::string("librccl lacks ") + sym;                            \
    return false;                                                          \
  }
  LOAD(GetUniqueId, "ncclGetUniqueId")
  LOAD(CommInitRank, "ncclCommInitRank")
  LOAD(CommDestroy, "ncclCommDestroy")
  LOAD(AllReduce, "ncclAllReduce")
  LOAD(Broadcast, "ncclBroadcast")
  LOAD(AllGather, "ncclAllGather")
  LOAD(Send, "ncclSend")
  LOAD(Recv, "ncclRecv")
  LOAD(GroupStart, "ncclGroupStart")
  LOAD(GroupEnd, "ncclGroupEnd")
  LOAD(GetErrorString, "ncclGetErrorString")
#undef LOAD
  g_api.handle = h;
  return true;
}

#define NCCLCHK(call)                                                          \
  do {                                                                         \
    ncclResult_t r_ = (call);                                                  \
    if (r_ != ncclSuccess) {                                                   \
      err = std::string(#call) + ": " + g_api.GetErrorString(r_);              \
      return SIM3OPT_ERR_COMM;                                                 \
    }                                                                          \
  } while (0)

#define HIPCHK(call)                                                           \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
      return SIM3OPT_ERR_HIP;                                                  \
    }                                                                          \
  } while (0)

}  // namespace

int comm_unique_id(uint8_t id_out[128], std::string& err) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!load_rccl(err)) return SIM3OPT_ERR_COMM;
  ncclUniqueId id;
  NCCLCHK(g_api.GetUniqueId(&id));
  std::memcpy(id_out, &id, 128);
  return SIM3OPT_OK;
}

int comm_init_rccl(Comm& c, int32_t rank, int32_t world, const uint8_t idb[128],
                   std::string& err) {
  if (!load_rccl(err)) return SIM3OPT_ERR_COMM;
  ncclUniqueId id;
  std::memcpy(&id, idb, 128);
  ncclComm_t comm = nullptr;
  NCCLCHK(g_api.CommInitRank(&comm, world, id, rank));
  c.rank = rank;
  c.world = world;
  c.kind = 1;
  c.nccl = comm;
  return SIM3OPT_OK;
}

static int ensure_stage(Comm& c, size_t n, std::string& err) {
  if (c.h_stage_len >= n) return SIM3OPT_OK;
  if (c.h_stage) (void)hipHostFree(c.h_stage);
  c.h_stage = nullptr;
  c.h_stage_len = 0;
  HIPCHK(hipHostMalloc((void**)&c.h_stage, sizeof(double) * n));
  c.h_stage_len = n;
  return SIM3OPT_OK;
}

int Comm::stamp(int kind_, bool begin, hipStream_t stream, std::string& err) {
  if (begin) {
    if (ev_used + 2 > ev.size()) {
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0));
      HIPCHK(hipEventCreate(&e1));
      ev.push_back(e0);
      ev.push_back(e1);
    }
    if (ev_kind.size() < ev.size() / 2) ev_kind.resize(ev.size() / 2);
    ev_kind[ev_used / 2] = kind_;
    HIPCHK(hipEventRecord(ev[ev_used], stream));
  } else {
    HIPCHK(hipEventRecord(ev[ev_used + 1], stream));
    ev_used += 2;
  }
  return SIM3OPT_OK;
}

// after a stream synchronisation: elapsed times of the recorded pairs into `times`
int Comm::drain(std::string& err) {
  for (size_t i = 0; i + 1 < ev_used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    if (ev_kind[i / 2] == 0) times.ms_allreduce += ms;
    else if (ev_kind[i / 2] == 1) times.ms_allgather += ms;
    else times.ms_exchange += ms;
  }
  ev_used = 0;
  return SIM3OPT_OK;
}

int Comm::allreduce(double* dptr, int n, int op, hipStream_t stream, std::string& err) {
  if (!active()) return SIM3OPT_OK;
  if (!timing) return allreduce_impl(dptr, n, op, stream, err);
  int rc = stamp(0, true, stream, err);
  if (rc) return rc;
  rc = allreduce_impl(dptr, n, op, stream, err);
  if (rc) return rc;
  times.n_allreduce += 1;
  times.bytes_allreduce += (int64_t)sizeof(double) * n;
  return stamp(0, false, stream, err);
}

int Comm::allgatherv(double* dvec, const std::vector<int64_t>& offs, hipStream_t stream, std::string& err) {
  if (!active()) return SIM3OPT_OK;
  if (!timing) return allgatherv_impl(dvec, offs, stream, err);
  int rc = stamp(1, true, stream, err);
  if (rc) return rc;
  rc = allgatherv_impl(dvec, offs, stream, err);
  if (rc) return rc;
  times.n_allgather += 1;
  times.bytes_allgather += (int64_t)sizeof(double) * offs[world];
  return stamp(1, false, stream, err);
}

int Comm::exchange(const double* sbuf, const std::vector<int64_t>& soffs, double* rbuf,
                   const std::vector<int64_t>& roffs, hipStream_t stream, std::string& err) {
  if (!active()) return SIM3OPT_OK;
  if (!timing) return exchange_impl(sbuf, soffs, rbuf, roffs, stream, err);
  int rc = stamp(2, true, stream, err);
  if (rc) return rc;
  rc = exchange_impl(sbuf, soffs, rbuf, roffs, stream, err);
  if (rc) return rc;
  times.n_exchange += 1;
  times.bytes_exchange += (int64_t)sizeof(double) * (soffs[world] + roffs[world]);
  return stamp(2, false, stream, err);
}

int Comm::exchange_impl(const double* sbuf, const std::vector<int64_t>& soffs, double* rbuf,
                        const std::vector<int64_t>& roffs, hipStream_t stream, std::string& err) {
  if (kind == 1) {
    NCCLCHK(g_api.GroupStart());
    for (int p = 0; p < world; ++p) {
      const size_t ns = (size_t)(soffs[p + 1] - soffs[p]), nr = (size_t)(roffs[p + 1] - roffs[p]);
      ncclResult_t rr = ncclSuccess;
      if (ns) rr = g_api.Send(sbuf + soffs[p], ns, ncclFloat64, p, (ncclComm_t)nccl, stream);
      if (rr == ncclSuccess && nr) rr = g_api.Recv(rbuf + roffs[p], nr, ncclFloat64, p, (ncclComm_t)nccl, stream);
      if (rr != ncclSuccess) {
        (void)g_api.GroupEnd();
        err = std::string("ncclSend / ncclRecv: ") + g_api.GetErrorString(rr);
        return SIM3OPT_ERR_COMM;
      }
    }
    NCCLCHK(g_api.GroupEnd());
    return SIM3OPT_OK;
  }
  if (!cb_alltoallv) {
    err = "neighbour exchange without an alltoallv callback";
    return SIM3OPT_ERR_COMM;
  }
  const size_t ns = (size_t)soffs[world], nr = (size_t)roffs[world];
  int rc = ensure_stage(*this, std::max<size_t>(ns, 1), err);
  if (rc) return rc;
  if (h_stage2_len < std::max<size_t>(nr, 1)) {
    if (h_stage2) (void)hipHostFree(h_stage2);
    h_stage2 = nullptr;
    h_stage2_len = 0;
    HIPCHK(hipHostMalloc((void**)&h_stage2, sizeof(double) * std::max<size_t>(nr, 1)));
    h_stage2_len = std::max<size_t>(nr, 1);
  }
  if (ns) HIPCHK(hipMemcpyAsync(h_stage, sbuf, sizeof(double) * ns, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (cb_alltoallv(cb_ctx, h_stage, soffs.data(), h_stage2, roffs.data(), rank, world) != 0) {
    err = "alltoallv callback failed";
    return SIM3OPT_ERR_COMM;
  }
  if (nr) HIPCHK(hipMemcpyAsync(rbuf, h_stage2, sizeof(double) * nr, hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));
  return SIM3OPT_OK;
}

int Comm::allreduce_impl(double* dptr, int n, int op, hipStream_t stream, std::string& err) {
  if (!active()) return SIM3OPT_OK;
  if (kind == 1) {
    NCCLCHK(g_api.AllReduce(dptr, dptr, (size_t)n, ncclFloat64, op == 1 ? ncclMax : ncclSum,
                            (ncclComm_t)nccl, stream));
    return SIM3OPT_OK;
  }
  int rc = ensure_stage(*this, (size_t)n, err);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h_stage, dptr, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (cb_allreduce(cb_ctx, h_stage, n, op) != 0) {
    err = "allreduce callback failed";
    return SIM3OPT_ERR_COMM;
  }
  HIPCHK(hipMemcpyAsync(dptr, h_stage, sizeof(double) * n, hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));  // the staging buffer is reused by the next call
  return SIM3OPT_OK;
}

int Comm::allgatherv_impl(double* dvec, const std::vector<int64_t>& offs, hipStream_t stream,
                          std::string& err) {
  if (!active()) return SIM3OPT_OK;
  if (kind == 1) {
    // equal spans (the engine's partition; buffers are padded to world x count): ONE in-place
    // ncclAllGather -- rank r's segment already sits at recvbuff + r * count
    int64_t cnt = 0;
    if (allgather_equal_plan(offs.data(), world, &cnt, nullptr)) {
      NCCLCHK(g_api.AllGather(dvec + (int64_t)rank * cnt, dvec, (size_t)cnt, ncclFloat64,
                              (ncclComm_t)nccl, stream));
      return SIM3OPT_OK;
    }
    // general spans (not produced by the engine's partition; kept for callers of the transport with
    // their own split): one grouped broadcast per owner
    NCCLCHK(g_api.GroupStart());
    for (int r = 0; r < world; ++r) {
      const size_t cnt = (size_t)(offs[r + 1] - offs[r]);
      if (cnt == 0) continue;
      ncclResult_t rr = g_api.Broadcast(dvec + offs[r], dvec + offs[r], cnt, ncclFloat64, r,
                                        (ncclComm_t)nccl, stream);
      if (rr != ncclSuccess) {
        (void)g_api.GroupEnd();
        err = std::string("ncclBroadcast: ") + g_api.GetErrorString(rr);
        return SIM3OPT_ERR_COMM;
      }
    }
    NCCLCHK(g_api.GroupEnd());
    return SIM3OPT_OK;
  }
  const size_t total = (size_t)offs[world];
  int rc = ensure_stage(*this, total, err);
  if (rc) return rc;
  const size_t mine = (size_t)(offs[rank + 1] - offs[rank]);
  if (mine)
    HIPCHK(hipMemcpyAsync(h_stage + offs[rank], dvec + offs[rank], sizeof(double) * mine,
                          hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (cb_allgatherv(cb_ctx, h_stage, offs.data(), rank, world) != 0) {
    err = "allgatherv callback failed";
    return SIM3OPT_ERR_COMM;
  }
  HIPCHK(hipMemcpyAsync(dvec, h_stage, sizeof(double) * total, hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));
  return SIM3OPT_OK;
}

void Comm::release() {
  if (kind == 1 && nccl && g_api.CommDestroy) (void)g_api.CommDestroy((ncclComm_t)nccl);
  nccl = nullptr;
  if (h_stage) (void)hipHostFree(h_stage);
  h_stage = nullptr;
  h_stage_len = 0;
  if (h_stage2) (void)hipHostFree(h_stage2);
  h_stage2 = nullptr;
  h_stage2_len = 0;
  cb_alltoallv = nullptr;
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  ev.clear();
  ev_kind.clear();
  ev_used = 0;
  kind = 0;
  world = 1;
  rank = 0;
  force = false;
}

}  // namespace sim3opt
