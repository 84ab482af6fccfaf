// devmem.hpp -- device allocations of the engine go through a small cache: g2o allows
// initializeOptimization() again and again (the incremental configuration re-plans after every loop
// closure), and 60 hipFree + 60 hipMalloc calls were most of the 6 ms such a re-initialisation cost.
// Freed blocks of up to 64 MB are kept (1 GB in all, per process) and handed out again for requests
// of the same rounded size; everything larger goes straight to hipMalloc / hipFree.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace sim3opt {

hipError_t dev_malloc(void** p, size_t bytes);
void dev_free(void* p);
void dev_cache_release();  // gives every cached block back to the driver
// live sim3opt_graph / sim3opt_ba handles of the process (delta = +1 / -1; returns the new count):
// the cache is released when the last one goes (sim3opt_release_device_cache does it on request)
int handle_count(int delta);

// The same idea for what else an engine creates and destroys around every re-initialisation (round 3: 2.9 of
// the 4 ms a re-initialisation of KITTI-00 took were a stream, a dozen events, a pinned scalar block and twenty
// synchronous 50-us uploads of a few KB each): idle streams, events and pinned host blocks are kept per device
// and handed out again; dev_cache_release() destroys them too.
hipError_t stream_acquire(hipStream_t* s);   // a non-blocking stream
void stream_release(hipStream_t s);          // (the caller has synchronised it)
hipError_t event_acquire(hipEvent_t* e);
void event_release(hipEvent_t e);
hipError_t host_malloc(void** p, size_t bytes);  // pinned
void host_free(void* p);

// Small host -> device copies of an initialisation go through one pinned staging block and are enqueued on the
// engine's stream (the caller synchronises once, when it has enqueued them all); large ones are copied directly.
class StagedUploads {
 public:
  ~StagedUploads() { release(); }
  // copies bytes from src (pageable) to dst (device); asynchronous on `stream` when it fits the staging block
  hipError_t put(void* dst, const void* src, size_t bytes, hipStream_t stream);
  void release();  // gives the staging block back (after the caller's synchronisation)

 private:
  static constexpr size_t BLOCK = (size_t)4 << 20, MAX_ITEM = (size_t)512 << 10;
  char* base_ = nullptr;
  size_t off_ = 0;
};

}  // namespace sim3opt
