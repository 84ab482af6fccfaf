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

}  // namespace sim3opt
