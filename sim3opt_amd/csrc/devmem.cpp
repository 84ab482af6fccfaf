// devmem.cpp -- see devmem.hpp
#include "devmem.hpp"

#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

namespace sim3opt {
namespace {

constexpr size_t MAX_BLOCK = (size_t)64 << 20;
constexpr size_t MAX_TOTAL = (size_t)1 << 30;

struct Cache {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;  // (device, rounded size) -> blocks
  std::unordered_map<void*, std::pair<int, size_t>> live;            // blocks handed out by dev_malloc
  size_t cached_bytes = 0;
};
Cache& cache() {
  static Cache* c = new Cache();  // (never destroyed: the driver may already be gone at exit)
  return *c;
}

// eight steps per octave: at most 12.5 % more than asked for, so that a graph that grew by one edge
// finds the blocks of its predecessor
size_t rounded(size_t bytes) {
  if (bytes < 256) return 256;
  size_t step = 256;
  while ((step << 4) <= bytes) step <<= 1;
  return (bytes + step - 1) / step * step;
}

void release_locked(Cache& c) {
  for (auto& kv : c.free_blocks)
    for (void* p : kv.second) (void)hipFree(p);
  c.free_blocks.clear();
  c.cached_bytes = 0;
}

}  // namespace

hipError_t dev_malloc(void** p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t sz = rounded(bytes);
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  if (sz <= MAX_BLOCK) {
    auto it = c.free_blocks.find({dev, sz});
    if (it != c.free_blocks.end() && !it->second.empty()) {
      *p = it->second.back();
      it->second.pop_back();
      c.cached_bytes -= sz;
      c.live[*p] = {dev, sz};
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, sz);
  if (e != hipSuccess && c.cached_bytes > 0) {  // out of memory with blocks in the cache: give them back, retry
    (void)hipGetLastError();
    release_locked(c);
    e = hipMalloc(p, sz);
  }
  if (e == hipSuccess) c.live[*p] = {dev, sz};
  return e;
}

void dev_free(void* p) {
  if (!p) return;
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.live.find(p);
  if (it == c.live.end()) {  // not ours
    (void)hipFree(p);
    return;
  }
  const std::pair<int, size_t> key = it->second;
  c.live.erase(it);
  if (key.second <= MAX_BLOCK && c.cached_bytes + key.second <= MAX_TOTAL) {
    c.free_blocks[key].push_back(p);
    c.cached_bytes += key.second;
  } else {
    (void)hipFree(p);
  }
}

int handle_count(int delta) {
  static std::mutex mu;
  static int n = 0;
  std::lock_guard<std::mutex> lock(mu);
  n += delta;
  return n;
}

void dev_cache_release() {
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  release_locked(c);
}

}  // namespace sim3opt
