// devmem.cpp -- see devmem.hpp
#include "devmem.hpp"

#include <cstring>
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
  std::map<int, std::vector<hipStream_t>> streams;                 // idle, per device
  std::map<int, std::vector<hipEvent_t>> events;
  // the device a handle was created on: it is filed under THAT device when it comes back, whatever the
  // calling thread's current device is then (an engine may be destroyed under another current device)
  std::unordered_map<hipStream_t, int> stream_dev;
  std::unordered_map<hipEvent_t, int> event_dev;
  std::map<std::pair<int, size_t>, std::vector<void*>> host_free;  // pinned blocks
  std::unordered_map<void*, std::pair<int, size_t>> host_live;
  size_t host_cached = 0;
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
  for (auto& kv : c.streams)
    for (hipStream_t s : kv.second) { (void)hipStreamDestroy(s); c.stream_dev.erase(s); }
  c.streams.clear();
  for (auto& kv : c.events)
    for (hipEvent_t e : kv.second) { (void)hipEventDestroy(e); c.event_dev.erase(e); }
  c.events.clear();
  for (auto& kv : c.host_free)
    for (void* p : kv.second) (void)hipHostFree(p);
  c.host_free.clear();
  c.host_cached = 0;
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

hipError_t stream_acquire(hipStream_t* s) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  Cache& c = cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    std::vector<hipStream_t>& v = c.streams[dev];
    if (!v.empty()) {
      *s = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  const hipError_t e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(c.mu);
    c.stream_dev[*s] = dev;
  }
  return e;
}

void stream_release(hipStream_t s) {
  if (!s) return;
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.stream_dev.find(s);
  if (it == c.stream_dev.end()) {  // not ours
    (void)hipStreamDestroy(s);
    return;
  }
  std::vector<hipStream_t>& v = c.streams[it->second];
  if (v.size() < 8) v.push_back(s);
  else {
    (void)hipStreamDestroy(s);
    c.stream_dev.erase(it);
  }
}

hipError_t event_acquire(hipEvent_t* e) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  Cache& c = cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    std::vector<hipEvent_t>& v = c.events[dev];
    if (!v.empty()) {
      *e = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  const hipError_t rc = hipEventCreate(e);
  if (rc == hipSuccess) {
    std::lock_guard<std::mutex> lock(c.mu);
    c.event_dev[*e] = dev;
  }
  return rc;
}

void event_release(hipEvent_t e) {
  if (!e) return;
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.event_dev.find(e);
  if (it == c.event_dev.end()) {  // not ours
    (void)hipEventDestroy(e);
    return;
  }
  std::vector<hipEvent_t>& v = c.events[it->second];
  if (v.size() < 4096) v.push_back(e);
  else {
    (void)hipEventDestroy(e);
    c.event_dev.erase(it);
  }
}

hipError_t host_malloc(void** p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t sz = rounded(bytes);
  Cache& c = cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.host_free.find({dev, sz});
    if (it != c.host_free.end() && !it->second.empty()) {
      *p = it->second.back();
      it->second.pop_back();
      c.host_cached -= sz;
      c.host_live[*p] = {dev, sz};
      return hipSuccess;
    }
  }
  const hipError_t e = hipHostMalloc(p, sz);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(c.mu);
    c.host_live[*p] = {dev, sz};
  }
  return e;
}

void host_free(void* p) {
  if (!p) return;
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.host_live.find(p);
  if (it == c.host_live.end()) {
    (void)hipHostFree(p);
    return;
  }
  const std::pair<int, size_t> key = it->second;
  c.host_live.erase(it);
  if (key.second <= ((size_t)8 << 20) && c.host_cached + key.second <= ((size_t)64 << 20)) {
    c.host_free[key].push_back(p);
    c.host_cached += key.second;
  } else {
    (void)hipHostFree(p);
  }
}

hipError_t StagedUploads::put(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if (bytes <= MAX_ITEM) {
    if (!base_) {
      void* p = nullptr;
      if (host_malloc(&p, BLOCK) == hipSuccess) base_ = static_cast<char*>(p);
      else (void)hipGetLastError();
      off_ = 0;
    }
    if (base_ && off_ + bytes <= BLOCK) {
      std::memcpy(base_ + off_, src, bytes);
      const hipError_t e = hipMemcpyAsync(dst, base_ + off_, bytes, hipMemcpyHostToDevice, stream);
      off_ = (off_ + bytes + 63) & ~(size_t)63;
      return e;
    }
  }
  return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

void StagedUploads::release() {
  if (base_) host_free(base_);
  base_ = nullptr;
  off_ = 0;
}

void dev_cache_release() {
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  release_locked(c);
}

}  // namespace sim3opt
