// Micro-benchmark: cost of a counter grid barrier among a SMALL number of workgroups (the regime a
// fused kernel for the multigrid's smallest levels would run in).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int* timeout) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // release
    atomicAdd(counter, 1u);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 2000000) { *timeout = 1; break; }
    }
    __threadfence();  // acquire
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void k_bar(unsigned* counter, int nbar, double* data, int* timeout) {
  const int G = gridDim.x;
  double acc = 0.0;
  for (int b = 0; b < nbar; ++b) {
    // a token amount of work: each workgroup writes a line, reads its neighbour's after the barrier
    if (threadIdx.x == 0) data[16 * blockIdx.x] = (double)(b + 1);
    grid_barrier(counter, (unsigned)(b + 1) * G, timeout);
    acc += data[16 * ((blockIdx.x + 1) % G)];
  }
  if (threadIdx.x == 0) data[16 * blockIdx.x + 1] = acc;
}

int main() {
  unsigned* counter; double* data; int* timeout;
  hipMalloc(&counter, 4); hipMalloc(&data, 8 * 16 * 1024); hipMalloc(&timeout, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nbar = 200;
  for (int G : {8, 16, 32, 64, 128, 176, 256}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(counter, 0, 4); hipMemset(timeout, 0, 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_bar, dim3(G), dim3(512), 0, 0, counter, nbar, data, timeout);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    int to = 0; hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost);
    double acc; hipMemcpy(&acc, data + 1, 8, hipMemcpyDeviceToHost);
    std::printf("G %4d: %.2f us per barrier (timeout %d, check %.0f == %.0f)\n", G, best * 1000.0 / nbar, to, acc,
                nbar * (nbar + 1) / 2.0);
  }
  return 0;
}
