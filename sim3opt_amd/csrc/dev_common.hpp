// dev_common.hpp -- device helpers shared by every kernel header (included by engine_impl.hpp inside
// namespace sim3opt, after DevScalars): fixed-order reductions, the Sim3 load, the FP32 pair layout.
#pragma once
// ------------------------------------------------------------------------------------------
// reductions (fixed order => deterministic)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Lanes of ONE wavefront exchanging values through LDS: the hardware executes a wavefront's LDS instructions in
// order, so no instruction is needed -- but the compiler must not move an LDS access across the hand-over (it may
// otherwise split the lanes by a condition and let one side read before the other side wrote: seen, end of round 4),
// and the hand-over must stay convergent.  Wavefront-scope fences + the (instruction-free) wave barrier say that.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double block_sum(double v, double* sh4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

__device__ __forceinline__ double sum_partials(const double* __restrict__ p, int n, double* sh4) {
  // (four loads in flight per thread: with thousands of partials the plain loop was a chain of
  // load-wait-add round trips)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * WG < n; i += 4 * WG) {
    const double v0 = p[i], v1 = p[i + WG], v2 = p[i + 2 * WG], v3 = p[i + 3 * WG];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; i < n; i += WG) a0 += p[i];
  return block_sum((a0 + a1) + (a2 + a3), sh4);
}

__device__ __forceinline__ Sim3 load_sim3(const Sim3* __restrict__ p) {
  Sim3 s;
  const double* d = reinterpret_cast<const double*>(p);
  s.q[0] = d[0]; s.q[1] = d[1]; s.q[2] = d[2]; s.q[3] = d[3];
  s.t[0] = d[4]; s.t[1] = d[5]; s.t[2] = d[6]; s.s = d[7];
  return s;
}

// FP32 copies of the blocks (multigrid matrix passes) are stored as interleaved PAIRS: entry e of
// block k sits at 98 (k / 2) + 2 e + (k mod 2), so that ONE 8-byte load per lane brings the same entry
// of two consecutive blocks -- 392 bytes per wavefront instruction, like an FP64 block, instead of 196.
__host__ __device__ __forceinline__ size_t f32_pair_index(int64_t k, int e) {
  return (size_t)98 * (size_t)(k >> 1) + (size_t)(2 * e) + (size_t)(k & 1);
}
