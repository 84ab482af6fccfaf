/* sim3opt_bench.h -- measurement hooks of libsim3opt: NOT part of the drop-in interface
 * (include/sim3opt.h is what a maintainer of the reference binds).  bench.py and the tuning scripts
 * under scripts/ use them to time single kernels on a graph that is already resident in HBM. */
#ifndef SIM3OPT_BENCH_H
#define SIM3OPT_BENCH_H

#include "sim3opt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* times `reps` back-to-back launches of the block-CSR SpMV kernel; returns mean ms */
int sim3opt_bench_spmv(sim3opt_graph* g, int32_t reps, double* ms_mean);
/* HBM read calibration over the same value array (bench only): mode 0 = 16 B/lane contiguous,
 * 1 = 8 B/lane contiguous, 2 = 8 B/lane on 49 of 64 lanes per 392-B block (the SpMV's shape) */
int sim3opt_bench_stream(sim3opt_graph* g, int32_t mode, int32_t reps, double* ms_mean);
#ifdef SIM3OPT_BENCH_HOOKS
/* Measurement prototype, NOT exported by the product library (only by a SIM3OPT_BENCH_HOOKS build,
 * csrc/engine_proto.hip): the two-phase SpMV over upper-triangle storage (csrc/symm_proto.hpp) on the
 * last linearisation.  out[0] / out[1]: mean ms of phase 1 / phase 2 over
 * `reps` launches, out[2]: max difference to the product SpMV relative to max |q|, out[3]: bytes of its
 * stream.  Single GPU. */
int sim3opt_bench_spmv_symmetric(sim3opt_graph* g, int32_t reps, double out[4]);
/* Measurement prototype (csrc/rowlane_proto.hpp): the cycle's level-0 FP32 passes with a group of 7 lanes per block
 * row against the product kernel.  out[0..1]: ms of the product's residual / smoothing pass, out[2..3]: the
 * prototype's, out[4..5]: the prototype's with four systems sharing the block stream, out[6..7]: max difference of
 * the one-system results to the product's relative to max |q|.  Needs the multigrid hierarchy; single GPU. */
int sim3opt_bench_spmv_rowlane(sim3opt_graph* g, int32_t reps, int32_t rows_per_group, double out[8]);
#endif

#ifdef __cplusplus
}
#endif

#endif /* SIM3OPT_BENCH_H */
