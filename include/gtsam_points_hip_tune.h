/*
 * gtsam_points_hip_tune.h -- measurement entry points of libgtsam_points_hip_tune.so (gp_microbench.hip): micro-benchmarks of the
 * access patterns and VALU instructions the tile kernel is made of, and the known-byte-count stream that calibrates the rocprofv3
 * FETCH_SIZE counter.  Tuning / profiling tools only (scripts/stream_bench.py, scripts/alu_rate.py, bench.py under
 * GP_BENCH_CALIBRATE): NOT part of the drop-in boundary and not linked into libgtsam_points_hip.so.
 */
#ifndef GTSAM_POINTS_HIP_TUNE_H
#define GTSAM_POINTS_HIP_TUNE_H

#include "gtsam_points_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* measurement hook (gp_microbench.hip): mode 0-2 time to just read the 48*n source bytes (strided dwords / float4 / LDS-DMA),
 * 3-12 source + voxel-gather access patterns, 100-115 VALU issue rates; see scripts/stream_bench.py, scripts/alu_rate.py */
int gp_debug_stream_bench(const float* points_dev, const float* covs_dev, int n, int mode, int iters, float* ms);
/* profiling hook: streams 48*n bytes with strided dword loads (calibrates the rocprofv3 FETCH_SIZE scale) */
/* one wave that spins for `microseconds` on `stream` (asynchronous): keeps the device busy across a host-side gap (probe of the idle-queue
 * effect on the tile kernel's duration) */
int gp_debug_spin(double microseconds, gp_stream_t stream);
int gp_debug_calibration_stream(const float* points_dev, const float* covs_dev, int n, int iters, gp_stream_t stream);
/* test hook for the stable radix sort behind the structure builds (gp_sort.hpp; tests/test_sort_gpu.py): argsort of n keys by their low key_bits bits.
 * keys_dev is overwritten; sorted keys -> keys_out_dev, original indices -> vals_out_dev.  Synchronous. */
int gp_debug_sort_pairs(unsigned* keys_dev, int n, int key_bits, unsigned* keys_out_dev, int* vals_out_dev, gp_stream_t stream);
/* the same with a chosen number of ticket classes (gp_sort.hpp: 32 = fast path resting on in-order workgroup start, 1 = the form that needs no such order, negative
 * = test hook: tile 0 raises the fault word); *fault = 1 when a pass gave up a wait (output void). */
int gp_debug_sort_pairs_ex(unsigned* keys_dev, int n, int key_bits, unsigned* keys_out_dev, int* vals_out_dev, int ticket_classes, int* fault, gp_stream_t stream);
/* `workgroups` workgroups of 256 threads spinning for `microseconds` on `stream` (asynchronous): holds the CUs while something else runs on another stream */
int gp_debug_occupy(double microseconds, int workgroups, gp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GTSAM_POINTS_HIP_TUNE_H */
