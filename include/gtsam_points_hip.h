/*
 * gtsam_points_hip.h -- C-ABI of libgtsam_points_hip.so
 *
 * MI355X (gfx950) native replacement for the CUDA half of koide3/gtsam_points' VGICP
 * path.  This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference tree, v1.2.1).  The reference-side binding a maintainer would add is shown
 * in INTEGRATION.md; gtsam_points_amd/host/ holds a C++ mirror of the reference classes
 * written against this header.
 *
 * Conventions
 *   - All functions return 0 (GP_OK) on success, non-zero otherwise; gp_last_error()
 *     returns a thread-local message.  (The reference logs CUDA errors to stderr and
 *     continues, cuda/check_error.cu:8-18; the C++ mirror reproduces that on top of the
 *     status codes.)
 *   - Matrices are COLUMN-MAJOR (Eigen default).  A pose is double[16], the 4x4
 *     T_target^-1 * T_source ("delta") -- the double-precision analogue of the
 *     Eigen::Isometry3f the reference uploads (integrated_vgicp_factor_gpu.cpp:136-164).
 *     Double is required for <=1e-5 parity with the CPU IntegratedVGICPFactor.
 *   - Source clouds are caller-owned device arrays in the reference's GPU layout
 *     (types/point_cloud.hpp:114-118): points float[N][3], covs float[N][9] (column-major
 *     3x3), normals float[N][3] (optional).  The library never frees them.
 *   - Covariances: all nine entries are read.  A symmetric matrix (what estimate_covariances
 *     produces after the cast to float) is used as it is; of a non-symmetric one the kernels use
 *     the symmetric part (a_ij + a_ji) / 2, formed in double -- the part the reference CPU
 *     factor's full 3x3 algebra sees to first order in the asymmetry (the HessianFactor keeps
 *     only the upper triangle of H, integrated_matching_cost_factor.cpp:49).
 *   - gp_stream_t is a hipStream_t (the reference's CUstream_st*).  NULL = default stream.
 *   - Handles are thread-compatible (external synchronisation per handle).
 *   - Ownership: handles hold raw pointers, not references.  A factor points at its voxel map and its source arrays, a batch /
 *     multi-batch / factor set at its factors: destroy in the order batch -> factor -> map, and keep the source arrays alive as
 *     long as a factor uses them (the reference holds shared_ptrs; the C++ mirror does the same on top of these handles).
 *     gp_vgicp_factor_destroy / gp_vgicp_batch_destroy do not synchronise a caller-owned stream (the reference's clone() drops it
 *     first): finish the work on it before destroying.
 */
#ifndef GTSAM_POINTS_HIP_H
#define GTSAM_POINTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GP_OK 0
#define GP_ERROR_INVALID_ARGUMENT 1
#define GP_ERROR_HIP 2
#define GP_ERROR_NOT_LOADED 3 /* voxel map / cloud offloaded from the GPU */
#define GP_ERROR_IO 4
#define GP_ERROR_INDETERMINATE 5 /* normal equations not positive definite (IndeterminantLinearSystemException upstream) */

typedef void* gp_stream_t; /* hipStream_t */

/* ---- runtime (replaces cuda/check_error, cuda/cuda_stream, cuda/cuda_memory, cuda_device_*) ---- */

const char* gp_last_error(void);
const char* gp_version(void);
int gp_device_count(int* count);                                /* cuda/cuda_device_names */
int gp_set_device(int device);
int gp_get_device(int* device);
int gp_device_name(int device, char* name, size_t name_len);    /* cuda_device_names() */
int gp_device_synchronize(void);                                /* cuda/cuda_device_sync.cu */
/* Scratch and structure arrays are stream-ordered pool blocks; released ones are parked in a bounded per-thread cache (<= 96 blocks,
 * <= 4 GiB) and re-used by the next build on the same stream.  This returns the calling thread's parked blocks to the pool, and releases the (up to four) side
 * streams + events gp_estimate_covariances keeps per host thread and device: a long-lived pool thread that is done with a device calls it once. */
int gp_trim_device_cache(void);
int gp_stream_create(gp_stream_t* stream);                      /* cuda/cuda_stream.cu (cudaStreamNonBlocking) */
int gp_stream_destroy(gp_stream_t stream);
int gp_stream_synchronize(gp_stream_t stream);
int gp_malloc(void** ptr, size_t bytes);                        /* cuda/cuda_malloc_async.hpp */
int gp_free(void* ptr);
int gp_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, gp_stream_t stream); /* async; caller syncs */
int gp_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, gp_stream_t stream); /* async; caller syncs */
int gp_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, gp_stream_t stream);  /* async; caller syncs */
int gp_memset(void* dst_dev, int value, size_t bytes, gp_stream_t stream);
int gp_host_malloc(void** ptr, size_t bytes);                   /* pinned staging, cuda/cuda_buffer.cu */
int gp_host_free(void* ptr);

/* ---- StreamRoundRobin / TempBufferManager / StreamTempBufferRoundRobin ----
 * cuda/stream_roundrobin.hpp:14-31, cuda/stream_temp_buffer_roundrobin.hpp:19-65 */

typedef struct gp_temp_buffer gp_temp_buffer_t;
typedef struct gp_stream_pool gp_stream_pool_t;

int gp_temp_buffer_create(size_t init_buffer_size, gp_temp_buffer_t** out);  /* TempBufferManager(size) */
int gp_temp_buffer_get(gp_temp_buffer_t* tb, size_t size, void** dev_ptr);   /* get_buffer(): grow-only, x1.2 */
int gp_temp_buffer_clear(gp_temp_buffer_t* tb);                              /* keep the newest buffer */
int gp_temp_buffer_clear_all(gp_temp_buffer_t* tb);
int gp_temp_buffer_destroy(gp_temp_buffer_t* tb);

int gp_stream_pool_create(int num_streams, size_t init_buffer_size, gp_stream_pool_t** out); /* defaults 4, 512 KiB */
int gp_stream_pool_get(gp_stream_pool_t* pool, gp_stream_t* stream, gp_temp_buffer_t** buffer); /* get_stream_buffer() */
int gp_stream_pool_sync_all(gp_stream_pool_t* pool);
int gp_stream_pool_clear(gp_stream_pool_t* pool);
int gp_stream_pool_clear_all(gp_stream_pool_t* pool);
int gp_stream_pool_destroy(gp_stream_pool_t* pool);

/* ---- GaussianVoxelMapGPU : types/gaussian_voxelmap_gpu.hpp:20-114, .cu:25-573 ---- */

/* VoxelMapInfo, gaussian_voxelmap_gpu.hpp:20-25 */
typedef struct gp_voxelmap_info {
  int num_voxels;
  int num_buckets;
  int max_bucket_scan_count;
  float voxel_resolution;
} gp_voxelmap_info;

/* VoxelBucket {Eigen::Vector3i first; int second}, gaussian_voxelmap_gpu.hpp:30-33 (16 B) */
typedef struct gp_voxel_bucket {
  int coord[3];
  int voxel_index; /* < 0 : empty */
} gp_voxel_bucket;

/* raw device views = the reference's public data members (gaussian_voxelmap_gpu.hpp:86-101) */
typedef struct gp_voxelmap_views {
  const gp_voxel_bucket* buckets; /* [num_buckets] */
  const int* num_points;          /* [num_voxels] */
  const float* voxel_means;       /* [num_voxels][3] */
  const float* voxel_covs;        /* [num_voxels][9] column-major */
  const float* voxel_intensities; /* [num_voxels] */
} gp_voxelmap_views;

typedef struct gp_voxelmap gp_voxelmap_t;

/* GaussianVoxelMapGPU(resolution, init_num_buckets=16384, max_bucket_scan_count=10,
 *                     target_points_drop_rate=1e-3, stream), gaussian_voxelmap_gpu.cu:176-198.
 * resolution is double so that voxel coordinates are computed exactly as the CPU map does
 * (fast_floor(x * (1.0/leaf)), gaussian_voxelmap_cpu.cpp:59-61).
 * Deviation: the default (binned) build keeps EVERY voxel -- like the CPU map -- and sizes the reference-visible bucket table by
 * doubling from init_num_buckets until every voxel is inserted within max_bucket_scan_count probes (the sequence is entered at the first size >= 3 x the
 * number of voxels: a fuller table almost surely fails an attempt); target_points_drop_rate is
 * honoured only by the reference-shaped hashed build (gp_voxelmap_set_tuning(GP_TUNE_MAP_BUILD, 1) and the fallback for huge bounding boxes), whose
 * doubling sequence starts at the first size >= N/16, so with a drop rate > 0 num_buckets and the set of dropped points can
 * differ from the reference's. */
int gp_voxelmap_create(double resolution, int init_num_buckets, int max_bucket_scan_count, double target_points_drop_rate, gp_stream_t stream, gp_voxelmap_t** out);
int gp_voxelmap_destroy(gp_voxelmap_t* map);
/* insert(const PointCloud&): one-shot build from device arrays, gaussian_voxelmap_gpu.cu:211-307.
 * intensities_dev may be NULL (voxel intensities = 0, :232-240).  Returns when everything it issued on the map's stream has finished (a polled completion
 * flag behind the last kernel; a stream synchronisation when that takes longer than 0.5 ms).  The hashed kernel family's private line table is built on first use. */
int gp_voxelmap_insert(gp_voxelmap_t* map, const float* points_dev, const float* covs_dev, const float* intensities_dev, int num_points);
int gp_voxelmap_info_get(const gp_voxelmap_t* map, gp_voxelmap_info* info);
double gp_voxelmap_resolution(const gp_voxelmap_t* map);                       /* voxel_resolution() */
int gp_voxelmap_views_get(const gp_voxelmap_t* map, gp_voxelmap_views* views);
/* download_buckets / download_voxel_num_points / _means / _covs / _intensities, gaussian_voxelmap_gpu.cu:537-571.
 * Host buffers sized from gp_voxelmap_info. Any pointer may be NULL. Synchronous. */
int gp_voxelmap_download(const gp_voxelmap_t* map, gp_voxel_bucket* buckets, int* num_points, float* means, float* covs, float* intensities);
/* full-precision voxel statistics (double means[V][3], covs[V][9] col-major) for parity tests */
int gp_voxelmap_download_f64(const gp_voxelmap_t* map, int* coords, int* num_points, double* means, double* covs);
/* rebuild from flat voxel records (the host half of GaussianVoxelMapGPU::load, gaussian_voxelmap_gpu.cu:372-467):
 * coords int[V][3], num_points int[V], means float[V][3], covs6 float[V][6] (xx,xy,xz,yy,yz,zz), intensities float[V] */
int gp_voxelmap_assign(gp_voxelmap_t* map, int num_voxels, const int* coords, const int* num_points, const float* means, const float* covs6, const float* intensities);
/* save_compact(path) / load(path): text header + GaussianVoxelData records, types/gaussian_voxel_data.hpp:11-54,
 * gaussian_voxelmap_gpu.cu:309-370, :372-467; interoperable with GaussianVoxelMapCPU::save_compact/load. */
int gp_voxelmap_save_compact(const gp_voxelmap_t* map, const char* path);
int gp_voxelmap_load(const char* path, gp_stream_t stream, gp_voxelmap_t** out);
/* OffloadableGPU: memory_usage_gpu / loaded_on_gpu / offload_gpu / reload_gpu, gaussian_voxelmap_gpu.cu:469-535 */
size_t gp_voxelmap_memory_usage_gpu(const gp_voxelmap_t* map);
int gp_voxelmap_loaded_on_gpu(const gp_voxelmap_t* map);
/* 1 when the map carries the occupancy-block grid the default VGICP kernel looks voxels up in (bounding box <= 2^24 blocks of
 * 4x4x4 voxels), 0 when only the hashed tables exist (the kernels then use those) */
int gp_voxelmap_has_block_grid(const gp_voxelmap_t* map);
int gp_voxelmap_offload(gp_voxelmap_t* map, gp_stream_t stream);
int gp_voxelmap_reload(gp_voxelmap_t* map, gp_stream_t stream);
/* correspondence lookup of delta * p for every source point -> voxel index or -1
 * (lookup_voxels_kernel, cuda/kernels/lookup_voxels.cuh:19-97); normals_dev!=NULL enables surface validation */
int gp_voxelmap_lookup(const gp_voxelmap_t* map, const float* points_dev, const float* normals_dev, int num_points, const double delta[16], int* voxel_indices_dev, gp_stream_t stream);
/* overlap_gpu(target, source, delta): number of source points that hit a voxel,
 * types/gaussian_voxelmap_gpu_funcs.cu:192-236.  Synchronous. */
int gp_voxelmap_overlap(const gp_voxelmap_t* map, const float* points_dev, int num_points, const double delta[16], int* num_hits, gp_stream_t stream);
/* overlap_gpu(targets, source, Ts_target_source): number of source points that fall in a voxel of ANY target
 * (deltas = [num_targets][16] column-major doubles), types/gaussian_voxelmap_gpu_funcs.cu:265-335.  Synchronous. */
int gp_voxelmap_overlap_multi(const gp_voxelmap_t* const* targets, const double* deltas, int num_targets, const float* points_dev, int num_points, int* num_hits,
                              gp_stream_t stream);
/* overlap_gpu(targets, sources, Ts_target_source) -> one count per (target[i], source[i]) pair, one launch for all pairs,
 * types/gaussian_voxelmap_gpu_funcs.cu:337-404.  Synchronous. */
int gp_voxelmap_overlap_batch(const gp_voxelmap_t* const* targets, const float* const* points_dev, const int* num_points, const double* deltas, int num_pairs,
                              int* num_hits, gp_stream_t stream);

/* ---- merge_frames_gpu : types/gaussian_voxelmap_gpu_funcs.cu:65-152 ----
 * gp_transform_frames: out[begin_i + j] = pose_i * frame_i[j] for all frames in one launch (transform_means_kernel /
 * transform_covs_kernel, :42-62; f64 arithmetic on the f32 inputs).  out_covs_dev / out_intensities_dev may be NULL;
 * a NULL intensities entry yields zeros (:103-107).  Synchronous.
 * gp_merge_frames: the transform followed by the Gaussian voxel-map build at downsample_resolution (:122-123, the
 * reference hard-codes init_num_buckets = total points, scan count 10, drop rate 1e-3); the merged cloud is the returned
 * map's voxel_means / voxel_covs / voxel_intensities views (caller owns the map). */
int gp_transform_frames(const double* poses, const float* const* points_dev, const float* const* covs_dev, const float* const* intensities_dev, const int* num_points,
                        int num_frames, float* out_points_dev, float* out_covs_dev, float* out_intensities_dev, gp_stream_t stream);
int gp_merge_frames(const double* poses, const float* const* points_dev, const float* const* covs_dev, const float* const* intensities_dev, const int* num_points,
                    int num_frames, double downsample_resolution, double target_points_drop_rate, gp_stream_t stream, gp_voxelmap_t** out_map);

/* ---- PointCloudGPU::add_points_gpu / add_normals_gpu / add_covs_gpu : types/point_cloud_gpu.cu:26-62,110-201 ----
 * src_host: num_points vectors (vec3) or column-major matrices (mat3) of dimension src_dim in {3,4}, double or float,
 * exactly as the reference's Eigen::Matrix<T, D, 1> / <T, D, D> arrays lie in memory; dst_dev: float[N][3] / float[N][9].
 * The conversion runs on the device (the reference converts element-wise on the host).  Synchronous. */
int gp_cloud_upload_vec3(const void* src_host, int src_is_double, int src_dim, int num_points, float* dst_dev, gp_stream_t stream);
int gp_cloud_upload_mat3(const void* src_host, int src_is_double, int src_dim, int num_points, float* dst_dev, gp_stream_t stream);

/* ---- LinearizedSystem6 : cuda/kernels/linearized_system.cuh:10-71 ----
 * Same fields, double precision, no Eigen alignment padding; num_inliers is carried as a double so
 * that a stack of records is a homogeneous f64 array (sum-reducible with one RCCL all-reduce).
 * HessianFactor convention (applied by the caller): (H_target, H_target_source, -b_target, H_source, -b_source, error),
 * integrated_vgicp_factor_gpu.cpp:199-213. */
typedef struct gp_linearized6 {
  double num_inliers;
  double error;
  double H_target[36];
  double H_source[36];
  double H_target_source[36];
  double b_target[6];
  double b_source[6];
} gp_linearized6; /* 122 doubles = 976 B */

/* the reference's own float record layout (496 B with Eigen's 16-B alignment), for callers that kept it */
typedef struct gp_linearized6_f32 {
  int num_inliers;
  float error;
  float pad_[2];
  float H_target[36];
  float H_source[36];
  float H_target_source[36];
  float b_target[6];
  float b_source[6];
} gp_linearized6_f32;
void gp_linearized6_to_f32(const gp_linearized6* in, gp_linearized6_f32* out);

/* ---- IntegratedVGICPDerivatives / IntegratedVGICPFactorGPU device half ----
 * factors/integrated_vgicp_derivatives.cuh, integrated_vgicp_derivatives*.cu,
 * factors/integrated_vgicp_factor_gpu.cpp:136-272 */

typedef struct gp_vgicp_factor gp_vgicp_factor_t;

/* IntegratedVGICPDerivatives(target, source, stream, temp_buffer), integrated_vgicp_derivatives.cu:19-47.
 * Borrows the map handle and the three device arrays (normals_dev may be NULL).  stream==NULL -> the factor
 * creates and owns a non-blocking stream (:36-39); temp_buffer==NULL -> it owns its scratch (:41-43).
 * Source points with a NaN / inf coordinate have no correspondence (they count neither as inliers nor towards H, b, the error). */
int gp_vgicp_factor_create(const gp_voxelmap_t* target, const float* points_dev, const float* covs_dev, const float* normals_dev, int num_points, gp_stream_t stream, gp_temp_buffer_t* temp_buffer, gp_vgicp_factor_t** out);
int gp_vgicp_factor_destroy(gp_vgicp_factor_t* f);
int gp_vgicp_factor_set_surface_validation(gp_vgicp_factor_t* f, int enable); /* set_enable_surface_validation */
/* the source cloud's device arrays moved (PointCloudGPU::offload_gpu + reload_gpu, types/point_cloud_gpu.cu:304-370):
 * hand the factor the new pointers; every batch holding the factor rebuilds its table on the next issue */
int gp_vgicp_factor_set_source(gp_vgicp_factor_t* f, const float* points_dev, const float* covs_dev, const float* normals_dev);
/* Packed source mirrors.  The stream kernels read a private 36-B-per-point repack of (points_dev, covs_dev) -- replaces the per-point reads of
 * include/gtsam_points/cuda/kernels/vgicp_derivatives.cuh:36-50 -- built once per cloud at a factor's first table build and shared by every factor
 * on the same (points_dev, covs_dev, num_points).  The borrowed arrays are therefore IMMUTABLE while a factor holds them (the reference holds them through
 * PointCloud::ConstPtr).  An owner that rewrites them in place, or frees them while the address may be handed out again, calls
 * gp_source_mirror_invalidate(ptr) (forget mirrors built from ptr, so that later factors pack afresh) and gp_vgicp_factor_set_source on the factors
 * that keep using the cloud (same or new pointers: the factor drops its mirror and packs again).  gp_source_mirror_bytes: device bytes of all live mirrors. */
int gp_source_mirror_invalidate(const void* dev_ptr);
int64_t gp_source_mirror_bytes(void);
int gp_vgicp_factor_set_inlier_update_thresh(gp_vgicp_factor_t* f, double trans, double angle); /* kept for API parity; every linearise rescans all points */
int gp_vgicp_factor_num_points(const gp_vgicp_factor_t* f);
int gp_vgicp_factor_device(const gp_vgicp_factor_t* f); /* the device the source arrays live on */
gp_stream_t gp_vgicp_factor_stream(const gp_vgicp_factor_t* f);
/* sizes the factor reports through NonlinearFactorGPU (integrated_vgicp_factor_gpu.cpp:136-150):
 * 128 (pose, double[16]) / 976 (gp_linearized6) / 128 / 8 (double error) */
size_t gp_vgicp_linearization_input_size(void);
size_t gp_vgicp_linearization_output_size(void);
size_t gp_vgicp_evaluation_input_size(void);
size_t gp_vgicp_evaluation_output_size(void);
/* issue_linearize(lin_input_cpu, lin_input_gpu, lin_output_gpu), integrated_vgicp_factor_gpu.cpp:229-237 +
 * integrated_vgicp_derivatives_linearize.cu:23-55.  Asynchronous on the factor's stream.  pose_dev is the device
 * copy of the pose (double[16]); out_dev receives one gp_linearized6.  No alignment beyond 8 B is assumed. */
int gp_vgicp_factor_issue_linearize(gp_vgicp_factor_t* f, const double* pose_host, const double* pose_dev, gp_linearized6* out_dev);
/* issue_compute_error(lin cpu, eval cpu, lin gpu, eval gpu, out gpu), integrated_vgicp_factor_gpu.cpp:247-263 +
 * integrated_vgicp_derivatives_compute.cu:23-39: correspondences and fused covariance at pose_lin, residual at pose_eval. */
int gp_vgicp_factor_issue_compute_error(gp_vgicp_factor_t* f, const double* pose_lin_host, const double* pose_eval_host, const double* pose_lin_dev, const double* pose_eval_dev, double* out_dev);
int gp_vgicp_factor_sync(gp_vgicp_factor_t* f);                               /* sync_stream() */
/* synchronous fall-backs (IntegratedVGICPDerivatives::linearize / compute_error, integrated_vgicp_derivatives.cu:80-109) */
int gp_vgicp_factor_linearize(gp_vgicp_factor_t* f, const double pose[16], gp_linearized6* out_host);
int gp_vgicp_factor_compute_error(gp_vgicp_factor_t* f, const double pose_lin[16], const double pose_eval[16], double* out_host);

/* ---- NonlinearFactorSetGPU fast path: one batched launch over a factor table ----
 * replaces the per-factor loop of cuda/nonlinear_factor_set_gpu.cpp:64-218 (F x {reduce, select} launches,
 * 2F stream syncs) by ONE tiled launch over all factors' points for a synchronous rigid-pose call (poses read where the caller left them, the workgroup that
 * stores a factor's last partial row finalizes the factor, records and completion words go straight into host-mapped memory), and by a tiled launch + a finalize
 * launch for the asynchronous entry points (device-resident records). */

typedef struct gp_vgicp_batch gp_vgicp_batch_t;

int gp_vgicp_batch_create(gp_vgicp_factor_t* const* factors, int num_factors, gp_stream_t stream, gp_vgicp_batch_t** out);
int gp_vgicp_batch_destroy(gp_vgicp_batch_t* batch);
int gp_vgicp_batch_size(const gp_vgicp_batch_t* batch);
int64_t gp_vgicp_batch_total_points(const gp_vgicp_batch_t* batch);
int64_t gp_vgicp_batch_algorithmic_bytes(const gp_vgicp_batch_t* batch); /* sum 48 N + 16 buckets + 52 voxels + 560 */
/* what a pass of the batch's built table really requests with perfect reuse of the lookup structures: the source stream as the kernel reads it (36 B per point
 * through the packed mirrors, GP_TUNE_SOURCE_MIRROR, else 48; + 12 with surface validation), each distinct map's block grid (or bucket table) and 64-B records once,
 * pose in, record out.  Reported beside the algorithmic figure, which internal repacking does not change. */
int64_t gp_vgicp_batch_actual_bytes(gp_vgicp_batch_t* batch);
/* asynchronous on the batch stream; poses_host = double[F][16]; out_dev = gp_linearized6[F] in device memory */
int gp_vgicp_batch_issue_linearize(gp_vgicp_batch_t* batch, const double* poses_host, gp_linearized6* out_dev);
int gp_vgicp_batch_issue_compute_error(gp_vgicp_batch_t* batch, const double* poses_lin_host, const double* poses_eval_host, double* out_dev);
int gp_vgicp_batch_sync(gp_vgicp_batch_t* batch);
/* the two asynchronous passes with the pose tables ALREADY in device memory (double[F][16] each, column-major: what the device-side retract of gp_lm_graph_* produces):
 * nothing is staged or copied, the host has nothing to wait for.  Same kernels, same records.  rigid != 0: the caller vouches that every 3x3 block is orthonormal to
 * 1e-9 -- the test the host-pose entry points make themselves to choose between the 29-sum kernel + adjoint expansion and the 92-sum kernel that is exact for any block */
int gp_vgicp_batch_issue_linearize_dev(gp_vgicp_batch_t* batch, const double* poses_dev, int rigid, gp_linearized6* out_dev);
int gp_vgicp_batch_issue_compute_error_dev(gp_vgicp_batch_t* batch, const double* poses_lin_dev, const double* poses_eval_dev, double* out_dev);
/* ... and the error evaluation as a synchronous call that POLLS the finalize kernel's completion words (pinned) instead of synchronising the stream: on return everything
 * queued in front of it on the batch's stream is complete, work queued behind it meanwhile is not waited for.  _begin / _end: the same in two halves (one begin at a time). */
int gp_vgicp_batch_compute_error_dev(gp_vgicp_batch_t* batch, const double* poses_lin_dev, const double* poses_eval_dev, double* out_host);
int gp_vgicp_batch_issue_compute_error_dev_begin(gp_vgicp_batch_t* batch, const double* poses_lin_dev, const double* poses_eval_dev);
int gp_vgicp_batch_compute_error_dev_end(gp_vgicp_batch_t* batch, double* out_host);
int gp_vgicp_batch_stream(const gp_vgicp_batch_t* batch, gp_stream_t* out); /* the stream the batch was created on */
/* synchronous: upload poses, compute, download F records into out_host */
int gp_vgicp_batch_linearize(gp_vgicp_batch_t* batch, const double* poses_host, gp_linearized6* out_host);
/* the same pass without the copy into a caller array: *out_view points at the F records where the kernels stored them (the
 * batch's pinned, host-mapped result buffer), valid until the next call on this batch.  The 976 bytes per factor are then read once,
 * by their consumer (NonlinearFactorSetGPU's store_linearized loop), instead of copied first: 6 us of a 98 us 256-factor pass,
 * 12 us of a 289 us 512-factor pass (profiles/r02_sync_batch_overheads.txt) */
int gp_vgicp_batch_linearize_view(gp_vgicp_batch_t* batch, const double* poses_host, const gp_linearized6** out_view);
int gp_vgicp_batch_compute_error(gp_vgicp_batch_t* batch, const double* poses_lin_host, const double* poses_eval_host, double* out_host);
/* timing hook for bench.py: re-runs only the device work (pose upload excluded) `iters` times on the batch stream
 * between two hipEvents and returns the average milliseconds per pass, and separately the two kernels' times */
int gp_vgicp_batch_time_linearize(gp_vgicp_batch_t* batch, const double* poses_host, int iters, float* ms_total, float* ms_main_kernel, float* ms_finalize_kernel);

/* ---- many-factor batches sharded over the GPUs of one node, driven from ONE process ----
 * The reference has no multi-GPU code; this is the sharded form of NonlinearFactorSetGPU::linearize
 * (cuda/nonlinear_factor_set_gpu.cpp:64-139) that BASELINE.json's north_star asks for: the factor list is partitioned into
 * shards, every shard runs the batched kernels on its own device and stream, every shard writes its records into its rows of a
 * zeroed [F x 122] f64 stack, ONE ncclAllReduce(sum) per device over that stack (RCCL over xGMI; each row has one writer, so the
 * sum is exact), one D2H from the first shard's device.  RCCL is dlopen()ed only when a multi-batch spans several devices. */

/* contiguous partition of a factor list into num_shards ranges minimising the largest sum of weights (weights = source points of
 * each factor; list the factors source-submap-major so that a shard holds whole submaps).  Pure host code: needs no GPU. */
typedef struct gp_shard_plan gp_shard_plan_t;
int gp_shard_plan_create(const int64_t* weights, int num_factors, int num_shards, gp_shard_plan_t** out);
int gp_shard_plan_num_shards(const gp_shard_plan_t* plan);
int gp_shard_plan_range(const gp_shard_plan_t* plan, int shard, int* begin, int* end); /* factors [begin, end) */
int gp_shard_plan_destroy(gp_shard_plan_t* plan);

/* a voxel map replicated onto another device (maps are immutable after insert(): a shard that references a target map of a
 * neighbouring shard gets its own copy; arrays travel device-to-device).  The clone is independent of the original. */
int gp_voxelmap_clone_to_device(const gp_voxelmap_t* map, int device, gp_stream_t stream_on_device, gp_voxelmap_t** out);

typedef struct gp_vgicp_multi_batch gp_vgicp_multi_batch_t;
/* factors[i] must have been created from arrays / a map resident on the device of its shard.
 * shard_of_factor == NULL: one shard per device the factors live on (num_shards ignored).  Otherwise shard_of_factor[i] in
 * [0, num_shards): several shards may share a device (the single-GPU rehearsal of an N-GPU plan).
 * use_rccl: 0 = no collective: every shard's finalize kernel stores its records straight into its rows of one host-pinned, portable stack; 1 = ncclAllReduce(sum)
 * of the zeroed [F x 122] f64 stack (required: error when the shards share a device or librccl.so does not load; one shard on one device is a valid 1-rank
 * communicator); 2 = in-place ncclAllGather when the shards are equal contiguous ranges in rank order (what gp_shard_plan deals for equal weights: half the bytes of
 * the all-reduce, no zeroing), else the all-reduce; -1 = automatic: 2 when the shards sit on distinct devices and librccl.so loads, else 0.
 * gp_vgicp_multi_batch_uses_rccl: 0 no collective, 1 all-reduce, 2 all-gather. */
int gp_vgicp_multi_batch_create(gp_vgicp_factor_t* const* factors, int num_factors, const int* shard_of_factor, int num_shards, int use_rccl,
                                gp_vgicp_multi_batch_t** out);
int gp_vgicp_multi_batch_destroy(gp_vgicp_multi_batch_t* mb);
int gp_vgicp_multi_batch_size(const gp_vgicp_multi_batch_t* mb);
int gp_vgicp_multi_batch_num_shards(const gp_vgicp_multi_batch_t* mb);
int gp_vgicp_multi_batch_uses_rccl(const gp_vgicp_multi_batch_t* mb);
int gp_vgicp_multi_batch_shard_info(const gp_vgicp_multi_batch_t* mb, int shard, int* device, int* num_factors, int64_t* num_points);
/* synchronous; poses_host = double[F][16] and out_host = gp_linearized6[F] in the order of `factors` */
int gp_vgicp_multi_batch_linearize(gp_vgicp_multi_batch_t* mb, const double* poses_host, gp_linearized6* out_host);
int gp_vgicp_multi_batch_compute_error(gp_vgicp_multi_batch_t* mb, const double* poses_lin_host, const double* poses_eval_host, double* out_host);
/* HIP-event times of the last pass: the slowest shard's kernels, and what followed them (all-reduce + D2H, or the host gather) */
int gp_vgicp_multi_batch_last_timing(const gp_vgicp_multi_batch_t* mb, float* ms_compute, float* ms_exchange);

/* ---- the exchange of the one-process-per-GPU form as direct stores over xGMI (gp_peer.hip; no reference counterpart: the reference has no multi-GPU code) ----
 * Every rank needs every rank's records (north_star: "all-reduce of the stacked H blocks").  For small exchanges -- the headline's 8 x 976 B -- a collective library is all
 * latency; here every rank stores its rows straight into a buffer of every peer (mapped through hipIpc handles the caller exchanges once), flags its arrival, waits for the
 * peers' flags and hands the complete [world][row_doubles] stack to the host: one single-workgroup kernel per step and rank.  row_doubles <= 8192, world <= 16.
 *   create   allocates this rank's buffer on the current device and writes its IPC handle (gp_peer_exchange_handle_bytes() bytes) to handle_out
 *   connect  handles = [world][handle bytes] in rank order (the own entry is ignored): maps the peers' buffers
 *   rows     this rank's [world][row_doubles] f64 stack of generation 0 / 1 on the device (two generations alternate; a rank's own rows go to row `rank`)
 *   begin    names the generation (0 / 1) whose stack the NEXT exchange fills; does not advance the step (a caller whose own kernels fail between begin and finish
 *            leaves the exchange where it was)
 *   finish   behind the kernels that wrote the rank's rows, on the same stream: the exchange kernel; host_out_pinned (may be NULL) = pinned [world][row_doubles] f64
 *            (checked: pageable or device memory is GP_ERROR_INVALID_ARGUMENT).  The step's sequence number advances here, once the kernel is launched
 *   check    after the stream's synchronisation: GP_OK, or an error when a peer did not arrive within the kernel's time box.  A rank that gives up poisons its
 *            arrival words at every peer, so all ranks fail in the same step; the exchange is then broken for good (destroy it, fall back to a collective)
 *   set_timeout_ms   the time box a rank waits for its peers (default 2000 ms; a rank skew above it -- a debugger, a long GC pause -- is an error here where a
 *            collective would wait) */
typedef struct gp_peer_exchange gp_peer_exchange_t;
int gp_peer_exchange_handle_bytes(void);
int gp_peer_exchange_create(int world, int rank, int row_doubles, gp_peer_exchange_t** out, void* handle_out);
int gp_peer_exchange_connect(gp_peer_exchange_t* px, const void* handles);
void* gp_peer_exchange_rows(gp_peer_exchange_t* px, int generation);
int gp_peer_exchange_begin(gp_peer_exchange_t* px);
int gp_peer_exchange_set_timeout_ms(gp_peer_exchange_t* px, double ms);
int gp_peer_exchange_finish(gp_peer_exchange_t* px, gp_stream_t stream, double* host_out_pinned);
int gp_peer_exchange_check(const gp_peer_exchange_t* px);
int gp_peer_exchange_destroy(gp_peer_exchange_t* px);

/* ---- exact k-NN, covariance estimation, GICP (BASELINE configs[4]; CPU-only upstream) ----
 * KdTree::knn_search (ann/small_kdtree.hpp:437-474, KnnResult ann/knn_result.hpp:36-117), estimate_covariances
 * (features/covariance_estimation.cpp:18-77), IntegratedGICPFactor (factors/impl/integrated_gicp_factor_impl.hpp:132-296) */

typedef struct gp_point_grid gp_point_grid_t; /* cell-sorted copy of a cloud: the GPU stand-in for the reference's KdTree */
int gp_point_grid_create(const float* points_dev, int num_points, double cell_size, gp_stream_t stream, gp_point_grid_t** out);
/* as above with a GP_TUNE_KNN_STRUCTURE value for THIS structure and, for measurement, a device buffer of 8 uint64 work counters the searches on it
 * add to (or NULL): {shell walks, f32 distance evaluations, f64 distance evaluations, block entries read, occupied cells visited, octant stages} */
int gp_point_grid_create_ex(const float* points_dev, int num_points, double cell_size, int structure, unsigned long long* counters_dev, gp_stream_t stream,
                            gp_point_grid_t** out);
int gp_point_grid_destroy(gp_point_grid_t* grid);
/* exact k nearest neighbours (1 <= k <= 32) of each query within max_sq_dist (strict '<', like KnnResult::push).
 * indices_dev int[nq][k] (-1 padded), sq_dists_dev double[nq][k] (may be NULL), num_found_dev int[nq] (may be NULL). Asynchronous. */
int gp_knn_search(const gp_point_grid_t* grid, const float* queries_dev, int num_queries, int k, double max_sq_dist, int* indices_dev, double* sq_dists_dev,
                  int* num_found_dev, gp_stream_t stream);
/* estimate_covariances(points, n, k): k-NN incl. the query -> sample covariance -> V diag(1e-3,1,1) V^-1; fewer than k -> identity.
 * covs_dev float[n][9] column-major; cell_size <= 0 picks 0.25 m; *num_short = points with < k neighbours. Synchronous. */
int gp_estimate_covariances(const float* points_dev, int num_points, int k, double cell_size, float* covs_dev, int* num_short, gp_stream_t stream);
/* as above with a GP_TUNE_KNN_STRUCTURE value and (measurement) a device buffer of 8 work counters or NULL, see gp_point_grid_create_ex */
int gp_estimate_covariances_ex(const float* points_dev, int num_points, int k, double cell_size, float* covs_dev, int* num_short, int structure,
                               unsigned long long* counters_dev, gp_stream_t stream);

typedef struct gp_gicp_factor gp_gicp_factor_t;
/* IntegratedGICPFactor(target, source) with its target 1-NN structure; max_correspondence_distance_sq defaults to 1.0 upstream (:30) */
int gp_gicp_factor_create(const float* target_points_dev, const float* target_covs_dev, int num_target, const float* points_dev, const float* covs_dev, int num_points,
                          double max_correspondence_distance_sq, gp_stream_t stream, gp_gicp_factor_t** out);
int gp_gicp_factor_create_ex(const float* target_points_dev, const float* target_covs_dev, int num_target, const float* points_dev, const float* covs_dev, int num_points,
                             double max_correspondence_distance_sq, int structure, unsigned long long* counters_dev, gp_stream_t stream, gp_gicp_factor_t** out);
int gp_gicp_factor_destroy(gp_gicp_factor_t* f);
int gp_gicp_factor_linearize(gp_gicp_factor_t* f, const double pose[16], gp_linearized6* out_host);           /* update_correspondences + evaluate */
/* evaluate(delta_eval) on the correspondences and Mahalanobis matrices of pose_lin.  The correspondences of the last linearise (or error
 * evaluation) are kept on the device: when pose_lin is bit for bit the pose they were computed at they are re-used -- the reference's
 * error() likewise evaluates on the stored correspondences (impl/integrated_gicp_factor_impl.hpp:183-185) -- otherwise the search runs
 * again at pose_lin.  gp_gicp_factor_linearize always searches, as update_correspondences does with its default (zero) tolerances. */
int gp_gicp_factor_compute_error(gp_gicp_factor_t* f, const double pose_lin[16], const double pose_eval[16], double* out_host);

/* ---- the step after the path: damped normal equations assembled and solved on the device ----
 * DenseLinearSystemBuilder (optimizers/linear_system_builder.cpp:39-48): A = sum of the Hessian blocks scattered by key,
 *   b = sum of g (= -b_target / -b_source, integrated_matching_cost_factor.cpp:49), c = sum of the errors;
 * buildDampedSystem (optimizers/levenberg_marquardt_ext.cpp:146-161): A + lambda I, or A + lambda clamp(diag A, min, max);
 * DenseLinearSolver::solve(A, b) (optimizers/linear_solver.hpp:18-22): A x = b, here by a blocked LL^T in f64.
 * Variables are 6-dof poses in num_slots slots; factor_slots = [num_factors][2] = (target slot, source slot), a negative
 * slot = a pose that is not a variable (fixed / unary factor).  records_dev = the stacked [num_factors] gp_linearized6
 * records exactly as gp_vgicp_batch_issue_linearize (or the multi-GPU all-reduce) leaves them in HBM. */
typedef struct gp_dense_system gp_dense_system_t;
int gp_dense_system_create(int num_slots, const int* factor_slots, int num_factors, gp_stream_t stream, gp_dense_system_t** out);
int gp_dense_system_destroy(gp_dense_system_t* sys);
int gp_dense_system_size(const gp_dense_system_t* sys); /* 6 * num_slots */
/* prior_diag_host: optional [6 * num_slots] extra diagonal (prior factors), may be NULL */
int gp_dense_system_build(gp_dense_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                          const double* prior_diag_host);
/* A as a full symmetric column-major [n][n], b [n], c; any pointer may be NULL.  Synchronous. */
int gp_dense_system_download(const gp_dense_system_t* sys, double* A_host, double* b_host, double* c_host);
/* x (host and / or device copy, either may be NULL); consumes the built system.  GP_ERROR_INDETERMINATE if not positive definite. */
int gp_dense_system_solve(gp_dense_system_t* sys, double* x_host, double* x_dev);
/* One damped step of the optimizer's inner loop (LevenbergMarquardtOptimizerExt::tryLambda: buildDampedSystem + solve, optimizers/levenberg_marquardt_ext.cpp:146-161, 200-220)
 * as ONE call with ONE synchronisation (the dense form with a prior_diag_host: two): build + download(b, c) + solve, bit-identical to the three calls.  x_host [n], b_host [n] (the undamped gradient side the optimizer's
 * model-fidelity test needs), c_host [1]; any may be NULL.  GP_ERROR_INDETERMINATE if not positive definite: b_host / c_host are valid, x_host is not written. */
int gp_dense_system_step(gp_dense_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                         const double* prior_diag_host, double* x_host, double* b_host, double* c_host);

/* gp_dense_system_step in two halves: issue queues the step's kernels on the system's stream and returns; finish waits for the stream and hands over x / b / c
 * (GP_ERROR_INDETERMINATE as the one call).  Work queued on the same stream between the two -- a consumer of the solution where the step leaves it ON THE DEVICE
 * (gp_dense_system_device_solution: x [n] in slot order, the status word: != 0 = indeterminate, x is not to be used) -- shares the step's one wait (gp_lm_graph_try_lambda). */
int gp_dense_system_issue_step(gp_dense_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                               const double* prior_diag_host);
int gp_dense_system_finish_step(gp_dense_system_t* sys, double* x_host, double* b_host, double* c_host);
int gp_dense_system_device_solution(gp_dense_system_t* sys, const double** x_dev, const int** status_dev);
/* a system of ONE pose (a scan registered onto a map) runs its step -- assembly, error sum, damping, 6 x 6 Cholesky, both substitutions, hand-over -- as ONE launch of one
 * 64-thread workgroup, the multi-launch path's arithmetic operation for operation (bit-identical); 0 selects the multi-launch path (tests, A/B); returns what the next
 * step runs (1 / 0).  A step with a prior_diag_host takes the multi-launch path. */
int gp_dense_system_set_one_launch(gp_dense_system_t* sys, int enable);
/* finish_step for a caller that has SEEN the stream pass the step (a completion word of work it queued behind the step): no wait of its own */
int gp_dense_system_collect_step(gp_dense_system_t* sys, double* x_host, double* b_host, double* c_host);

/* ---- the same step, block-sparse: SparseLinearSystemBuilder<6> + SparseLinearSolver ----
 * SparseLinearSystemBuilder<BLOCK_SIZE> (include/gtsam_points/optimizers/linear_system_builder.hpp:41-72): A as a lower-triangular
 *   block-sparse matrix in a given ordering, b, c;  SparseLinearSolver::solve(A, b) (optimizers/linear_solver.hpp:24-29), called from
 *   levenberg_marquardt_ext.cpp:200-220.  Here: 6x6 blocks in a block-column structure that already holds the fill of the factor,
 *   left-looking block LL^T in f64 scheduled over the elimination tree (independent subtrees = one workgroup each, then the
 *   separator columns), forward substitution fused, four launches per solve; every block is a gather in a fixed order (deterministic).
 * ordering: 0 = natural (the slot order is the elimination order: "the ordering from the factor key list"),
 *           1 = nested dissection by BFS bisection of the pose graph (shallow elimination tree: chains become ~log2(P) levels),
 *           2 = minimum degree by multiple elimination (the fill of a COLAMD-class ordering, which is what GTSAM gives the reference's solves:
 *               optimizers/levenberg_marquardt_ext.cpp:200-220); 3 = the same with a slack of one on the degree (more nodes per round: a bushier tree),
 *           4 = automatic: 1 and 3 are both tried, the schedule with the shorter critical path (then the smaller factor) is kept.
 * The numeric phase runs one launch per LEVEL of the schedule: the independent subtrees, then the chains of separator columns level by level.
 * Same slot / record conventions as gp_dense_system_*; no limit on num_slots. */
typedef struct gp_sparse_system gp_sparse_system_t;
int gp_sparse_system_create(int num_slots, const int* factor_slots, int num_factors, int ordering, gp_stream_t stream, gp_sparse_system_t** out);
int gp_sparse_system_destroy(gp_sparse_system_t* sys);
int gp_sparse_system_size(const gp_sparse_system_t* sys); /* 6 * num_slots */
/* structure: blocks of A (lower triangle incl. diagonal), blocks of L (incl. fill), 6x6 block products per factorisation,
 * independent subtrees, columns in the top part; any pointer may be NULL */
int gp_sparse_system_info(const gp_sparse_system_t* sys, int64_t* nnz_a_blocks, int64_t* nnz_l_blocks, int64_t* block_products, int* num_subtrees, int* top_columns);
int gp_sparse_system_build(gp_sparse_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                           const double* prior_diag_host);
/* for checkers: A expanded to a full symmetric column-major [n][n] in slot order, b [n], c; any pointer may be NULL.  Synchronous. */
int gp_sparse_system_download(const gp_sparse_system_t* sys, double* A_host, double* b_host, double* c_host);
/* x in slot order (host and / or device copy, either may be NULL); consumes the built system.  GP_ERROR_INDETERMINATE if not positive definite. */
int gp_sparse_system_solve(gp_sparse_system_t* sys, double* x_host, double* x_dev);
/* gp_dense_system_step's block-sparse form, one wait.  The assembly kernel applies the damping and hands b, c to the host.  A system whose factor and index lists fit the
 * LDS of one compute unit (<= 128 poses, <= ~400 blocks of L: BASELINE configs[2]'s 64-pose graph does) is then factored and solved -- all levels, both substitutions, x
 * and status to the host -- by ONE launch of one 512-thread workgroup with every operand in LDS (sparse_small_step_kernel; a team of waves per work list that meets
 * through LDS words where a level has at most four lists -- no workgroup barrier inside a list --, else a lone wave per list / lock-step teams): two launches per step;
 * larger systems take 2 + 2 x levels launches.  The forms are bit-identical; gp_sparse_system_set_one_launch(sys, 0) selects the multi-launch form for a qualifying
 * system, 2 the one-launch step's first form (every list a team of waves in lock step), 3 its second (a lone wave per list throughout), 1 the default; returns what the
 * next step runs (0 / 2 / 3 / 1). */
int gp_sparse_system_step(gp_sparse_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                          const double* prior_diag_host, double* x_host, double* b_host, double* c_host);
int gp_sparse_system_set_one_launch(gp_sparse_system_t* sys, int enable);
/* gp_sparse_system_step in two halves, as gp_dense_system_issue_step / _finish_step / _device_solution (prior_diag_host is copied by issue: the caller's array is free on return) */
int gp_sparse_system_issue_step(gp_sparse_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                                const double* prior_diag_host);
int gp_sparse_system_finish_step(gp_sparse_system_t* sys, double* x_host, double* b_host, double* c_host);
int gp_sparse_system_device_solution(gp_sparse_system_t* sys, const double** x_slots_dev, const int** status_dev);
int gp_sparse_system_collect_step(gp_sparse_system_t* sys, double* x_host, double* b_host, double* c_host);
/* measurement hook: thread 0 of sparse_small_step_kernel stamps its phases (s_memtime) into dev_buffer (64 uint64: [0] start, [1] lists and system in LDS, [2] factored,
 * [3] substituted, [4] end, [8 + 4 r + 0..3] round r < 14 of the first level: start / gathered / diagonal block done / blocks below done); NULL = off */
int gp_debug_sparse_step_trace(gp_sparse_system_t* sys, unsigned long long* dev_buffer);
/* ---- one Levenberg-Marquardt trial without the host in the middle: the values live beside the records (gp_lm.hip) ----
 * The reference's optimizer (optimizers/levenberg_marquardt_ext.cpp) hands `values` to its GPU factor set twice per trial -- linearization_hook_->linearize(values)
 * (iterate(), :352-392 -> cuda/nonlinear_factor_set_gpu.cpp:64-101) and linearization_hook_->error(newValues) (tryLambda(), :245 -> :103-139) -- and retracts on the host
 * between them (:239).  A gp_lm_graph keeps the N poses in device memory: batch member i is the pairwise factor between poses pose_pairs[2 i] (target) and
 * pose_pairs[2 i + 1] (source) and is evaluated at target^-1 source (integrated_matching_cost_factor.cpp:28-31); pose_fixed[i] != 0 holds pose i (NULL = none held:
 * the gauge is then the caller's problem, the step reports GP_ERROR_INDETERMINATE); the free poses take the variable slots 0, 1, ... in pose order.  The graph builds
 * its own damped system on the batch's stream (one free pose: the dense 6 x 6 step; else block-sparse in `ordering`, gp_sparse_system_create) and does NOT own the batch.
 *   set_values   values_host = double[N][16], column-major 4x4 (all orthonormal to 1e-9: the rigid kernels serve the graph from then on; else the general ones);  get_values: the current values (after accept: the accepted trial's, as the device computed them)
 *   linearize    asynchronous: the batch's linearise at the current values' relative poses -> records in HBM
 *   try_lambda   damped step + retract (Pose3::retract: T Expmap(xi), xi = (omega, v) = the step's six entries of the pose's slot) + the batch's error evaluation on the
 *                linearisation's correspondences at the trial values: queued back to back, ONE wait (a poll of the evaluation's completion words).  x_host [6 slots], b_host [6 slots], c_host (the cost at the
 *                linearisation point), new_error (the cost at the trial values), new_values_host [N][16]; any may be NULL.  GP_ERROR_INDETERMINATE: b / c valid, no trial.
 *   accept       the last successful trial's values become the current ones (a swap: their relative poses are already in place); linearize again before the next trial
 *   optimize     the reference's loop over the three: GTSAM's LevenbergMarquardtParams defaults in gp_lm_params_default; lambda I damping only (diagonalDamping = false) */
typedef struct gp_lm_graph gp_lm_graph_t;
typedef struct gp_lm_params {
  double lambda_initial, lambda_factor, lambda_upper_bound, lambda_lower_bound; /* 1e-5, 10, 1e5, 0 */
  double relative_error_tol, absolute_error_tol, min_model_fidelity;            /* 1e-5, 1e-5, 1e-3 */
  double min_diagonal, max_diagonal;                                            /* 1e-6, 1e32 (diagonal damping only) */
  int max_iterations, diagonal_damping;                                         /* 100, 0 */
} gp_lm_params;
typedef struct gp_lm_summary {
  int iterations, inner_iterations; /* linearisations, trials */
  int gave_up;                      /* lambda reached lambda_upper_bound */
  int reserved_;
  double final_error, final_lambda;
} gp_lm_summary;
void gp_lm_params_default(gp_lm_params* params);
int gp_lm_graph_create(gp_vgicp_batch_t* batch, const int* pose_pairs, int num_poses, const unsigned char* pose_fixed, int ordering, gp_lm_graph_t** out);
int gp_lm_graph_destroy(gp_lm_graph_t* graph);
int gp_lm_graph_num_variables(const gp_lm_graph_t* graph); /* 6 x free poses */
int gp_lm_graph_set_values(gp_lm_graph_t* graph, const double* values_host);
int gp_lm_graph_get_values(gp_lm_graph_t* graph, double* values_host);
int gp_lm_graph_linearize(gp_lm_graph_t* graph);
int gp_lm_graph_try_lambda(gp_lm_graph_t* graph, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal, double* x_host, double* b_host, double* c_host,
                           double* new_error, double* new_values_host);
int gp_lm_graph_accept(gp_lm_graph_t* graph);
/* speculation (on by default): behind a trial's error evaluation the linearise at the TRIAL values is queued into a second record buffer, so that an accepted step finds
 * its linearisation already running while the host decides (a rejected one wastes that launch); 0 = off.  Same bits either way.  Returns the previous setting. */
int gp_lm_graph_set_speculation(gp_lm_graph_t* graph, int enable);
/* the graph's own damped system: gp_sparse_system_set_one_launch / gp_dense_system_set_one_launch passed through (0: the multi-launch step, the retract as a kernel of
 * its own behind it; default 1: the one-launch step where the system qualifies, the retract as its epilogue).  Same bits. */
int gp_lm_graph_set_one_launch(gp_lm_graph_t* graph, int enable);
int gp_lm_graph_optimize(gp_lm_graph_t* graph, const gp_lm_params* params, gp_lm_summary* summary);
/* for checkers: the records of the last linearise and the relative poses of the current values, where they lie in device memory (valid until the graph is destroyed;
 * contents as of the work queued so far on the batch's stream) */
int gp_lm_graph_records(gp_lm_graph_t* graph, const gp_linearized6** records_dev, const double** relative_poses_dev);

/* the symbolic phase alone (pure host code, no device needed): elimination order perm[k] = slot eliminated k-th, elimination tree
 * parent[k] (-1 = root), block counts and the schedule; any output pointer may be NULL */
int gp_sparse_symbolic(int num_slots, const int* factor_slots, int num_factors, int ordering, int* perm_out, int* parent_out, int64_t* nnz_a_blocks, int64_t* nnz_l_blocks,
                       int* num_subtrees, int* top_columns);
/* the schedule of the numeric phase (pure host code): launch levels (level 0 = the independent subtrees, then the chains of separator columns level by
 * level, one workgroup per chain), the critical path in columns (sum over the levels of the longest work list) and the number of work lists */
int gp_sparse_symbolic_schedule(int num_slots, const int* factor_slots, int num_factors, int ordering, int* num_levels, int* critical_columns, int* num_lists);
/* ... and its work lists one by one (pure host code; arrays of `capacity` >= num_lists ints): the level a list runs in, its columns, the 6x6 block products its columns
 * gather in all and the most a single column gathers */
int gp_debug_sparse_work_lists(int num_slots, const int* factor_slots, int num_factors, int ordering, int capacity, int* level, int* columns, int* products, int* max_column_products);

/* ---- per-handle tuning (not part of the reference API) --------------------------------------------------------------------------
 * Every knob below belongs to ONE batch / factor / map / search structure; the library keeps no process-global switches, so two
 * handles driven from two threads never see each other's settings (thread-compatible per handle, re-entrant across
 * handles; tests/test_vgicp_gpu.py::test_two_threads_two_batches).  The library reads NO environment variable (no getenv in csrc/): nothing that is
 * computed, launched or printed depends on the environment (the A/B variables of rounds 2-3 and GP_KNN_DEBUG are gone; tests/test_capi_cpu.py greps for it).
 *
 * GP_TUNE_KERNEL selects the tile-kernel family of a VGICP batch.  M = (C_B + R C_A R^T)^-1, the transform and the residual are f64
 * in every family; "f32 outer products" computes what follows the inverse in f32 (measured parity vs the CPU factor <= 1e-7
 * relative, gate 1e-5).  A family that does not apply to a batch falls back (GP_TUNE_EFFECTIVE_KERNEL reads what a built table runs):
 *   GP_KERNEL_REFERENCE  reference-shaped kernel (reference bucket table, 92 explicit sums for non-orthonormal poses): cross-check
 *   GP_KERNEL_HASHED     pipeline kernel over the hashed line table, f32 outer products: maps without a block grid
 *   GP_KERNEL_GRID_F64   pipeline kernel over the occupancy-block grid, f64 throughout
 *   GP_KERNEL_LOOKAHEAD  round-2 pipeline kernel (block grid, f32 outer products, look-ahead lookup): maps with >= 2^26 voxels
 *   GP_KERNEL_STREAM     third generation (csrc/gp_vgicp_stream.hpp): per-wave chunk streams, balanced single-factor launches,
 *                        surface validation inside the ring.  Default.
 * (The A/B switches of rounds 2-3 -- poses by copy engine, finalize parts / width / host expansion, the fused GICP kernel -- were measured, decided and removed:
 *  profiles/r02_*, r03_*, DESIGN.md.) */
enum {
  GP_KERNEL_REFERENCE = 0,
  GP_KERNEL_HASHED = 2,
  GP_KERNEL_GRID_F64 = 3,
  GP_KERNEL_LOOKAHEAD = 8,
  GP_KERNEL_STREAM = 12
};
enum {
  GP_TUNE_KERNEL = 0,           /* GP_KERNEL_* */
  GP_TUNE_SOURCE_POLICY = 1,    /* cache policy of the source stream: 0 per batch (non-temporal iff no two factors share a source cloud), 1 default, 2 non-temporal */
  GP_TUNE_XCD_CHUNK = 2,        /* workgroup -> tile map: 0 = every XCD walks a contiguous eighth of the tile list, c > 0 = runs of c tiles dealt round robin */
  GP_TUNE_STAGGER = 3,          /* round-2 kernels: the odd wave slots of every SIMD start `value` x 512 clocks late (0 = off) */
  GP_TUNE_TILE_INTERLEAVE = 4,  /* 1 = consecutive factors that share a source cloud take turns tile by tile, 0 (default) = factor-major */
  GP_TUNE_BALANCE = 5,          /* stream kernel, one large factor: how much more a dispatch round of workgroups takes than the next one, in 1/1000 of
                                   the mean share (0 = flat split; -1 = automatic, the default: 250 for small shares down to 100 for large ones; a compute unit issues from
                                   its oldest waves first, csrc/gp_vgicp_shared.hpp) */
  GP_TUNE_EFFECTIVE_KERNEL = 6, /* read-only: the family the batch's current table runs (-1 before the first pass) */
  GP_TUNE_XCD_WEIGHT_0 = 8,     /* .. + 7: stream kernel, one large factor: share of XCD x in 1/1000 of the mean share (500..1500); setting any of the eight
                                   replaces the library's measured table (the others then count as 1000) */
  GP_TUNE_FUSED_FINALIZE = 17,  /* synchronous rigid-pose linearise / error evaluation of the stream family: 1 (default) = ONE launch -- a large single factor: the tile
                                   workgroup whose arrival completes an eighth of the row list sums that eighth and hands the sums to the host; a batch or a small
                                   factor: the workgroup that stores a factor's last row sums, expands and delivers the factor's record.  0 = tile kernel, then finalize
                                   kernel.  The records of the two forms are bit-identical (the same functions in the same order, csrc/gp_vgicp_finalize.hpp).
                                   1.5-1.8 us off a 1 M-point step, 10-20 % off a 256- / 512-factor call (profiles/r03_fused_finalize.jsonl, r03_fused_by_factor.txt) */
  GP_TUNE_TILE_CHUNKS = 18,     /* stream family, fixed-tile launches (batches, small single factors): 64-point chunks per wave of a tile (a tile = 256 x value points);
                                   0 (default) = 8 when the batch has >= 2048 tiles of 2048 points, else the largest of 4 / 2 / 1 that still gives >= 768 tiles */
  GP_TUNE_TEST_ARRIVAL_SKEW = 20, /* test hook (batches only): puts the host's count of arrival counter 0 `value` ahead of the device's, as a lost launch would; the next fused
                                   step must notice that its completion words do not arrive, reset the counters and finish through the finalize kernel */
  GP_TUNE_MAX_WORKGROUPS = 19,  /* stream family, one large factor: workgroups of the planned launch, 8 .. 1024 (default 1024 = one resident round) */
  GP_TUNE_SOURCE_MIRROR = 21,   /* stream family: 1 (default) = the kernels stream the sources' packed private mirrors (36 B per point: 12 B point + the six floats of the
                                   symmetric covariance, chunk-major; built once per cloud at the first table build, shared by all factors on the cloud, only from
                                   covariances that are symmetric to the last bit -- records are bit-identical to 0 = the caller's arrays (12 + 36 B per point) */
  GP_TUNE_EFFECTIVE_MIRROR = 22,/* read-only: 1 when the batch's current table streams the packed mirrors (-1 before the first pass) */
  GP_TUNE_EXPERIMENT = 23,      /* measurement only: instantiations of the stream kernel built for an A/B of round 5 (1 = block-grid warm-up, 2 = f32 covariance rotation,
                                   which BREAKS the parity contract); applies to a synchronous planned single-factor linearise, 0 (default) = the product kernel */
  GP_TUNE_TIMING = 7,           /* measurement: 1 = gp_vgicp_batch_linearize brackets its two kernels with HIP events (gp_vgicp_batch_last_kernel_ms) */
  GP_TUNE_MAP_BUILD = 16,       /* gp_voxelmap: 1 = reference-shaped hashed build (atomicCAS claims + atomic sums; also the fallback of clouds whose
                                   bounding box is too large for the block grid), 0 = binned deterministic build (default) */
  GP_TUNE_BUCKET_LOAD = 24,     /* gp_voxelmap (binned build): load factor in per cent (5 .. 90, default 33) at which the reference-visible bucket table enters the reference's doubling
                                   sequence (gaussian_voxelmap_gpu.cu:269-291).  At 50 .. 67 % some probe chain among 10^5 voxels exceeds max_bucket_scan_count almost surely and the failed
                                   attempt costs a fill + an insertion pass + a wait; 33 % means up to twice the entries (16 B each) of a table sized at 67 % */
  GP_TUNE_KNN_STRUCTURE = 32    /* search structures (gp_point_grid, gp_estimate_covariances_ex, gp_gicp_factor): 0 = binned structure, per-lane search
                                   (default); 1 = hashed multi-level grid (also the fallback of clouds whose bounding box is too large for the block
                                   grid); 3 = as 0 with the row-tiled covariance pass in front of the per-lane search (exact, measured slower:
                                   DESIGN.md section 4.8); 4 = as 0 with a second binned level between the cells and the superblocks; 6 = as 0 with the covariance queries in plain
                                   cell-sorted order (round 3) instead of heavy-first (own-cell population < k first: round 4), for the A/B; 7 = as 0 without round 5's
                                   cooperative pass (covariance_far_kernel: sparse neighbourhoods searched by one wave per query), i.e. round 4's search, for the A/B; >= 16: staging experiment of round 4,
                                   16 | fine shells << 4 | shells of blocks << 8 (measured: no effect, profiles/r04_c5_staging.jsonl) */
};
int gp_vgicp_batch_set_tuning(gp_vgicp_batch_t* batch, int key, int value);
int gp_vgicp_batch_get_tuning(const gp_vgicp_batch_t* batch, int key, int* value);
/* with GP_TUNE_TIMING = 1: the durations of the tile kernel and of the finalize kernel of the last synchronous gp_vgicp_batch_linearize[_view],
 * i.e. of the kernels as they run INSIDE a step (behind the idle queue the host leaves between two passes), HIP events on the batch's stream */
int gp_vgicp_batch_last_kernel_ms(const gp_vgicp_batch_t* batch, float* tile_ms, float* finalize_ms);
/* fused synchronous single-factor linearise steps (GP_TUNE_FUSED_FINALIZE = 1) time themselves on the device's 100 MHz constant clock: the first workgroup of
 * every XCD stores its start, every part's finalizer when the part's last partial row was in and when its sums left for the host.  Since the last reset:
 * number of such steps, mean duration of the streaming part (first start .. last row in: what the roofline fraction is quoted on, as the step ran it) and
 * of the whole kernel's work (.. last sums out), in microseconds.  No events, no profiler, nothing added to the step but three 8-byte stores per part. */
int gp_vgicp_batch_device_times(gp_vgicp_batch_t* batch, int reset, double* steps, double* stream_us_mean, double* kernel_us_mean);
/* the per-factor entry points (gp_vgicp_factor_linearize, ..._issue_*) run a batch of one: this is its tuning */
int gp_vgicp_factor_set_tuning(gp_vgicp_factor_t* factor, int key, int value);
int gp_voxelmap_set_tuning(gp_voxelmap_t* map, int key, int value);
/* host-side check hook (runs without a device): the 29 target-side sums of a rigid pass (ACC layout of csrc/gp_device.hpp: count, error,
 * M[6], K[9], TL[6], q x Mr [3], Mr [3]) and the pose delta (column-major 4x4) -> the complete record, i.e. H_t from the sums and
 * H_s = Ad^T H_t Ad, H_ts = -H_t Ad, b_s = -Ad^T b_t (integrated_vgicp_factor_gpu.cpp:199-213 consumes them).  This is the expansion the
 * synchronous single-factor call runs on the host on the added sums of its finalize parts. */
int gp_debug_expand_rigid(const double sums[32], const double pose[16], gp_linearized6* out);
/* host-side check hook (runs without a device): the tiles (first point, number of points) a planned single-factor launch of the stream kernel deals
 * to its workgroups for a factor of n points and a GP_TUNE_BALANCE value, in tile-list order (XCD-major); *num_tiles = workgroups of the launch */
int gp_debug_stream_plan(int n, int skew_permille, const int* xcd_weights_permille /* [8] or NULL = the library's table */, int capacity, int* begin, int* count,
                         int* num_tiles);
/* host-side check hook (runs without a device): for an explicit shard assignment, *rows_per_shard = rows every rank contributes to the in-place ncclAllGather of a
 * multi-device pass (0: the plan does not allow it -- unequal or non-contiguous shards -- and the pass all-reduces), and send_offset_doubles[k] = where shard k's send
 * buffer starts inside the [F x width] stack, in doubles.  Runs the functions gp_vgicp_multi_batch_* itself uses. */
int gp_debug_multi_gather_plan(const int* shard_of_factor, int num_factors, int num_shards, int width, int64_t* rows_per_shard, int64_t* send_offset_doubles);
/* gp_estimate_covariances runs its second launch on a low-priority side stream; two streams overlap only when their hardware queues sit on different dispatch pipes, so
 * the library probes (once per host thread and device, beside the first caller stream) up to four candidate streams and keeps the one whose queue does not wait for the caller's grid
 * (gp_knn.hip, SideStream).  This returns the measured delays in microseconds (< 0 = not probed) and the index of the stream in use beside `caller`. */
int gp_debug_side_stream_probe(gp_stream_t caller, float delays_us[4], int* chosen);
/* test hooks for the structure builds' sort fallback (gp_sort.hpp / gp_binning.hip; thread-local, no device state): the next `count` builds of this thread (voxel-map
 * insert, k-NN structure) see their first radix sort report "a tile waited for a workgroup that was never started" and must rebuild through the one-class sort;
 * count < 0: the next |count| builds, and the faulted sort also leaves garbage keys (far outside every cell range) in part of its output, as a really expired wait does.
 * gp_debug_sort_fallbacks = how many builds of this thread did so far. */
int gp_debug_inject_sort_fault(int count);
int gp_debug_sort_fallbacks(void);
/* timeline hook (measurement): per-workgroup phase timestamps (s_memtime) of THIS batch's single-factor linearise into dev_buffer
 * ([2048][16] uint64: slots 0-7 phases, 8 HW_ID, 9 XCC_ID, 10 / 11 start / end on the device-wide clock; row 2047: the finalize kernel of the
 * synchronous call); NULL disables */
int gp_vgicp_batch_set_trace_buffer(gp_vgicp_batch_t* batch, void* dev_buffer);
#ifdef __cplusplus
}
#endif
#endif /* GTSAM_POINTS_HIP_H */
