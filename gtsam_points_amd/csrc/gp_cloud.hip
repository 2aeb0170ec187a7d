// gp_cloud.hip -- the callers and data formats either side of the VGICP path:
//
//   * overlap_gpu over several targets / several pairs   types/gaussian_voxelmap_gpu_funcs.cu:265-406
//       the reference runs one thrust::transform + one CUB reduce (called twice) per target or pair, with a bool
//       array in HBM in between; here ONE launch walks a tile table over all pairs, every point probes its targets in
//       registers, hits are counted with ballot + one atomic per wave.
//   * merge_frames_gpu                                     types/gaussian_voxelmap_gpu_funcs.cu:65-152
//       transform every frame by its pose (f64 arithmetic on the f32 inputs, one launch for all frames), then the
//       Gaussian voxel-map build at the down-sampling resolution; the merged cloud is the map's voxel arrays (already
//       in the PointCloudGPU layout), handed over device-to-device instead of the reference's D2H + H2D round trip.
//   * PointCloudGPU::add_points_gpu / add_covs_gpu / add_normals_gpu   types/point_cloud_gpu.cu:26-62,110-201
//       the reference converts double/float 3- or 4-vectors element by element on the host into a temporary vector and
//       synchronises per attribute; here the raw host array goes up as it is (pinned staging) and a pack kernel writes the
//       float[N][3] / float[N][9] device layout.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "gp_host.hpp"

namespace gp {

struct OverlapTarget {
  VoxelMapView map;
  double pose[16];
};

struct OverlapJob {
  const float* points;
  int n;
  int target_begin, target_count;  // union over these targets
};

struct OverlapTile {
  int job, begin;
};

// a point counts once if delta_k * p falls into a voxel of ANY of the job's targets (bool_or_kernel, :180-182)
__global__ void __launch_bounds__(256) overlap_jobs_kernel(const OverlapJob* __restrict__ jobs, const OverlapTile* __restrict__ tiles,
                                                           const OverlapTarget* __restrict__ targets, int* __restrict__ hit_counts) {
  const OverlapTile tile = tiles[blockIdx.x];
  const OverlapJob job = jobs[tile.job];
  const int i = tile.begin + threadIdx.x;
  bool hit = false;
  if (i < job.n) {
    const double px = (double)job.points[3 * (size_t)i], py = (double)job.points[3 * (size_t)i + 1], pz = (double)job.points[3 * (size_t)i + 2];
    for (int k = 0; k < job.target_count && !hit; k++) {
      const OverlapTarget& t = targets[job.target_begin + k];
      const Pose T = load_pose(t.pose);
      const double qx = T.r00 * px + T.r01 * py + T.r02 * pz + T.tx;
      const double qy = T.r10 * px + T.r11 * py + T.r12 * pz + T.ty;
      const double qz = T.r20 * px + T.r21 * py + T.r22 * pz + T.tz;
      hit = finite3(qx, qy, qz) && lookup_voxel(t.map, fast_floor(qx * t.map.inv_leaf), fast_floor(qy * t.map.inv_leaf), fast_floor(qz * t.map.inv_leaf)) >= 0;
    }
  }
  const unsigned long long hits = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && hits) atomicAdd(hit_counts + tile.job, __popcll(hits));
}

struct FrameDesc {
  const float* points;
  const float* covs;
  const float* intensities;  // may be null: zeros (:103-107)
  int n;
  int out_begin;
  double pose[16];
};

// transform_means_kernel / transform_covs_kernel (:42-62) for all frames in one launch; f64 arithmetic, f32 results
__global__ void __launch_bounds__(256) transform_frames_kernel(const FrameDesc* __restrict__ frames, const OverlapTile* __restrict__ tiles,
                                                               float* __restrict__ out_points, float* __restrict__ out_covs, float* __restrict__ out_intensities) {
  const OverlapTile tile = tiles[blockIdx.x];
  const FrameDesc& f = frames[tile.job];
  const int i = tile.begin + threadIdx.x;
  if (i >= f.n) return;
  const Pose T = load_pose(f.pose);
  const size_t o = (size_t)f.out_begin + i;
  const float* p = f.points + 3 * (size_t)i;
  const double px = (double)p[0], py = (double)p[1], pz = (double)p[2];
  out_points[3 * o] = (float)(T.r00 * px + T.r01 * py + T.r02 * pz + T.tx);
  out_points[3 * o + 1] = (float)(T.r10 * px + T.r11 * py + T.r12 * pz + T.ty);
  out_points[3 * o + 2] = (float)(T.r20 * px + T.r21 * py + T.r22 * pz + T.tz);
  if (out_covs) {
    const float* c = f.covs + 9 * (size_t)i;  // column-major 3x3
    const double R[3][3] = {{T.r00, T.r01, T.r02}, {T.r10, T.r11, T.r12}, {T.r20, T.r21, T.r22}};
    double RC[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int col = 0; col < 3; col++) RC[r][col] = R[r][0] * (double)c[col * 3] + R[r][1] * (double)c[col * 3 + 1] + R[r][2] * (double)c[col * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int col = 0; col < 3; col++) out_covs[9 * o + col * 3 + r] = (float)(RC[r][0] * R[col][0] + RC[r][1] * R[col][1] + RC[r][2] * R[col][2]);
  }
  if (out_intensities) out_intensities[o] = f.intensities ? f.intensities[i] : 0.0f;
}

// host array of SRC_DIM-vectors (T = double | float) -> float[N][3]           (add_points_gpu / add_normals_gpu)
template <typename T>
__global__ void __launch_bounds__(256) pack_vec3_kernel(const T* __restrict__ src, int src_dim, int n, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n) return;
  const T* s = src + (size_t)src_dim * i;
  dst[3 * i] = (float)s[0];
  dst[3 * i + 1] = (float)s[1];
  dst[3 * i + 2] = (float)s[2];
}

// host array of column-major DIM x DIM matrices -> the top-left 3x3 as float[N][9] column-major   (add_covs_gpu)
template <typename T>
__global__ void __launch_bounds__(256) pack_mat3_kernel(const T* __restrict__ src, int src_dim, int n, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n) return;
  const T* s = src + (size_t)src_dim * src_dim * i;
#pragma unroll
  for (int col = 0; col < 3; col++)
#pragma unroll
    for (int r = 0; r < 3; r++) dst[9 * i + col * 3 + r] = (float)s[col * src_dim + r];
}

// the packed mirror of a source cloud (SourceMirror, gp_host.hpp): thread = point; a workgroup writes four 2304-byte chunks.  flag |= 1 when some
// covariance is not symmetric to the last bit (NaN entries compare unequal and land here as well): such a cloud keeps the API layout.
__global__ void __launch_bounds__(256) pack_source_mirror_kernel(const float* __restrict__ points, const float* __restrict__ covs, int n, char* __restrict__ dst,
                                                                 int* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if ((i >> 6) >= ((size_t)n + 63) / 64) return;  // whole waves behind the last chunk (the grid is rounded up to four chunks per workgroup)
  float p[3] = {0.f, 0.f, 0.f}, c[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < (size_t)n) {
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = points[3 * i + k];
#pragma unroll
    for (int k = 0; k < 9; k++) c[k] = covs[9 * i + k];
  }
  // column-major 3x3: (r, c) = c9[3 c + r]; the kernels read c00, c01 = c9[3], c02 = c9[6], c11 = c9[4], c12 = c9[7], c22 = c9[8] (load_cov6)
  const bool asym = c[3] != c[1] || c[6] != c[2] || c[7] != c[5];
  if (__builtin_amdgcn_ballot_w64(asym) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
  float* chunk = reinterpret_cast<float*>(dst + (i >> 6) * (size_t)kMirrorChunkBytes);  // (the last chunk is padded with zeros; the kernels never stream it)
  const int l = (int)(i & 63);
  chunk[3 * l + 0] = p[0], chunk[3 * l + 1] = p[1], chunk[3 * l + 2] = p[2];
  chunk[192 + 3 * l + 0] = c[0], chunk[192 + 3 * l + 1] = c[3], chunk[192 + 3 * l + 2] = c[6];
  chunk[384 + 3 * l + 0] = c[4], chunk[384 + 3 * l + 1] = c[7], chunk[384 + 3 * l + 2] = c[8];
}

namespace {
using MirrorKey = std::tuple<int, const float*, const float*, int>;
std::recursive_mutex g_mirror_mutex;  // (recursive: a mirror that fails to pack is destroyed under the lock, and its destructor takes it)
std::map<MirrorKey, std::weak_ptr<SourceMirror>>& mirror_registry() {
  static auto* m = new std::map<MirrorKey, std::weak_ptr<SourceMirror>>;  // never destroyed: factors may outlive static destruction order
  return *m;
}
std::atomic<long long> g_mirror_bytes{0};
}  // namespace

SourceMirror::~SourceMirror() {
  g_mirror_bytes -= (long long)data.bytes;
  std::lock_guard<std::recursive_mutex> lock(g_mirror_mutex);
  auto& reg = mirror_registry();
  auto it = reg.find(MirrorKey{device, points, covs, n});
  if (it != reg.end() && it->second.expired()) reg.erase(it);
}

int acquire_source_mirror(const float* points, const float* covs, int n, int device, hipStream_t stream, std::shared_ptr<SourceMirror>* out) {
  out->reset();
  if (!points || !covs || n < 64) return GP_OK;
  std::lock_guard<std::recursive_mutex> lock(g_mirror_mutex);  // (held through the pack: two threads asking for the same cloud get one mirror)
  auto& reg = mirror_registry();
  const MirrorKey key{device, points, covs, n};
  auto it = reg.find(key);
  if (it != reg.end()) {
    if (auto live = it->second.lock()) {
      *out = live;
      return GP_OK;
    }
  }
  int cur = 0;
  GP_HIP(hipGetDevice(&cur));
  if (cur != device) GP_HIP(hipSetDevice(device));
  auto m = std::make_shared<SourceMirror>();
  m->points = points, m->covs = covs, m->n = n, m->device = device;
  const size_t chunks = ((size_t)n + 63) / 64;
  DeviceArray flag;
  int rc = m->data.alloc(chunks * (size_t)kMirrorChunkBytes);
  if (rc == GP_OK) {
    g_mirror_bytes += (long long)m->data.bytes;
    rc = flag.alloc_async(sizeof(int), stream);
  }
  int h_flag = 1;
  if (rc == GP_OK) {
    hipError_t e = hipMemsetAsync(flag.ptr, 0, sizeof(int), stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(pack_source_mirror_kernel, dim3((unsigned)((chunks * 64 + 255) / 256)), dim3(256), 0, stream, points, covs, n, m->data.as<char>(), flag.as<int>());
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, flag.ptr, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) rc = hip_fail(e, "pack_source_mirror_kernel", __FILE__, __LINE__);
  }
  flag.release_on(stream);
  if (cur != device) (void)hipSetDevice(cur);
  if (rc != GP_OK) return rc;
  m->usable = h_flag == 0;
  if (!m->usable) {  // nothing will stream it: give the memory back, keep the verdict
    g_mirror_bytes -= (long long)m->data.bytes;
    m->data.release();
  }
  reg[key] = m;
  *out = m;
  return GP_OK;
}

}  // namespace gp

namespace {

int make_tiles(const std::vector<int>& sizes, std::vector<gp::OverlapTile>* tiles) {
  tiles->clear();
  for (size_t j = 0; j < sizes.size(); j++)
    for (int b = 0; b < sizes[j]; b += 256) tiles->push_back(gp::OverlapTile{(int)j, b});
  return (int)tiles->size();
}

// shared driver of the overlap entry points: jobs over a flat target list
int run_overlap(const std::vector<gp::OverlapJob>& jobs, const std::vector<gp::OverlapTarget>& targets, int* num_hits, hipStream_t s) {
  const size_t J = jobs.size();
  for (size_t j = 0; j < J; j++) num_hits[j] = 0;
  std::vector<int> sizes(J);
  for (size_t j = 0; j < J; j++) sizes[j] = jobs[j].n;
  std::vector<gp::OverlapTile> tiles;
  const int T = make_tiles(sizes, &tiles);
  if (T == 0) return GP_OK;
  gp::DeviceArray d_jobs, d_tiles, d_targets, d_counts;
  GP_TRY(d_jobs.alloc(sizeof(gp::OverlapJob) * J));
  GP_TRY(d_tiles.alloc(sizeof(gp::OverlapTile) * (size_t)T));
  GP_TRY(d_targets.alloc(sizeof(gp::OverlapTarget) * std::max<size_t>(targets.size(), 1)));
  GP_TRY(d_counts.alloc(sizeof(int) * J));
  GP_HIP(hipMemcpyAsync(d_jobs.ptr, jobs.data(), sizeof(gp::OverlapJob) * J, hipMemcpyHostToDevice, s));
  GP_HIP(hipMemcpyAsync(d_tiles.ptr, tiles.data(), sizeof(gp::OverlapTile) * (size_t)T, hipMemcpyHostToDevice, s));
  if (!targets.empty()) GP_HIP(hipMemcpyAsync(d_targets.ptr, targets.data(), sizeof(gp::OverlapTarget) * targets.size(), hipMemcpyHostToDevice, s));
  GP_HIP(hipMemsetAsync(d_counts.ptr, 0, sizeof(int) * J, s));
  hipLaunchKernelGGL(gp::overlap_jobs_kernel, dim3(T), dim3(256), 0, s, d_jobs.as<gp::OverlapJob>(), d_tiles.as<gp::OverlapTile>(),
                     d_targets.as<gp::OverlapTarget>(), d_counts.as<int>());
  GP_HIP(hipGetLastError());
  GP_HIP(hipMemcpyAsync(num_hits, d_counts.ptr, sizeof(int) * J, hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));  // the uploads above come from pageable vectors that die with this frame
  return GP_OK;
}

int fill_target(const gp_voxelmap_t* map, const double* delta, gp::OverlapTarget* t) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: Failed to cast target voxelmap to GaussianVoxelMapGPU!!");  // :274-276
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "overlap: target voxel map is not loaded on the GPU");
  t->map = map->view();
  memcpy(t->pose, delta, sizeof(double) * 16);
  return GP_OK;
}

template <typename T>
int pack_upload(const void* src_host, int src_dim, int n, float* dst_dev, bool matrix, hipStream_t s) {
  const size_t per = matrix ? (size_t)src_dim * src_dim : (size_t)src_dim;
  const size_t bytes = sizeof(T) * per * (size_t)n;
  gp::DeviceArray staging;
  GP_TRY(staging.alloc(bytes));
  GP_HIP(hipMemcpyAsync(staging.ptr, src_host, bytes, hipMemcpyHostToDevice, s));
  const dim3 grid((unsigned)(((size_t)n + 255) / 256));
  if (matrix) {
    hipLaunchKernelGGL(gp::pack_mat3_kernel<T>, grid, dim3(256), 0, s, staging.as<T>(), src_dim, n, dst_dev);
  } else {
    hipLaunchKernelGGL(gp::pack_vec3_kernel<T>, grid, dim3(256), 0, s, staging.as<T>(), src_dim, n, dst_dev);
  }
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize(s));  // the staging buffer is released on return (the reference syncs per attribute too, :45,61)
  return GP_OK;
}

}  // namespace

extern "C" {

int gp_voxelmap_overlap_multi(const gp_voxelmap_t* const* targets, const double* deltas, int num_targets, const float* points_dev, int num_points, int* num_hits,
                              gp_stream_t stream) {
  if (!targets || !deltas || num_targets < 0 || !num_hits || num_points < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_overlap_multi: bad arguments");
  if (!points_dev && num_points > 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: GPU source points have not been allocated!!");  // :270-273
  std::vector<gp::OverlapTarget> ts((size_t)num_targets);
  for (int k = 0; k < num_targets; k++) GP_TRY(fill_target(targets[k], deltas + 16 * (size_t)k, &ts[k]));
  std::vector<gp::OverlapJob> jobs(1, gp::OverlapJob{points_dev, num_points, 0, num_targets});
  return run_overlap(jobs, ts, num_hits, (hipStream_t)stream);
}

int gp_voxelmap_overlap_batch(const gp_voxelmap_t* const* targets, const float* const* points_dev, const int* num_points, const double* deltas, int num_pairs,
                              int* num_hits, gp_stream_t stream) {
  if (!targets || !points_dev || !num_points || !deltas || num_pairs < 0 || !num_hits)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_overlap_batch: bad arguments");
  std::vector<gp::OverlapTarget> ts((size_t)num_pairs);
  std::vector<gp::OverlapJob> jobs((size_t)num_pairs);
  for (int k = 0; k < num_pairs; k++) {
    GP_TRY(fill_target(targets[k], deltas + 16 * (size_t)k, &ts[k]));
    if (num_points[k] < 0 || (!points_dev[k] && num_points[k] > 0)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: GPU source points have not been allocated!!");
    jobs[k] = gp::OverlapJob{points_dev[k], num_points[k], k, 1};
  }
  if (num_pairs == 0) return GP_OK;
  return run_overlap(jobs, ts, num_hits, (hipStream_t)stream);
}

int gp_transform_frames(const double* poses, const float* const* points_dev, const float* const* covs_dev, const float* const* intensities_dev, const int* num_points,
                        int num_frames, float* out_points_dev, float* out_covs_dev, float* out_intensities_dev, gp_stream_t stream) {
  if (!poses || !points_dev || !num_points || num_frames < 0 || !out_points_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_transform_frames: bad arguments");
  if (out_covs_dev && !covs_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_transform_frames: covariances requested but not given");
  std::vector<gp::FrameDesc> frames((size_t)num_frames);
  std::vector<int> sizes((size_t)num_frames);
  int begin = 0;
  for (int i = 0; i < num_frames; i++) {
    if (num_points[i] < 0 || (num_points[i] > 0 && (!points_dev[i] || (out_covs_dev && !covs_dev[i]))))
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_transform_frames: frame without GPU points / covariances");
    gp::FrameDesc& f = frames[i];
    f.points = points_dev[i];
    f.covs = covs_dev ? covs_dev[i] : nullptr;
    f.intensities = intensities_dev ? intensities_dev[i] : nullptr;
    f.n = sizes[i] = num_points[i];
    f.out_begin = begin;
    memcpy(f.pose, poses + 16 * (size_t)i, sizeof(double) * 16);
    begin += num_points[i];
  }
  std::vector<gp::OverlapTile> tiles;
  const int T = make_tiles(sizes, &tiles);
  if (T == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  gp::DeviceArray d_frames, d_tiles;
  GP_TRY(d_frames.alloc(sizeof(gp::FrameDesc) * (size_t)num_frames));
  GP_TRY(d_tiles.alloc(sizeof(gp::OverlapTile) * (size_t)T));
  GP_HIP(hipMemcpyAsync(d_frames.ptr, frames.data(), sizeof(gp::FrameDesc) * (size_t)num_frames, hipMemcpyHostToDevice, s));
  GP_HIP(hipMemcpyAsync(d_tiles.ptr, tiles.data(), sizeof(gp::OverlapTile) * (size_t)T, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(gp::transform_frames_kernel, dim3(T), dim3(256), 0, s, d_frames.as<gp::FrameDesc>(), d_tiles.as<gp::OverlapTile>(), out_points_dev, out_covs_dev,
                     out_intensities_dev);
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

int gp_merge_frames(const double* poses, const float* const* points_dev, const float* const* covs_dev, const float* const* intensities_dev, const int* num_points,
                    int num_frames, double downsample_resolution, double target_points_drop_rate, gp_stream_t stream, gp_voxelmap_t** out_map) {
  if (!out_map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_merge_frames: null out");
  *out_map = nullptr;
  if (!poses || !points_dev || !covs_dev || !num_points || num_frames <= 0 || !(downsample_resolution > 0.0))
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_merge_frames: bad arguments");
  int64_t total = 0;
  for (int i = 0; i < num_frames; i++) total += std::max(num_points[i], 0);
  if (total <= 0 || total > (int64_t(1) << 30)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_merge_frames: total number of points must be in [1, 2^30]");
  gp::DeviceArray all_points, all_covs, all_ints;
  GP_TRY(all_points.alloc(sizeof(float) * 3 * (size_t)total));
  GP_TRY(all_covs.alloc(sizeof(float) * 9 * (size_t)total));
  GP_TRY(all_ints.alloc(sizeof(float) * (size_t)total));
  GP_TRY(gp_transform_frames(poses, points_dev, covs_dev, intensities_dev, num_points, num_frames, all_points.as<float>(), all_covs.as<float>(), all_ints.as<float>(),
                             stream));
  // GaussianVoxelMapGPU downsampling(resolution, num_all_points, 10, 1e-3, stream)  (:122)
  gp_voxelmap_t* map = nullptr;
  GP_TRY(gp_voxelmap_create(downsample_resolution, (int)total, 10, target_points_drop_rate, stream, &map));
  const int rc = gp_voxelmap_insert(map, all_points.as<float>(), all_covs.as<float>(), all_ints.as<float>(), (int)total);
  if (rc != GP_OK) {
    gp_voxelmap_destroy(map);
    return rc;
  }
  *out_map = map;
  return GP_OK;
}

// an owner is about to rewrite or free device arrays that factors may have mirrored (PointCloudGPU::add_*_gpu on a live cloud, offload_gpu,
// the destructor): forget every mirror built from `dev_ptr` (as points or covariances), so that a later factor on the same address packs afresh.
// Factors that still hold such a mirror keep streaming it until gp_vgicp_factor_set_source hands them the new arrays (they borrow the old ones).
int gp_source_mirror_invalidate(const void* dev_ptr) {
  if (!dev_ptr) return GP_OK;
  std::lock_guard<std::recursive_mutex> lock(gp::g_mirror_mutex);
  auto& reg = gp::mirror_registry();
  for (auto it = reg.begin(); it != reg.end();) {
    if (std::get<1>(it->first) == dev_ptr || std::get<2>(it->first) == dev_ptr) it = reg.erase(it);
    else ++it;
  }
  return GP_OK;
}

// device bytes held by live packed mirrors (all devices): what the factors' memory_usage_gpu() adds on top of the caller's arrays
int64_t gp_source_mirror_bytes(void) { return (int64_t)gp::g_mirror_bytes.load(); }

int gp_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, gp_stream_t stream) {
  if (bytes == 0) return GP_OK;
  if (!dst_dev || !src_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_memcpy_d2d: null pointer");
  GP_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GP_OK;
}

int gp_cloud_upload_vec3(const void* src_host, int src_is_double, int src_dim, int num_points, float* dst_dev, gp_stream_t stream) {
  if (num_points < 0 || (src_dim != 3 && src_dim != 4)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_cloud_upload_vec3: dim must be 3 or 4");
  if (num_points == 0) return GP_OK;
  if (!src_host || !dst_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_cloud_upload_vec3: null pointer");
  return src_is_double ? pack_upload<double>(src_host, src_dim, num_points, dst_dev, false, (hipStream_t)stream)
                       : pack_upload<float>(src_host, src_dim, num_points, dst_dev, false, (hipStream_t)stream);
}

int gp_cloud_upload_mat3(const void* src_host, int src_is_double, int src_dim, int num_points, float* dst_dev, gp_stream_t stream) {
  if (num_points < 0 || (src_dim != 3 && src_dim != 4)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_cloud_upload_mat3: dim must be 3 or 4");
  if (num_points == 0) return GP_OK;
  if (!src_host || !dst_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_cloud_upload_mat3: null pointer");
  return src_is_double ? pack_upload<double>(src_host, src_dim, num_points, dst_dev, true, (hipStream_t)stream)
                       : pack_upload<float>(src_host, src_dim, num_points, dst_dev, true, (hipStream_t)stream);
}

}  // extern "C"
