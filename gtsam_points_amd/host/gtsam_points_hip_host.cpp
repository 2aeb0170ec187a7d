// gtsam_points_hip_host.cpp -- the out-of-line half of the C++ mirror: definitions of the free functions the REFERENCE declares in
// its own headers (types/gaussian_voxelmap.hpp:72-165 overlap_gpu, types/point_cloud_cpu.hpp:321-325 merge_frames_gpu,
// types/gaussian_voxelmap_gpu.hpp:110-114 download_*, types/point_cloud_gpu.hpp:139-142 download_*_gpu,
// cuda/nonlinear_factor_set_gpu_create.hpp:10), with exactly those signatures.  Compile this file into the application (or a
// small static library) next to libgtsam_points_hip.so.
#include <array>
#include <cstring>

#include "gaussian_voxelmap_gpu.hpp"
#include "nonlinear_factor_set_gpu.hpp"
#include "point_cloud_gpu.hpp"

namespace gtsam_points {

namespace {

const GaussianVoxelMapGPU* cast_gpu(const GaussianVoxelMap::ConstPtr& m) { return dynamic_cast<const GaussianVoxelMapGPU*>(m.get()); }

template <typename V>
std::vector<V> download_array(const void* dev, size_t n, CUstream_st* stream) {
  std::vector<V> out(n);
  if (n && dev) {
    check_error << gp_memcpy_d2h(out.data(), dev, sizeof(V) * n, gp_stream(stream));
    check_error << gp_stream_synchronize(gp_stream(stream));
  }
  return out;
}

// make_sure_loaded_on_gpu (types/gaussian_voxelmap_gpu_funcs.cu:21-40): an offloaded operand is touch()-reloaded before the lookup --
// "a bit hacky" upstream as well (const_cast), but overlap-based keyframe / loop selection must work with offloading enabled
void make_sure_loaded_on_gpu(const GaussianVoxelMapGPU* target, CUstream_st* stream) {
  if (target && !target->loaded_on_gpu()) const_cast<GaussianVoxelMapGPU*>(target)->touch(stream);
}
void make_sure_loaded_on_gpu(const PointCloud::ConstPtr& source, CUstream_st* stream) {
  if (source->points_gpu) return;  // already on the GPU
  auto source_gpu = std::dynamic_pointer_cast<const PointCloudGPU>(source);
  if (!source_gpu) {
    std::cerr << "error: Source point cloud is not a PointCloudGPU!!" << std::endl;  // :34-37
    abort();
  }
  const_cast<PointCloudGPU*>(source_gpu.get())->touch(stream);
  if (!source->points_gpu) {
    std::cerr << "error: GPU source points have not been allocated!!" << std::endl;
    abort();
  }
}

}  // namespace

// ---- overlap_gpu (types/gaussian_voxelmap_gpu_funcs.cu:192-406) ----------------------------------------------------------------

// device-resident pose: an Eigen::Isometry3f in GPU memory (:192-236).  It is brought to the host (64 B) and widened; the lookup
// itself runs with the double pose like every other entry point.
double overlap_gpu(const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source, const Eigen::Isometry3f* T_target_source_gpu, CUstream_st* stream) {
  Eigen::Isometry3f T;
  check_error << gp_memcpy_d2h(T.data(), T_target_source_gpu, sizeof(float) * 16, gp_stream(stream));
  check_error << gp_stream_synchronize(gp_stream(stream));
  return overlap_gpu(target, source, T.cast<double>(), stream);
}

double overlap_gpu(const GaussianVoxelMap::ConstPtr& target_, const PointCloud::ConstPtr& source, const Eigen::Isometry3d& T_target_source, CUstream_st* stream) {
  const GaussianVoxelMapGPU* target = cast_gpu(target_);
  if (!target) {
    std::cerr << "error: Failed to cast target voxelmap to GaussianVoxelMapGPU!!" << std::endl;  // :199-203
    abort();
  }
  make_sure_loaded_on_gpu(target, stream);  // :205-206, :251-252
  make_sure_loaded_on_gpu(source, stream);
  int hits = 0;
  check_error << gp_voxelmap_overlap(target->handle(), as_floats(source->points_gpu), static_cast<int>(source->size()), pose16(T_target_source).data(), &hits,
                                     gp_stream(stream));
  return source->size() ? static_cast<double>(hits) / source->size() : 0.0;
}

// a point cloud as the target (:238-263): its voxel map is built on the fly at the reference's default resolution handling --
// upstream this overload expects the target frame to carry a voxel map attribute and aborts otherwise; a frame that IS-A
// GaussianVoxelMapGPU-owning PointCloudGPU is not a concept of the mirror, so the target's points are voxelised here at 1.0 m,
// the resolution the reference's tests use for submaps (test_matching_cost_factors.cpp:84,89)
double overlap_gpu(const PointCloud::ConstPtr& target, const PointCloud::ConstPtr& source, const Eigen::Isometry3d& T_target_source, CUstream_st* stream) {
  make_sure_loaded_on_gpu(target, stream);
  make_sure_loaded_on_gpu(source, stream);
  if (!target->covs_gpu) {
    std::cerr << "error: target covariances are not on the GPU!!" << std::endl;
    abort();
  }
  auto map = std::make_shared<GaussianVoxelMapGPU>(1.0f, 8192 * 2, 10, 1e-3, stream);
  map->insert(*target);
  return overlap_gpu(std::static_pointer_cast<const GaussianVoxelMap>(map), source, T_target_source, stream);
}

// fraction of source points inside a voxel of ANY target (:265-335): one launch for all targets
double overlap_gpu(const std::vector<GaussianVoxelMap::ConstPtr>& targets_, const PointCloud::ConstPtr& source, const std::vector<Eigen::Isometry3d>& Ts_target_source,
                   CUstream_st* stream) {
  make_sure_loaded_on_gpu(source, stream);  // :270-273 (abort()s when the source is no PointCloudGPU and has no device points)
  std::vector<const gp_voxelmap_t*> handles(targets_.size());
  std::vector<double> deltas(16 * targets_.size());
  for (size_t i = 0; i < targets_.size(); i++) {
    const GaussianVoxelMapGPU* t = cast_gpu(targets_[i]);
    if (!t) std::cerr << "error: Failed to cast target voxelmap to GaussianVoxelMapGPU!!" << std::endl;  // :278-280 (no abort upstream)
    make_sure_loaded_on_gpu(t, stream);
    handles[i] = t ? t->handle() : nullptr;
    std::memcpy(deltas.data() + 16 * i, pose16(Ts_target_source[i]).data(), sizeof(double) * 16);
  }
  int hits = 0;
  check_error << gp_voxelmap_overlap_multi(handles.data(), deltas.data(), static_cast<int>(handles.size()), as_floats(source->points_gpu), static_cast<int>(source->size()),
                                           &hits, gp_stream(stream));
  return source->size() ? static_cast<double>(hits) / source->size() : 0.0;
}

// one rate per (target[i], source[i]) pair (:337-404): ONE launch for all pairs
std::vector<double> overlap_gpu(const std::vector<GaussianVoxelMap::ConstPtr>& targets_, const std::vector<PointCloud::ConstPtr>& sources,
                                const std::vector<Eigen::Isometry3d>& Ts_target_source, CUstream_st* stream) {
  if (targets_.size() != sources.size()) {
    std::cerr << "error: The number of target voxelmaps and source point clouds must be the same!!" << std::endl;  // :342-345
    abort();
  }
  const size_t P = sources.size();
  std::vector<const gp_voxelmap_t*> handles(P);
  std::vector<const float*> pts(P);
  std::vector<int> ns(P), hits(P, 0);
  std::vector<double> deltas(16 * P);
  for (size_t i = 0; i < P; i++) {
    const GaussianVoxelMapGPU* t = cast_gpu(targets_[i]);
    if (!t) std::cerr << "error: Failed to cast target voxelmap to GaussianVoxelMapGPU!!" << std::endl;
    make_sure_loaded_on_gpu(t, stream);
    make_sure_loaded_on_gpu(sources[i], stream);
    handles[i] = t ? t->handle() : nullptr;
    pts[i] = as_floats(sources[i]->points_gpu);
    ns[i] = static_cast<int>(sources[i]->size());
    std::memcpy(deltas.data() + 16 * i, pose16(Ts_target_source[i]).data(), sizeof(double) * 16);
  }
  if (P) check_error << gp_voxelmap_overlap_batch(handles.data(), pts.data(), ns.data(), deltas.data(), static_cast<int>(P), hits.data(), gp_stream(stream));
  std::vector<double> rates(P);
  for (size_t i = 0; i < P; i++) rates[i] = ns[i] ? static_cast<double>(hits[i]) / ns[i] : 0.0;
  return rates;
}

// ---- merge_frames_gpu (gaussian_voxelmap_gpu_funcs.cu:65-152) --------------------------------------------------------------------
// the merged cloud is the voxel arrays of the down-sampling map, handed over device to device; the CPU attributes (points, covs,
// intensities) are filled from ONE download of those arrays, so that the result is a complete frame like the reference's
// (its add_points / add_covs / add_intensities at :146-149 go host -> device; here the device copy exists first)
PointCloud::Ptr merge_frames_gpu(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution,
                                 CUstream_st* stream) {
  const size_t F = frames.size();
  std::vector<const float*> pts(F), covs(F), ints(F);
  std::vector<int> ns(F);
  std::vector<double> flat(16 * F);
  for (size_t i = 0; i < F; i++) {
    make_sure_loaded_on_gpu(frames[i], stream);
    pts[i] = as_floats(frames[i]->points_gpu);
    covs[i] = as_floats(frames[i]->covs_gpu);
    ints[i] = frames[i]->intensities_gpu;
    ns[i] = static_cast<int>(frames[i]->size());
    std::memcpy(flat.data() + 16 * i, pose16(poses[i]).data(), sizeof(double) * 16);
  }
  gp_voxelmap_t* map = nullptr;
  check_error << gp_merge_frames(flat.data(), pts.data(), covs.data(), ints.data(), ns.data(), static_cast<int>(F), downsample_resolution, 1e-3, gp_stream(stream), &map);
  auto merged = std::make_shared<PointCloudGPU>();
  if (!map) return merged;
  gp_voxelmap_info info;
  gp_voxelmap_views views;
  check_error << gp_voxelmap_info_get(map, &info);
  check_error << gp_voxelmap_views_get(map, &views);
  const size_t V = static_cast<size_t>(info.num_voxels);
  void *p = nullptr, *c = nullptr, *it = nullptr;
  check_error << gp_malloc(&p, 12 * V);
  check_error << gp_malloc(&c, 36 * V);
  check_error << gp_malloc(&it, 4 * V);
  check_error << gp_memcpy_d2d(p, views.voxel_means, 12 * V, gp_stream(stream));
  check_error << gp_memcpy_d2d(c, views.voxel_covs, 36 * V, gp_stream(stream));
  check_error << gp_memcpy_d2d(it, views.voxel_intensities, 4 * V, gp_stream(stream));
  check_error << gp_stream_synchronize(gp_stream(stream));
  merged->adopt(static_cast<float*>(p), static_cast<float*>(c), static_cast<float*>(it), V);
  merged->download_attributes(stream);  // points / covs / intensities on the CPU side as well
  check_error << gp_voxelmap_destroy(map);
  return merged;
}

// ---- download_* ---------------------------------------------------------------------------------------------------------------------
std::vector<VoxelBucket> download_buckets(const GaussianVoxelMapGPU& m, CUstream_st* stream) {
  return download_array<VoxelBucket>(m.buckets, (size_t)m.voxelmap_info.num_buckets, stream);
}
std::vector<int> download_voxel_num_points(const GaussianVoxelMapGPU& m, CUstream_st* stream) {
  return download_array<int>(m.num_points, (size_t)m.voxelmap_info.num_voxels, stream);
}
std::vector<Eigen::Vector3f> download_voxel_means(const GaussianVoxelMapGPU& m, CUstream_st* stream) {
  return download_array<Eigen::Vector3f>(m.voxel_means, (size_t)m.voxelmap_info.num_voxels, stream);
}
std::vector<Eigen::Matrix3f> download_voxel_covs(const GaussianVoxelMapGPU& m, CUstream_st* stream) {
  return download_array<Eigen::Matrix3f>(m.voxel_covs, (size_t)m.voxelmap_info.num_voxels, stream);
}
std::vector<float> download_voxel_intensities(const GaussianVoxelMapGPU& m, CUstream_st* stream) {
  return download_array<float>(m.voxel_intensities, (size_t)m.voxelmap_info.num_voxels, stream);
}

std::vector<Eigen::Vector3f> download_points_gpu(const PointCloud& frame, CUstream_st* stream) { return download_array<Eigen::Vector3f>(frame.points_gpu, frame.size(), stream); }
std::vector<Eigen::Matrix3f> download_covs_gpu(const PointCloud& frame, CUstream_st* stream) { return download_array<Eigen::Matrix3f>(frame.covs_gpu, frame.size(), stream); }
std::vector<Eigen::Vector3f> download_normals_gpu(const PointCloud& frame, CUstream_st* stream) { return download_array<Eigen::Vector3f>(frame.normals_gpu, frame.size(), stream); }
std::vector<float> download_intensities_gpu(const PointCloud& frame, CUstream_st* stream) { return download_array<float>(frame.intensities_gpu, frame.size(), stream); }

std::shared_ptr<NonlinearFactorSet> create_nonlinear_factor_set_gpu() { return std::make_shared<NonlinearFactorSetGPU>(); }

}  // namespace gtsam_points
