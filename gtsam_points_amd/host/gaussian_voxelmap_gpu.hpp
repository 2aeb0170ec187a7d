// gaussian_voxelmap_gpu.hpp -- GaussianVoxelMapGPU (types/gaussian_voxelmap_gpu.hpp:38-114) over the C-ABI.
#pragma once
#include <gtsam_points_hip.h>

#include <array>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "check_error.hpp"
#include "point_cloud_gpu.hpp"

struct ihipStream_t;

namespace gtsam_points {

using VoxelMapInfo = gp_voxelmap_info;  // gaussian_voxelmap_gpu.hpp:20-25
using VoxelBucket = gp_voxel_bucket;    // :30-33 ({coord[3], voxel_index} == {first, second})

class GaussianVoxelMap {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMap>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMap>;
  virtual ~GaussianVoxelMap() {}
  virtual double voxel_resolution() const = 0;
  virtual void insert(const PointCloud& frame) = 0;
  virtual void save_compact(const std::string& path) const = 0;
};

class GaussianVoxelMapGPU : public GaussianVoxelMap, public OffloadableGPU {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMapGPU>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMapGPU>;

  GaussianVoxelMapGPU(float resolution, int init_num_buckets = 8192 * 2, int max_bucket_scan_count = 10, double target_points_drop_rate = 1e-3, ihipStream_t* stream = nullptr)
  : stream(stream), init_num_buckets(init_num_buckets), target_points_drop_rate(target_points_drop_rate) {
    check_error << gp_voxelmap_create(resolution, init_num_buckets, max_bucket_scan_count, target_points_drop_rate, stream, &h);
    refresh();
  }
  ~GaussianVoxelMapGPU() override { check_error << gp_voxelmap_destroy(h); }

  double voxel_resolution() const override { return gp_voxelmap_resolution(h); }

  void insert(const PointCloud& frame) override {
    if (!frame.check_points_gpu() || !frame.check_covs_gpu()) {
      std::cerr << "error: GPU points/covs not allocated!!" << std::endl;  // gaussian_voxelmap_gpu.cu:212-215
      abort();
    }
    check_error << gp_voxelmap_insert(h, frame.points_gpu, frame.covs_gpu, frame.intensities_gpu, static_cast<int>(frame.size()));
    refresh();
  }

  void save_compact(const std::string& path) const override { check_error << gp_voxelmap_save_compact(h, path.c_str()); }

  static GaussianVoxelMapGPU::Ptr load(const std::string& path) {
    gp_voxelmap_t* loaded = nullptr;
    if (gp_voxelmap_load(path.c_str(), nullptr, &loaded) != GP_OK) {
      std::cerr << gp_last_error() << std::endl;
      return nullptr;  // gaussian_voxelmap_gpu.cu:374-377
    }
    auto map = std::shared_ptr<GaussianVoxelMapGPU>(new GaussianVoxelMapGPU(loaded));
    return map;
  }

  size_t memory_usage_gpu() const override { return gp_voxelmap_memory_usage_gpu(h); }
  bool loaded_on_gpu() const override { return gp_voxelmap_loaded_on_gpu(h) != 0; }
  bool offload_gpu(ihipStream_t* s = nullptr) override {
    const bool ok = gp_voxelmap_offload(h, s) == GP_OK;
    refresh();
    return ok;
  }
  bool reload_gpu(ihipStream_t* s = nullptr) override {
    if (loaded_on_gpu()) return false;  // gaussian_voxelmap_gpu.cu:509-511
    const bool ok = gp_voxelmap_reload(h, s) == GP_OK;
    refresh();
    return ok;
  }

  gp_voxelmap_t* handle() const { return h; }

public:
  // the reference's public data members (gaussian_voxelmap_gpu.hpp:86-101), refreshed after every mutating call
  ihipStream_t* stream;
  const int init_num_buckets;
  const double target_points_drop_rate;
  VoxelMapInfo voxelmap_info{};
  const VoxelBucket* buckets = nullptr;
  const int* num_points = nullptr;
  const float* voxel_means = nullptr;       // Eigen::Vector3f[num_voxels]
  const float* voxel_covs = nullptr;        // Eigen::Matrix3f[num_voxels]
  const float* voxel_intensities = nullptr;

private:
  explicit GaussianVoxelMapGPU(gp_voxelmap_t* adopted) : stream(nullptr), init_num_buckets(8192), target_points_drop_rate(0.1), h(adopted) { refresh(); }
  void refresh() {
    check_error << gp_voxelmap_info_get(h, &voxelmap_info);
    gp_voxelmap_views v{};
    check_error << gp_voxelmap_views_get(h, &v);
    buckets = v.buckets;
    num_points = v.num_points;
    voxel_means = v.voxel_means;
    voxel_covs = v.voxel_covs;
    voxel_intensities = v.voxel_intensities;
  }
  gp_voxelmap_t* h = nullptr;
};

// download_* (gaussian_voxelmap_gpu.hpp:110-114)
inline std::vector<VoxelBucket> download_buckets(const GaussianVoxelMapGPU& m, ihipStream_t* = nullptr) {
  std::vector<VoxelBucket> out(m.voxelmap_info.num_buckets);
  check_error << gp_voxelmap_download(m.handle(), out.data(), nullptr, nullptr, nullptr, nullptr);
  return out;
}
inline std::vector<int> download_voxel_num_points(const GaussianVoxelMapGPU& m, ihipStream_t* = nullptr) {
  std::vector<int> out(m.voxelmap_info.num_voxels);
  check_error << gp_voxelmap_download(m.handle(), nullptr, out.data(), nullptr, nullptr, nullptr);
  return out;
}
inline std::vector<float> download_voxel_means(const GaussianVoxelMapGPU& m, ihipStream_t* = nullptr) {  // xyz per voxel
  std::vector<float> out(3 * (size_t)m.voxelmap_info.num_voxels);
  check_error << gp_voxelmap_download(m.handle(), nullptr, nullptr, out.data(), nullptr, nullptr);
  return out;
}
inline std::vector<float> download_voxel_covs(const GaussianVoxelMapGPU& m, ihipStream_t* = nullptr) {  // 3x3 col-major per voxel
  std::vector<float> out(9 * (size_t)m.voxelmap_info.num_voxels);
  check_error << gp_voxelmap_download(m.handle(), nullptr, nullptr, nullptr, out.data(), nullptr);
  return out;
}
inline std::vector<float> download_voxel_intensities(const GaussianVoxelMapGPU& m, ihipStream_t* = nullptr) {
  std::vector<float> out(m.voxelmap_info.num_voxels);
  check_error << gp_voxelmap_download(m.handle(), nullptr, nullptr, nullptr, nullptr, out.data());
  return out;
}

// overlap_gpu(target, source, delta) (types/gaussian_voxelmap_gpu_funcs.cu:192-236); delta = column-major 4x4 double
inline double overlap_gpu(const GaussianVoxelMap::ConstPtr& target_, const PointCloud::ConstPtr& source, const double delta[16]) {
  auto target = std::dynamic_pointer_cast<const GaussianVoxelMapGPU>(target_);
  if (!target || !source->points_gpu) {
    std::cerr << "error: target voxelmap or source points are not on the GPU!!" << std::endl;  // :194-203
    abort();
  }
  int hits = 0;
  check_error << gp_voxelmap_overlap(target->handle(), source->points_gpu, static_cast<int>(source->size()), delta, &hits, nullptr);
  return source->size() ? static_cast<double>(hits) / source->size() : 0.0;
}

// overlap_gpu(targets, source, Ts_target_source): fraction of source points in a voxel of ANY target (:265-335);
// deltas = column-major 4x4 doubles, one per target
inline double overlap_gpu(const std::vector<GaussianVoxelMap::ConstPtr>& targets_, const PointCloud::ConstPtr& source, const std::vector<std::array<double, 16>>& deltas) {
  if (!source->points_gpu) {
    std::cerr << "error: GPU source points have not been allocated!!" << std::endl;  // :270-273
    abort();
  }
  std::vector<const gp_voxelmap_t*> handles(targets_.size());
  for (size_t i = 0; i < targets_.size(); i++) {
    auto t = std::dynamic_pointer_cast<const GaussianVoxelMapGPU>(targets_[i]);
    if (!t) std::cerr << "error: Failed to cast target voxelmap to GaussianVoxelMapGPU!!" << std::endl;  // :278-280 (no abort upstream)
    handles[i] = t ? t->handle() : nullptr;
  }
  int hits = 0;
  check_error << gp_voxelmap_overlap_multi(handles.data(), deltas.empty() ? nullptr : deltas[0].data(), static_cast<int>(handles.size()), source->points_gpu,
                                           static_cast<int>(source->size()), &hits, nullptr);
  return source->size() ? static_cast<double>(hits) / source->size() : 0.0;
}

// overlap_gpu(targets, sources, Ts_target_source) -> one rate per pair (:337-404), ONE launch for all pairs
inline std::vector<double> overlap_gpu(const std::vector<GaussianVoxelMap::ConstPtr>& targets_, const std::vector<PointCloud::ConstPtr>& sources,
                                       const std::vector<std::array<double, 16>>& deltas) {
  if (targets_.size() != sources.size()) {
    std::cerr << "error: The number of target voxelmaps and source point clouds must be the same!!" << std::endl;  // :342-345
    abort();
  }
  const size_t P = sources.size();
  std::vector<const gp_voxelmap_t*> handles(P);
  std::vector<const float*> pts(P);
  std::vector<int> ns(P), hits(P, 0);
  for (size_t i = 0; i < P; i++) {
    auto t = std::dynamic_pointer_cast<const GaussianVoxelMapGPU>(targets_[i]);
    if (!t) std::cerr << "error: Failed to cast target voxelmap to GaussianVoxelMapGPU!!" << std::endl;
    handles[i] = t ? t->handle() : nullptr;
    pts[i] = sources[i]->points_gpu;
    ns[i] = static_cast<int>(sources[i]->size());
  }
  if (P) check_error << gp_voxelmap_overlap_batch(handles.data(), pts.data(), ns.data(), deltas[0].data(), static_cast<int>(P), hits.data(), nullptr);
  std::vector<double> rates(P);
  for (size_t i = 0; i < P; i++) rates[i] = ns[i] ? static_cast<double>(hits[i]) / ns[i] : 0.0;
  return rates;
}

// merge_frames_gpu(poses, frames, downsample_resolution) (:65-152): the merged cloud is the voxel arrays of the down-sampling
// map, copied device to device into a new PointCloudGPU (the reference downloads them and uploads them again)
inline PointCloud::Ptr merge_frames_gpu(const std::vector<std::array<double, 16>>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution,
                                        ihipStream_t* stream = nullptr) {
  const size_t F = frames.size();
  std::vector<const float*> pts(F), covs(F), ints(F);
  std::vector<int> ns(F);
  for (size_t i = 0; i < F; i++) {
    pts[i] = frames[i]->points_gpu;
    covs[i] = frames[i]->covs_gpu;
    ints[i] = frames[i]->intensities_gpu;
    ns[i] = static_cast<int>(frames[i]->size());
  }
  gp_voxelmap_t* map = nullptr;
  check_error << gp_merge_frames(poses[0].data(), pts.data(), covs.data(), ints.data(), ns.data(), static_cast<int>(F), downsample_resolution, 1e-3, stream, &map);
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
  check_error << gp_memcpy_d2d(p, views.voxel_means, 12 * V, stream);
  check_error << gp_memcpy_d2d(c, views.voxel_covs, 36 * V, stream);
  check_error << gp_memcpy_d2d(it, views.voxel_intensities, 4 * V, stream);
  check_error << gp_stream_synchronize(stream);
  merged->adopt(static_cast<float*>(p), static_cast<float*>(c), static_cast<float*>(it), V);
  check_error << gp_voxelmap_destroy(map);
  return merged;
}

}  // namespace gtsam_points
