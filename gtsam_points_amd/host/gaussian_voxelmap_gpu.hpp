// gaussian_voxelmap_gpu.hpp -- GaussianVoxelMapGPU (types/gaussian_voxelmap_gpu.hpp:38-114) over the C-ABI.
#pragma once
#include <gtsam_points_hip.h>

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

class GaussianVoxelMapGPU : public GaussianVoxelMap {
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

  size_t memory_usage_gpu() const { return gp_voxelmap_memory_usage_gpu(h); }
  bool loaded_on_gpu() const { return gp_voxelmap_loaded_on_gpu(h) != 0; }
  bool offload_gpu(ihipStream_t* s = nullptr) {
    const bool ok = gp_voxelmap_offload(h, s) == GP_OK;
    refresh();
    return ok;
  }
  bool reload_gpu(ihipStream_t* s = nullptr) {
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

}  // namespace gtsam_points
