// gaussian_voxelmap_gpu.hpp -- GaussianVoxelMapGPU (types/gaussian_voxelmap_gpu.hpp:20-114) over the C-ABI, deriving from the
// reference's own GaussianVoxelMap and OffloadableGPU.  The free functions the reference DECLARES in types/gaussian_voxelmap.hpp
// (overlap_gpu, 5 overloads) and types/point_cloud_cpu.hpp (merge_frames_gpu) are DEFINED in gtsam_points_hip_host.cpp with the
// reference's signatures.
#pragma once
#include <gtsam_points/types/gaussian_voxelmap.hpp>
#include <gtsam_points/types/offloadable.hpp>

#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "check_error.hpp"
#include "point_cloud_gpu.hpp"

namespace gtsam_points {

struct VoxelMapInfo {  // gaussian_voxelmap_gpu.hpp:20-25 (same layout as gp_voxelmap_info)
  int num_voxels;
  int num_buckets;
  int max_bucket_scan_count;
  float voxel_resolution;
};
static_assert(sizeof(VoxelMapInfo) == sizeof(gp_voxelmap_info), "VoxelMapInfo layout");

struct VoxelBucket {  // :30-33 (same layout as gp_voxel_bucket)
  Eigen::Vector3i first;
  int second;
};
static_assert(sizeof(VoxelBucket) == sizeof(gp_voxel_bucket), "VoxelBucket layout");

class GaussianVoxelMapGPU : public GaussianVoxelMap, public OffloadableGPU {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMapGPU>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMapGPU>;

  GaussianVoxelMapGPU(float resolution, int init_num_buckets = 8192 * 2, int max_bucket_scan_count = 10, double target_points_drop_rate = 1e-3, CUstream_st* stream = 0)
  : stream(stream), init_num_buckets(init_num_buckets), target_points_drop_rate(target_points_drop_rate) {
    check_error << gp_voxelmap_create(resolution, init_num_buckets, max_bucket_scan_count, target_points_drop_rate, gp_stream(stream), &h);
    refresh();
  }
  ~GaussianVoxelMapGPU() override { check_error << gp_voxelmap_destroy(h); }

  double voxel_resolution() const override { return gp_voxelmap_resolution(h); }

  void insert(const PointCloud& frame) override {
    if (!frame.points_gpu || !frame.covs_gpu) {
      std::cerr << "error: GPU points/covs not allocated!!" << std::endl;  // gaussian_voxelmap_gpu.cu:212-215
      abort();
    }
    check_error << gp_voxelmap_insert(h, as_floats(frame.points_gpu), as_floats(frame.covs_gpu), frame.intensities_gpu, static_cast<int>(frame.size()));
    refresh();
  }

  void save_compact(const std::string& path) const override { check_error << gp_voxelmap_save_compact(h, path.c_str()); }

  static GaussianVoxelMapGPU::Ptr load(const std::string& path) {
    gp_voxelmap_t* loaded = nullptr;
    if (gp_voxelmap_load(path.c_str(), nullptr, &loaded) != GP_OK) {
      std::cerr << gp_last_error() << std::endl;
      return nullptr;  // gaussian_voxelmap_gpu.cu:374-377
    }
    return std::shared_ptr<GaussianVoxelMapGPU>(new GaussianVoxelMapGPU(loaded));
  }

  // replica on another device of the node (multi-GPU sharding: a target map referenced from several shards)
  GaussianVoxelMapGPU::Ptr clone_to_device(int device, CUstream_st* stream_on_device = 0) const {
    gp_voxelmap_t* c = nullptr;
    check_error << gp_voxelmap_clone_to_device(h, device, gp_stream(stream_on_device), &c);
    return c ? std::shared_ptr<GaussianVoxelMapGPU>(new GaussianVoxelMapGPU(c)) : nullptr;
  }

  size_t memory_usage_gpu() const override { return gp_voxelmap_memory_usage_gpu(h); }
  bool loaded_on_gpu() const override { return gp_voxelmap_loaded_on_gpu(h) != 0; }
  bool offload_gpu(CUstream_st* s = 0) override {
    const bool ok = gp_voxelmap_offload(h, gp_stream(s)) == GP_OK;
    refresh();
    return ok;
  }
  bool reload_gpu(CUstream_st* s = 0) override {
    if (loaded_on_gpu()) return false;  // gaussian_voxelmap_gpu.cu:509-511
    const bool ok = gp_voxelmap_reload(h, gp_stream(s)) == GP_OK;
    refresh();
    return ok;
  }

  gp_voxelmap_t* handle() const { return h; }

public:
  // the reference's public data members (gaussian_voxelmap_gpu.hpp:86-101), refreshed after every mutating call
  CUstream_st* stream;
  const int init_num_buckets;
  const double target_points_drop_rate;
  VoxelMapInfo voxelmap_info{};
  const VoxelBucket* buckets = nullptr;
  const int* num_points = nullptr;
  const Eigen::Vector3f* voxel_means = nullptr;
  const Eigen::Matrix3f* voxel_covs = nullptr;
  const float* voxel_intensities = nullptr;

private:
  explicit GaussianVoxelMapGPU(gp_voxelmap_t* adopted) : stream(nullptr), init_num_buckets(8192), target_points_drop_rate(0.1), h(adopted) { refresh(); }
  void refresh() {
    gp_voxelmap_info info{};
    check_error << gp_voxelmap_info_get(h, &info);
    voxelmap_info = VoxelMapInfo{info.num_voxels, info.num_buckets, info.max_bucket_scan_count, info.voxel_resolution};
    gp_voxelmap_views v{};
    check_error << gp_voxelmap_views_get(h, &v);
    buckets = reinterpret_cast<const VoxelBucket*>(v.buckets);
    num_points = v.num_points;
    voxel_means = reinterpret_cast<const Eigen::Vector3f*>(v.voxel_means);
    voxel_covs = reinterpret_cast<const Eigen::Matrix3f*>(v.voxel_covs);
    voxel_intensities = v.voxel_intensities;
  }
  gp_voxelmap_t* h = nullptr;
};

// download_* (gaussian_voxelmap_gpu.hpp:110-114); defined in gtsam_points_hip_host.cpp
std::vector<VoxelBucket> download_buckets(const GaussianVoxelMapGPU& voxelmap, CUstream_st* stream = nullptr);
std::vector<int> download_voxel_num_points(const GaussianVoxelMapGPU& voxelmap, CUstream_st* stream = nullptr);
std::vector<Eigen::Vector3f> download_voxel_means(const GaussianVoxelMapGPU& voxelmap, CUstream_st* stream = nullptr);
std::vector<Eigen::Matrix3f> download_voxel_covs(const GaussianVoxelMapGPU& voxelmap, CUstream_st* stream = nullptr);
std::vector<float> download_voxel_intensities(const GaussianVoxelMapGPU& vm, CUstream_st* stream = nullptr);

// merge_frames_gpu (types/point_cloud_cpu.hpp:321-325, gaussian_voxelmap_gpu_funcs.cu:65-152); defined in gtsam_points_hip_host.cpp
PointCloud::Ptr merge_frames_gpu(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution,
                                 CUstream_st* stream = 0);

}  // namespace gtsam_points
