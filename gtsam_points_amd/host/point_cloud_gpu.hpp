// point_cloud_gpu.hpp -- device attribute arrays of a frame, reference layout (types/point_cloud.hpp:103-118,
// types/point_cloud_gpu.cu:26-62,110-201).  Only what the VGICP path reads is mirrored: points/covs/normals/intensities.
#pragma once
#include <gtsam_points_hip.h>

#include <memory>
#include <vector>

#include "check_error.hpp"

namespace gtsam_points {

struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud>;
  using ConstPtr = std::shared_ptr<const PointCloud>;
  virtual ~PointCloud() {}
  size_t size() const { return num_points; }
  bool check_points_gpu() const { return points_gpu != nullptr; }
  bool check_covs_gpu() const { return covs_gpu != nullptr; }

  size_t num_points = 0;
  float* points_gpu = nullptr;       // Eigen::Vector3f[N]
  float* normals_gpu = nullptr;      // Eigen::Vector3f[N]
  float* covs_gpu = nullptr;         // Eigen::Matrix3f[N] (column-major)
  float* intensities_gpu = nullptr;  // float[N]
};

// owning variant: host double/float D in {3,4} -> float3 / 3x3 float staging -> device (add_*_gpu, point_cloud_gpu.cu:110-201)
struct PointCloudGPU : public PointCloud {
  using Ptr = std::shared_ptr<PointCloudGPU>;
  ~PointCloudGPU() override {
    check_error << gp_free(points_gpu);
    check_error << gp_free(normals_gpu);
    check_error << gp_free(covs_gpu);
    check_error << gp_free(intensities_gpu);
  }

  template <typename T, int D>
  void add_points_gpu(const T* points, int n) {
    num_points = n;
    upload3<T, D>(points, n, &points_gpu);
  }
  template <typename T, int D>
  void add_normals_gpu(const T* normals, int n) { upload3<T, D>(normals, n, &normals_gpu); }
  // covs: n matrices of D x D (column-major), D in {3,4}
  template <typename T, int D>
  void add_covs_gpu(const T* covs, int n) {
    std::vector<float> staging(9 * (size_t)n);
    for (int i = 0; i < n; i++)
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) staging[9 * (size_t)i + c * 3 + r] = static_cast<float>(covs[(size_t)i * D * D + c * D + r]);
    upload(staging, &covs_gpu);
  }
  template <typename T>
  void add_intensities_gpu(const T* intensities, int n) {
    std::vector<float> staging(intensities, intensities + n);
    upload(staging, &intensities_gpu);
  }

private:
  template <typename T, int D>
  void upload3(const T* src, int n, float** dst) {
    std::vector<float> staging(3 * (size_t)n);
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) staging[3 * (size_t)i + k] = static_cast<float>(src[(size_t)i * D + k]);
    upload(staging, dst);
  }
  void upload(const std::vector<float>& staging, float** dst) {
    check_error << gp_free(*dst);
    *dst = nullptr;
    void* p = nullptr;
    check_error << gp_malloc(&p, sizeof(float) * staging.size());
    check_error << gp_memcpy_h2d(p, staging.data(), sizeof(float) * staging.size(), nullptr);
    check_error << gp_stream_synchronize(nullptr);  // stream sync per attribute, as the reference does
    *dst = static_cast<float*>(p);
  }
};

}  // namespace gtsam_points
