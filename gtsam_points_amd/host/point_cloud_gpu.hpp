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

// owning variant: the host arrays go up as they lie in memory (Eigen::Matrix<T, D, 1> / <T, D, D>, D in {3,4}) and a pack
// kernel writes the float3 / 3x3 float device layout (gp_cloud_upload_*; add_*_gpu, point_cloud_gpu.cu:110-201 converts
// element by element on the host)
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
    upload_packed<T, D>(points, n, 3, &points_gpu);
  }
  template <typename T, int D>
  void add_normals_gpu(const T* normals, int n) { upload_packed<T, D>(normals, n, 3, &normals_gpu); }
  // covs: n matrices of D x D (column-major), D in {3,4}
  template <typename T, int D>
  void add_covs_gpu(const T* covs, int n) { upload_packed<T, D>(covs, n, 9, &covs_gpu); }
  template <typename T>
  void add_intensities_gpu(const T* intensities, int n) {
    std::vector<float> staging(intensities, intensities + n);
    replace(&intensities_gpu, sizeof(float) * staging.size());
    check_error << gp_memcpy_h2d(intensities_gpu, staging.data(), sizeof(float) * staging.size(), nullptr);
    check_error << gp_stream_synchronize(nullptr);  // stream sync per attribute, as the reference does
  }
  // adopt device arrays that are already in the reference layout (merge_frames_gpu hands its result over this way)
  void adopt(float* points, float* covs, float* intensities, size_t n) {
    replace(&points_gpu, 0);
    replace(&covs_gpu, 0);
    replace(&intensities_gpu, 0);
    points_gpu = points;
    covs_gpu = covs;
    intensities_gpu = intensities;
    num_points = n;
  }

private:
  static_assert(sizeof(float) == 4, "device layout is IEEE binary32");
  void replace(float** dst, size_t bytes) {
    check_error << gp_free(*dst);
    *dst = nullptr;
    if (bytes) {
      void* p = nullptr;
      check_error << gp_malloc(&p, bytes);
      *dst = static_cast<float*>(p);
    }
  }
  template <typename T, int D>
  void upload_packed(const T* src, int n, int width, float** dst) {
    static_assert(D == 3 || D == 4, "D in {3,4}");
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float or double");
    replace(dst, sizeof(float) * width * (size_t)n);
    if (width == 9) {
      check_error << gp_cloud_upload_mat3(src, sizeof(T) == 8, D, n, *dst, nullptr);
    } else {
      check_error << gp_cloud_upload_vec3(src, sizeof(T) == 8, D, n, *dst, nullptr);
    }
  }
};

}  // namespace gtsam_points
