// point_cloud_gpu.hpp -- PointCloudGPU: the device attribute arrays of a frame in the reference layout
// (types/point_cloud_gpu.hpp:22-143, types/point_cloud_gpu.cu:26-62,110-201,281-370).
// Derives from the reference's own PointCloud and OffloadableGPU.  The reference class additionally IS-A PointCloudCPU
// (host copies of every attribute); this one keeps the packed float arrays it uploaded instead, which is what reload_gpu() needs.
#pragma once
#include <gtsam_points/types/offloadable.hpp>
#include <gtsam_points/types/point_cloud.hpp>

#include <cstdint>
#include <memory>
#include <vector>

#include "check_error.hpp"

namespace gtsam_points {

struct PointCloudGPU : public PointCloud, public OffloadableGPU {
  using Ptr = std::shared_ptr<PointCloudGPU>;
  using ConstPtr = std::shared_ptr<const PointCloudGPU>;

  PointCloudGPU() {}
  template <typename T, int D>
  PointCloudGPU(const Eigen::Matrix<T, D, 1>* points, int num_points) {
    add_points_gpu(points, num_points);
  }
  ~PointCloudGPU() override {
    release(reinterpret_cast<void**>(&points_gpu));
    release(reinterpret_cast<void**>(&normals_gpu));
    release(reinterpret_cast<void**>(&covs_gpu));
    release(reinterpret_cast<void**>(&intensities_gpu));
  }

  std::uint64_t generation = 0;  // bumped whenever the device arrays are re-allocated (factors re-read the pointers)

  // the host array goes up as it lies in memory (Eigen::Matrix<T, D, 1> / <T, D, D>, D in {3, 4}, T float or double) and a pack
  // kernel writes the float3 / 3x3 float device layout -- bit-identical to the reference's host-side cast<float>() (:110-201)
  template <typename T, int D>
  void add_points_gpu(const Eigen::Matrix<T, D, 1>* points, int n, CUstream_st* stream = 0) {
    static_assert(sizeof(Eigen::Matrix<T, D, 1>) == sizeof(T) * D, "dense Eigen storage expected");
    num_points = n;
    upload_packed<T, D>(points, n, 3, reinterpret_cast<float**>(&points_gpu), stream);
    keep(points_host, as_floats(points_gpu), 3, stream);
  }
  template <typename T, int D, typename Alloc>
  void add_points_gpu(const std::vector<Eigen::Matrix<T, D, 1>, Alloc>& points, CUstream_st* stream = 0) {
    add_points_gpu(points.data(), static_cast<int>(points.size()), stream);
  }
  template <typename T, int D>
  void add_normals_gpu(const Eigen::Matrix<T, D, 1>* normals, int n, CUstream_st* stream = 0) {
    upload_packed<T, D>(normals, n, 3, reinterpret_cast<float**>(&normals_gpu), stream);
    keep(normals_host, as_floats(normals_gpu), 3, stream);
  }
  template <typename T, int D, typename Alloc>
  void add_normals_gpu(const std::vector<Eigen::Matrix<T, D, 1>, Alloc>& normals, CUstream_st* stream = 0) {
    add_normals_gpu(normals.data(), static_cast<int>(normals.size()), stream);
  }
  template <typename T, int D>
  void add_covs_gpu(const Eigen::Matrix<T, D, D>* covs, int n, CUstream_st* stream = 0) {
    static_assert(sizeof(Eigen::Matrix<T, D, D>) == sizeof(T) * D * D, "dense Eigen storage expected");
    upload_packed<T, D>(covs, n, 9, reinterpret_cast<float**>(&covs_gpu), stream);
    keep(covs_host, as_floats(covs_gpu), 9, stream);
  }
  template <typename T, int D, typename Alloc>
  void add_covs_gpu(const std::vector<Eigen::Matrix<T, D, D>, Alloc>& covs, CUstream_st* stream = 0) {
    add_covs_gpu(covs.data(), static_cast<int>(covs.size()), stream);
  }
  template <typename T>
  void add_intensities_gpu(const T* intensities, int n, CUstream_st* stream = 0) {
    intensities_host.assign(intensities, intensities + n);
    replace(reinterpret_cast<void**>(&intensities_gpu), sizeof(float) * intensities_host.size());
    check_error << gp_memcpy_h2d(intensities_gpu, intensities_host.data(), sizeof(float) * intensities_host.size(), gp_stream(stream));
    check_error << gp_stream_synchronize(gp_stream(stream));  // stream sync per attribute, as the reference does
    generation++;
  }
  template <typename T>
  void add_intensities_gpu(const std::vector<T>& intensities, CUstream_st* stream = 0) {
    add_intensities_gpu(intensities.data(), static_cast<int>(intensities.size()), stream);
  }

  // ---- OffloadableGPU (types/point_cloud_gpu.cu:281-370) ----
  size_t memory_usage_gpu() const override {
    return ((points_gpu ? 12 : 0) + (normals_gpu ? 12 : 0) + (covs_gpu ? 36 : 0) + (intensities_gpu ? 4 : 0)) * num_points;
  }
  bool loaded_on_gpu() const override { return points_gpu || normals_gpu || covs_gpu || intensities_gpu; }
  bool offload_gpu(CUstream_st* stream = 0) override {
    if (!loaded_on_gpu()) return false;  // nothing to offload (:305-307)
    fetch_if_missing(points_host, as_floats(points_gpu), 3, stream);
    fetch_if_missing(normals_host, as_floats(normals_gpu), 3, stream);
    fetch_if_missing(covs_host, as_floats(covs_gpu), 9, stream);
    fetch_if_missing(intensities_host, intensities_gpu, 1, stream);
    release(reinterpret_cast<void**>(&points_gpu));
    release(reinterpret_cast<void**>(&normals_gpu));
    release(reinterpret_cast<void**>(&covs_gpu));
    release(reinterpret_cast<void**>(&intensities_gpu));
    generation++;
    return true;
  }
  bool reload_gpu(CUstream_st* stream = 0) override {
    if (loaded_on_gpu()) return false;  // :339-341
    bool reloaded = false;
    reloaded |= push(points_host, reinterpret_cast<void**>(&points_gpu), stream);
    reloaded |= push(normals_host, reinterpret_cast<void**>(&normals_gpu), stream);
    reloaded |= push(covs_host, reinterpret_cast<void**>(&covs_gpu), stream);
    reloaded |= push(intensities_host, reinterpret_cast<void**>(&intensities_gpu), stream);
    if (reloaded) {
      check_error << gp_stream_synchronize(gp_stream(stream));
      generation++;
    }
    return reloaded;
  }
  // adopt device arrays that are already in the reference layout (merge_frames_gpu hands its result over this way)
  void adopt(float* points, float* covs, float* intensities, size_t n) {
    release(reinterpret_cast<void**>(&points_gpu));
    release(reinterpret_cast<void**>(&covs_gpu));
    release(reinterpret_cast<void**>(&intensities_gpu));
    points_gpu = reinterpret_cast<Eigen::Vector3f*>(points);
    covs_gpu = reinterpret_cast<Eigen::Matrix3f*>(covs);
    intensities_gpu = intensities;
    num_points = n;
    points_host.clear();
    covs_host.clear();
    intensities_host.clear();
    generation++;
  }

private:
  std::vector<float> points_host, normals_host, covs_host, intensities_host;  // packed device-layout copies for reload_gpu()

  void keep(std::vector<float>& host, const float* dev, int width, CUstream_st* stream) {
    host.resize((size_t)width * num_points);
    check_error << gp_memcpy_d2h(host.data(), dev, sizeof(float) * host.size(), gp_stream(stream));
    check_error << gp_stream_synchronize(gp_stream(stream));
    generation++;
  }
  void fetch_if_missing(std::vector<float>& host, const float* dev, int width, CUstream_st* stream) {
    if (dev && host.empty()) {
      host.resize((size_t)width * num_points);
      check_error << gp_memcpy_d2h(host.data(), dev, sizeof(float) * host.size(), gp_stream(stream));
      check_error << gp_stream_synchronize(gp_stream(stream));
    }
  }
  bool push(const std::vector<float>& host, void** dst, CUstream_st* stream) {
    if (host.empty()) return false;
    replace(dst, sizeof(float) * host.size());
    check_error << gp_memcpy_h2d(*dst, host.data(), sizeof(float) * host.size(), gp_stream(stream));
    return true;
  }
  static void release(void** dst) {
    check_error << gp_free(*dst);
    *dst = nullptr;
  }
  static void replace(void** dst, size_t bytes) {
    release(dst);
    if (bytes) check_error << gp_malloc(dst, bytes);
  }
  template <typename T, int D>
  void upload_packed(const void* src, int n, int width, float** dst, CUstream_st* stream) {
    static_assert(D == 3 || D == 4, "D in {3,4}");
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float or double");
    replace(reinterpret_cast<void**>(dst), sizeof(float) * width * (size_t)n);
    if (width == 9) {
      check_error << gp_cloud_upload_mat3(src, sizeof(T) == 8, D, n, *dst, gp_stream(stream));
    } else {
      check_error << gp_cloud_upload_vec3(src, sizeof(T) == 8, D, n, *dst, gp_stream(stream));
    }
  }
};

// download_*_gpu (types/point_cloud_gpu.hpp:139-142); defined in gtsam_points_hip_host.cpp
std::vector<Eigen::Vector3f> download_points_gpu(const gtsam_points::PointCloud& frame, CUstream_st* stream = nullptr);
std::vector<Eigen::Matrix3f> download_covs_gpu(const gtsam_points::PointCloud& frame, CUstream_st* stream = nullptr);
std::vector<Eigen::Vector3f> download_normals_gpu(const gtsam_points::PointCloud& frame, CUstream_st* stream = nullptr);
std::vector<float> download_intensities_gpu(const gtsam_points::PointCloud& frame, CUstream_st* stream = nullptr);

}  // namespace gtsam_points
