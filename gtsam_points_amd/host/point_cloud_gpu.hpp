// point_cloud_gpu.hpp -- device attribute arrays of a frame, reference layout (types/point_cloud.hpp:103-118,
// types/point_cloud_gpu.cu:26-62,110-201).  Only what the VGICP path reads is mirrored: points/covs/normals/intensities.
#pragma once
#include <gtsam_points_hip.h>

#include <memory>
#include <vector>

#include <atomic>
#include <cstdint>

#include "check_error.hpp"

struct ihipStream_t;

namespace gtsam_points {

// types/offloadable.hpp:17-63, offloadable.cpp: a global access counter; touch() = remember the access + make sure the data is
// on the GPU.  Applications sort by last_accessed_time() to decide what to offload.
class OffloadableGPU {
public:
  OffloadableGPU() : last_access(counter().load()) {}
  virtual ~OffloadableGPU() {}
  static std::uint64_t current_access_time() { return counter().load(); }
  std::uint64_t last_accessed_time() const { return last_access; }
  virtual bool touch(ihipStream_t* stream = nullptr) {
    last_access = counter()++;
    return reload_gpu(stream);
  }
  virtual size_t memory_usage_gpu() const = 0;
  virtual bool loaded_on_gpu() const = 0;
  virtual bool offload_gpu(ihipStream_t* stream = nullptr) = 0;
  virtual bool reload_gpu(ihipStream_t* stream = nullptr) = 0;

private:
  static std::atomic_uint64_t& counter() {
    static std::atomic_uint64_t c{0};
    return c;
  }
  std::uint64_t last_access;
};

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
struct PointCloudGPU : public PointCloud, public OffloadableGPU {
  using Ptr = std::shared_ptr<PointCloudGPU>;
  std::uint64_t generation = 0;  // bumped whenever the device arrays are re-allocated (factors re-read the pointers)
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
    keep(points_host, points_gpu, 3);
  }
  template <typename T, int D>
  void add_normals_gpu(const T* normals, int n) {
    upload_packed<T, D>(normals, n, 3, &normals_gpu);
    keep(normals_host, normals_gpu, 3);
  }
  // covs: n matrices of D x D (column-major), D in {3,4}
  template <typename T, int D>
  void add_covs_gpu(const T* covs, int n) {
    upload_packed<T, D>(covs, n, 9, &covs_gpu);
    keep(covs_host, covs_gpu, 9);
  }
  template <typename T>
  void add_intensities_gpu(const T* intensities, int n) {
    intensities_host.assign(intensities, intensities + n);
    replace(&intensities_gpu, sizeof(float) * intensities_host.size());
    check_error << gp_memcpy_h2d(intensities_gpu, intensities_host.data(), sizeof(float) * intensities_host.size(), nullptr);
    check_error << gp_stream_synchronize(nullptr);  // stream sync per attribute, as the reference does
  }

  // ---- OffloadableGPU (types/point_cloud_gpu.cu:281-370).  The reference class is a PointCloudCPU with device mirrors and
  // reloads from its host attributes; this one keeps the packed float arrays it uploaded (the device layout) for that purpose.
  size_t memory_usage_gpu() const override {
    return (points_gpu ? 12 : 0) * num_points + (normals_gpu ? 12 : 0) * num_points + (covs_gpu ? 36 : 0) * num_points + (intensities_gpu ? 4 : 0) * num_points;
  }
  bool loaded_on_gpu() const override { return points_gpu || normals_gpu || covs_gpu || intensities_gpu; }
  bool offload_gpu(ihipStream_t* = nullptr) override {
    if (!loaded_on_gpu()) return false;  // nothing to offload (:305-307)
    fetch_if_missing(points_host, points_gpu, 3);
    fetch_if_missing(normals_host, normals_gpu, 3);
    fetch_if_missing(covs_host, covs_gpu, 9);
    fetch_if_missing(intensities_host, intensities_gpu, 1);
    replace(&points_gpu, 0);
    replace(&normals_gpu, 0);
    replace(&covs_gpu, 0);
    replace(&intensities_gpu, 0);
    generation++;
    return true;
  }
  bool reload_gpu(ihipStream_t* = nullptr) override {
    if (loaded_on_gpu()) return false;  // :339-341
    bool reloaded = false;
    reloaded |= push(points_host, &points_gpu);
    reloaded |= push(normals_host, &normals_gpu);
    reloaded |= push(covs_host, &covs_gpu);
    reloaded |= push(intensities_host, &intensities_gpu);
    if (reloaded) {
      check_error << gp_stream_synchronize(nullptr);
      generation++;
    }
    return reloaded;
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
    points_host.clear();
    covs_host.clear();
    intensities_host.clear();
    generation++;
  }

private:
  std::vector<float> points_host, normals_host, covs_host, intensities_host;  // packed device-layout copies for reload_gpu()
  static_assert(sizeof(float) == 4, "device layout is IEEE binary32");
  void keep(std::vector<float>& host, const float* dev, int width) {
    host.resize((size_t)width * num_points);
    check_error << gp_memcpy_d2h(host.data(), dev, sizeof(float) * host.size(), nullptr);
    check_error << gp_stream_synchronize(nullptr);
    generation++;
  }
  void fetch_if_missing(std::vector<float>& host, const float* dev, int width) {
    if (dev && host.empty()) {
      host.resize((size_t)width * num_points);
      check_error << gp_memcpy_d2h(host.data(), dev, sizeof(float) * host.size(), nullptr);
      check_error << gp_stream_synchronize(nullptr);
    }
  }
  bool push(const std::vector<float>& host, float** dst) {
    if (host.empty()) return false;
    replace(dst, sizeof(float) * host.size());
    check_error << gp_memcpy_h2d(*dst, host.data(), sizeof(float) * host.size(), nullptr);
    return true;
  }
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
