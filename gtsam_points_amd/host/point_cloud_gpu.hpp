// point_cloud_gpu.hpp -- PointCloudGPU: the device attribute arrays of a frame in the reference layout
// (types/point_cloud_gpu.hpp:22-143, types/point_cloud_gpu.cu:26-62,110-201,281-370).
// Derives from the reference's own PointCloud and OffloadableGPU.  The reference class IS-A PointCloudCPU; here the host-side storage of
// the add_*() forms (Vector4d points / normals, Matrix4d covariances, double times / intensities: types/point_cloud_cpu.cpp:71-160) lives
// in this class, next to the packed float arrays the add_*_gpu() forms uploaded, which is what reload_gpu() needs.
#pragma once
#include <gtsam_points/types/offloadable.hpp>
#include <gtsam_points/types/point_cloud.hpp>

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "check_error.hpp"

namespace gtsam_points {

struct PointCloudGPU : public PointCloud, public OffloadableGPU {
  using Ptr = std::shared_ptr<PointCloudGPU>;
  using ConstPtr = std::shared_ptr<const PointCloudGPU>;

  PointCloudGPU() {}
  template <typename T, int D>
  PointCloudGPU(const Eigen::Matrix<T, D, 1>* points, int num_points) {
    add_points(points, num_points);  // CPU and GPU storage, as the reference constructor does (types/point_cloud_gpu.cu:18-22)
  }
  template <typename T, int D, typename Alloc>
  PointCloudGPU(const std::vector<Eigen::Matrix<T, D, 1>, Alloc>& points) : PointCloudGPU(points.data(), static_cast<int>(points.size())) {}
  PointCloudGPU(const PointCloudGPU&) = delete;  // forbid shallow copy (:36-37)
  PointCloudGPU& operator=(const PointCloudGPU&) = delete;

  // Deep copy (types/point_cloud_gpu.cu:26-62): host attributes go through the add_*() forms; attributes the source frame holds ONLY on
  // the device (e.g. the result of merge_frames_gpu) are copied device to device -- the reference leaves those out (its TODO at :29)
  static PointCloudGPU::Ptr clone(const PointCloud& frame, CUstream_st* stream = 0) {
    auto out = std::make_shared<PointCloudGPU>();
    const int n = static_cast<int>(frame.size());
    out->num_points = n;
    if (frame.points) out->add_points(frame.points, n, stream);
    if (frame.times) out->add_times(frame.times, n, stream);
    if (frame.normals) out->add_normals(frame.normals, n, stream);
    if (frame.covs) out->add_covs(frame.covs, n, stream);
    if (frame.intensities) out->add_intensities(frame.intensities, n, stream);
    out->copy_device_only(reinterpret_cast<void**>(&out->points_gpu), frame.points_gpu, !frame.points, 12, out->points_host, 3, stream);
    out->copy_device_only(reinterpret_cast<void**>(&out->times_gpu), frame.times_gpu, !frame.times, 4, out->times_host, 1, stream);
    out->copy_device_only(reinterpret_cast<void**>(&out->normals_gpu), frame.normals_gpu, !frame.normals, 12, out->normals_host, 3, stream);
    out->copy_device_only(reinterpret_cast<void**>(&out->covs_gpu), frame.covs_gpu, !frame.covs, 36, out->covs_host, 9, stream);
    out->copy_device_only(reinterpret_cast<void**>(&out->intensities_gpu), frame.intensities_gpu, !frame.intensities, 4, out->intensities_host, 1, stream);
    for (const auto& aux : frame.aux_attributes) {  // (:49-59)
      auto buffer = std::make_shared<std::vector<unsigned char, Eigen::aligned_allocator<unsigned char>>>(aux.second.first * frame.size());
      memcpy(buffer->data(), aux.second.second, aux.second.first * frame.size());
      out->aux_attributes_storage[aux.first] = buffer;
      out->aux_attributes[aux.first] = {aux.second.first, buffer->data()};
    }
    return out;
  }

  // ---- add_*(): CPU storage (PointCloudCPU::add_*, types/point_cloud_cpu.cpp:71-160) AND GPU storage (types/point_cloud_gpu.hpp:44-127) ----
  template <typename T>
  void add_times(const T* t, int n, CUstream_st* stream = 0) {
    times_storage.assign(n, 0.0);
    if (t)
      for (int i = 0; i < n; i++) times_storage[i] = static_cast<double>(t[i]);
    this->times = times_storage.data();
    add_times_gpu(t, n, stream);
  }
  template <typename T>
  void add_times(const std::vector<T>& t, CUstream_st* stream = 0) {
    add_times(t.data(), static_cast<int>(t.size()), stream);
  }
  template <typename T, int D>
  void add_points(const Eigen::Matrix<T, D, 1>* p, int n, CUstream_st* stream = 0) {
    points_storage.assign(n, Eigen::Vector4d(0.0, 0.0, 0.0, 1.0));
    if (p)
      for (int i = 0; i < n; i++)
        for (int r = 0; r < D; r++) points_storage[i](r) = static_cast<double>(p[i](r));
    this->points = points_storage.data();
    this->num_points = n;
    add_points_gpu(p, n, stream);
  }
  template <typename T, int D, typename Alloc>
  void add_points(const std::vector<Eigen::Matrix<T, D, 1>, Alloc>& p, CUstream_st* stream = 0) {
    add_points(p.data(), static_cast<int>(p.size()), stream);
  }
  template <typename T, int D>
  void add_normals(const Eigen::Matrix<T, D, 1>* nrm, int n, CUstream_st* stream = 0) {
    normals_storage.assign(n, Eigen::Vector4d::Zero());
    if (nrm)
      for (int i = 0; i < n; i++)
        for (int r = 0; r < D; r++) normals_storage[i](r) = static_cast<double>(nrm[i](r));
    this->normals = normals_storage.data();
    add_normals_gpu(nrm, n, stream);
  }
  template <typename T, int D, typename Alloc>
  void add_normals(const std::vector<Eigen::Matrix<T, D, 1>, Alloc>& nrm, CUstream_st* stream = 0) {
    add_normals(nrm.data(), static_cast<int>(nrm.size()), stream);
  }
  template <typename T, int D>
  void add_covs(const Eigen::Matrix<T, D, D>* c, int n, CUstream_st* stream = 0) {
    covs_storage.assign(n, Eigen::Matrix4d::Zero());
    if (c)
      for (int i = 0; i < n; i++)
        for (int col = 0; col < D; col++)
          for (int r = 0; r < D; r++) covs_storage[i](r, col) = static_cast<double>(c[i](r, col));
    this->covs = covs_storage.data();
    add_covs_gpu(c, n, stream);
  }
  template <typename T, int D, typename Alloc>
  void add_covs(const std::vector<Eigen::Matrix<T, D, D>, Alloc>& c, CUstream_st* stream = 0) {
    add_covs(c.data(), static_cast<int>(c.size()), stream);
  }
  template <typename T>
  void add_intensities(const T* v, int n, CUstream_st* stream = 0) {
    intensities_storage.assign(n, 0.0);
    if (v)
      for (int i = 0; i < n; i++) intensities_storage[i] = static_cast<double>(v[i]);
    this->intensities = intensities_storage.data();
    add_intensities_gpu(v, n, stream);
  }
  template <typename T>
  void add_intensities(const std::vector<T>& v, CUstream_st* stream = 0) {
    add_intensities(v.data(), static_cast<int>(v.size()), stream);
  }

  // add_times_gpu (types/point_cloud_gpu.cu:88-105): float timestamps on the device; a null source only allocates
  template <typename T>
  void add_times_gpu(const T* t, int n, CUstream_st* stream = 0) {
    times_host.assign((size_t)n, 0.0f);
    if (t)
      for (int i = 0; i < n; i++) times_host[i] = static_cast<float>(t[i]);
    replace(reinterpret_cast<void**>(&times_gpu), sizeof(float) * (size_t)n);
    if (t && n > 0) {
      check_error << gp_memcpy_h2d(times_gpu, times_host.data(), sizeof(float) * (size_t)n, gp_stream(stream));
      check_error << gp_stream_synchronize(gp_stream(stream));
    }
    generation++;
  }
  template <typename T>
  void add_times_gpu(const std::vector<T>& t, CUstream_st* stream = 0) {
    add_times_gpu(t.data(), static_cast<int>(t.size()), stream);
  }

  // download_points (types/point_cloud_gpu.cu:263-279): the device points become the CPU points (Vector4d, w = 1)
  void download_points(CUstream_st* stream = 0) {
    if (!points_gpu) return;
    std::vector<float> tmp(3 * (size_t)num_points);
    check_error << gp_memcpy_d2h(tmp.data(), points_gpu, sizeof(float) * tmp.size(), gp_stream(stream));
    check_error << gp_stream_synchronize(gp_stream(stream));
    points_storage.assign(num_points, Eigen::Vector4d(0.0, 0.0, 0.0, 1.0));
    for (size_t i = 0; i < (size_t)num_points; i++)
      for (int r = 0; r < 3; r++) points_storage[i](r) = static_cast<double>(tmp[3 * i + r]);
    this->points = points_storage.data();
  }
  // every attribute that lives on the device only becomes a CPU attribute as well (points Vector4d w = 1, covs Matrix4d with a zero
  // fourth row / column, intensities double: PointCloudCPU's layouts) and its packed float copy is kept for reload_gpu().
  // merge_frames_gpu returns its result through this: the reference builds that frame with add_points / add_covs / add_intensities
  // (gaussian_voxelmap_gpu_funcs.cu:146-149), so callers read merged->points[i] on the host.
  void download_attributes(CUstream_st* stream = 0) {
    const size_t n = (size_t)num_points;
    if (points_gpu && !this->points) {
      fetch_if_missing(points_host, as_floats(points_gpu), 3, stream);
      points_storage.assign(n, Eigen::Vector4d(0.0, 0.0, 0.0, 1.0));
      for (size_t i = 0; i < n; i++)
        for (int r = 0; r < 3; r++) points_storage[i](r) = static_cast<double>(points_host[3 * i + r]);
      this->points = points_storage.data();
    }
    if (covs_gpu && !this->covs) {
      fetch_if_missing(covs_host, as_floats(covs_gpu), 9, stream);
      covs_storage.assign(n, Eigen::Matrix4d::Zero());
      for (size_t i = 0; i < n; i++)
        for (int col = 0; col < 3; col++)
          for (int r = 0; r < 3; r++) covs_storage[i](r, col) = static_cast<double>(covs_host[9 * i + 3 * col + r]);
      this->covs = covs_storage.data();
    }
    if (normals_gpu && !this->normals) {
      fetch_if_missing(normals_host, as_floats(normals_gpu), 3, stream);
      normals_storage.assign(n, Eigen::Vector4d::Zero());
      for (size_t i = 0; i < n; i++)
        for (int r = 0; r < 3; r++) normals_storage[i](r) = static_cast<double>(normals_host[3 * i + r]);
      this->normals = normals_storage.data();
    }
    if (intensities_gpu && !this->intensities) {
      fetch_if_missing(intensities_host, intensities_gpu, 1, stream);
      intensities_storage.assign(n, 0.0);
      for (size_t i = 0; i < n; i++) intensities_storage[i] = static_cast<double>(intensities_host[i]);
      this->intensities = intensities_storage.data();
    }
    if (times_gpu && !this->times) {
      fetch_if_missing(times_host, times_gpu, 1, stream);
      times_storage.assign(n, 0.0);
      for (size_t i = 0; i < n; i++) times_storage[i] = static_cast<double>(times_host[i]);
      this->times = times_storage.data();
    }
  }
  ~PointCloudGPU() override {
    release(reinterpret_cast<void**>(&times_gpu));
    release(reinterpret_cast<void**>(&points_gpu));
    release(reinterpret_cast<void**>(&normals_gpu));
    release(reinterpret_cast<void**>(&covs_gpu));
    release(reinterpret_cast<void**>(&intensities_gpu));
  }

  std::uint64_t generation = 0;  // bumped whenever the device arrays are re-allocated (factors re-read the pointers)
  std::unordered_map<std::string, std::shared_ptr<void>> aux_attributes_storage;  // owner of aux_attributes' data (types/point_cloud_cpu.hpp)

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
    // a null source allocates (zero-filled) like add_times_gpu; the reference's add_intensities(nullptr, n) reaches here
    intensities_host.assign((size_t)n, 0.0f);
    if (intensities)
      for (int i = 0; i < n; i++) intensities_host[i] = static_cast<float>(intensities[i]);
    replace(reinterpret_cast<void**>(&intensities_gpu), sizeof(float) * intensities_host.size());
    if (n > 0) {
      check_error << gp_memcpy_h2d(intensities_gpu, intensities_host.data(), sizeof(float) * intensities_host.size(), gp_stream(stream));
      check_error << gp_stream_synchronize(gp_stream(stream));  // stream sync per attribute, as the reference does
    }
    generation++;
  }
  template <typename T>
  void add_intensities_gpu(const std::vector<T>& intensities, CUstream_st* stream = 0) {
    add_intensities_gpu(intensities.data(), static_cast<int>(intensities.size()), stream);
  }

  // ---- OffloadableGPU (types/point_cloud_gpu.cu:281-370) ----
  size_t memory_usage_gpu() const override {
    return ((times_gpu ? 4 : 0) + (points_gpu ? 12 : 0) + (normals_gpu ? 12 : 0) + (covs_gpu ? 36 : 0) + (intensities_gpu ? 4 : 0)) * num_points;
  }
  bool loaded_on_gpu() const override { return times_gpu || points_gpu || normals_gpu || covs_gpu || intensities_gpu; }
  bool offload_gpu(CUstream_st* stream = 0) override {
    if (!loaded_on_gpu()) return false;  // nothing to offload (:305-307)
    fetch_if_missing(times_host, times_gpu, 1, stream);
    fetch_if_missing(points_host, as_floats(points_gpu), 3, stream);
    fetch_if_missing(normals_host, as_floats(normals_gpu), 3, stream);
    fetch_if_missing(covs_host, as_floats(covs_gpu), 9, stream);
    fetch_if_missing(intensities_host, intensities_gpu, 1, stream);
    release(reinterpret_cast<void**>(&times_gpu));
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
    reloaded |= push(times_host, reinterpret_cast<void**>(&times_gpu), stream);
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
  std::vector<float> times_host, points_host, normals_host, covs_host, intensities_host;  // packed device-layout copies for reload_gpu()
  // CPU storage of the add_*() forms (PointCloudCPU's members, types/point_cloud_cpu.hpp)
  std::vector<double> times_storage, intensities_storage;
  std::vector<Eigen::Vector4d, Eigen::aligned_allocator<Eigen::Vector4d>> points_storage, normals_storage;
  std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> covs_storage;

  // clone(): an attribute the source frame holds on the device only
  void copy_device_only(void** dst, const void* src_dev, bool host_missing, size_t bytes_per_point, std::vector<float>& host, int width, CUstream_st* stream) {
    if (!src_dev || !host_missing || *dst) return;
    replace(dst, bytes_per_point * (size_t)num_points);
    check_error << gp_memcpy_d2d(*dst, src_dev, bytes_per_point * (size_t)num_points, gp_stream(stream));
    check_error << gp_stream_synchronize(gp_stream(stream));
    host.clear();
    (void)width;
    generation++;
  }

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
