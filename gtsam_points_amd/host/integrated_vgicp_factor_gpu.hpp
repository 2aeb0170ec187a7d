// integrated_vgicp_factor_gpu.hpp -- IntegratedVGICPFactorGPU (factors/integrated_vgicp_factor_gpu.{hpp,cpp}) over the C-ABI,
// a subclass of the reference's own NonlinearFactorGPU (factors/nonlinear_factor_gpu.hpp:49-121).
// Same constructors, setters, caching protocol (store_linearized / linearize / error) and abort()-on-precondition behaviour.
#pragma once
#include <gtsam/geometry/Pose3.h>
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam_points/factors/nonlinear_factor_gpu.hpp>
#include <gtsam_points/util/gtsam_migration.hpp>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <optional>

#include "check_error.hpp"
#include "gaussian_voxelmap_gpu.hpp"
#include "stream_temp_buffer_roundrobin.hpp"

namespace gtsam_points {

using LinearizedSystem6 = gp_linearized6;  // cuda/kernels/linearized_system.cuh:10-71, f64

class IntegratedVGICPFactorGPU : public NonlinearFactorGPU {
public:
  using shared_ptr = gtsam_points::shared_ptr<IntegratedVGICPFactorGPU>;

  /// binary factor (integrated_vgicp_factor_gpu.hpp:54-60)
  IntegratedVGICPFactorGPU(gtsam::Key target_key, gtsam::Key source_key, const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source,
                           CUstream_st* stream = nullptr, std::shared_ptr<TempBufferManager> temp_buffer = nullptr)
  : NonlinearFactorGPU(gtsam::KeyVector{target_key, source_key}), is_binary(true), target(std::dynamic_pointer_cast<const GaussianVoxelMapGPU>(target)), source(source), temp_buffer(temp_buffer) {
    init(stream);
  }
  /// unary factor with a fixed target pose (:71-77)
  IntegratedVGICPFactorGPU(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source,
                           CUstream_st* stream = nullptr, std::shared_ptr<TempBufferManager> temp_buffer = nullptr)
  : NonlinearFactorGPU(gtsam::KeyVector{source_key}), is_binary(false), fixed_target_pose(fixed_target_pose), target(std::dynamic_pointer_cast<const GaussianVoxelMapGPU>(target)), source(source), temp_buffer(temp_buffer) {
    init(stream);
  }
  ~IntegratedVGICPFactorGPU() override { check_error << gp_vgicp_factor_destroy(h); }
  IntegratedVGICPFactorGPU(const IntegratedVGICPFactorGPU&) = delete;
  IntegratedVGICPFactorGPU& operator=(const IntegratedVGICPFactorGPU&) = delete;

  void print(const std::string& s = "", const gtsam::KeyFormatter& keyFormatter = gtsam::DefaultKeyFormatter) const override {
    std::cout << s << "IntegratedVGICPFactorGPU";
    if (is_binary) {
      std::cout << "(" << keyFormatter(keys()[0]) << ", " << keyFormatter(keys()[1]) << ")" << std::endl;
    } else {
      std::cout << "(fixed, " << keyFormatter(keys()[0]) << ")" << std::endl;
    }
    std::cout << "target_resolusion=" << target->voxel_resolution() << ", |source|=" << source->size() << "pts" << std::endl;
  }

  size_t memory_usage() const { return sizeof(*this); }
  size_t memory_usage_gpu() const { return sizeof(double) * 16 + sizeof(int); }
  void set_enable_offloading(bool enable) { enable_offloading = enable; }
  void set_enable_surface_validation(bool enable) { check_error << gp_vgicp_factor_set_surface_validation(h, enable ? 1 : 0); }
  void set_inlier_update_thresh(double trans, double angle) { check_error << gp_vgicp_factor_set_inlier_update_thresh(h, trans, angle); }
  int num_inliers() const { return num_inliers_; }
  double inlier_fraction() const { return num_inliers_ / static_cast<double>(source->size()); }
  GaussianVoxelMapGPU::ConstPtr get_target() const { return target; }
  Eigen::Isometry3f get_fixed_target_pose() const { return Eigen::Isometry3d(fixed_target_pose.matrix()).cast<float>(); }  // integrated_vgicp_factor_gpu.hpp:106
  gp_vgicp_factor_t* handle() const { return h; }
  int device() const { return gp_vgicp_factor_device(h); }  // the GPU the operands live on (multi-GPU sharding)

  gtsam::NonlinearFactor::shared_ptr clone() const override {  // drops stream/buffer like the reference (:122-134)
    if (is_binary) return gtsam::make_shared<IntegratedVGICPFactorGPU>(keys()[0], keys()[1], target, source, nullptr, nullptr);
    return gtsam::make_shared<IntegratedVGICPFactorGPU>(fixed_target_pose, keys()[0], target, source, nullptr, nullptr);
  }

  size_t dim() const override { return 6; }

  double error(const gtsam::Values& values) const override {  // :166-183
    if (evaluation_result) {
      const double err = *evaluation_result;
      evaluation_result.reset();
      return err;
    }
    std::cerr << "warning: computing error in sync mode seriously affects the processing speed!!" << std::endl;
    if (!linearized) linearize(values);
    const Eigen::Matrix4d lin = linearization_point.matrix(), eval = calc_delta(values).matrix();
    double err = 0.0;
    check_error << gp_vgicp_factor_compute_error(h, lin.data(), eval.data(), &err);
    return err;
  }

  gtsam::GaussianFactor::shared_ptr linearize(const gtsam::Values& values) const override {  // :185-216
    linearized = true;
    linearization_point = calc_delta(values);
    LinearizedSystem6 l;
    if (linearization_result) {
      l = *linearization_result;
      linearization_result.reset();
    } else {
      std::cerr << "warning: performing linearization in sync mode seriously affects the processing speed!!" << std::endl;
      touch_points();
      const Eigen::Matrix4d lin = linearization_point.matrix();
      check_error << gp_vgicp_factor_linearize(h, lin.data(), &l);
      num_inliers_ = static_cast<int>(l.num_inliers);
    }
    gtsam::Matrix6 Ht, Hs, Hts;
    gtsam::Vector6 bt, bs;
    std::memcpy(Ht.data(), l.H_target, sizeof(double) * 36);
    std::memcpy(Hs.data(), l.H_source, sizeof(double) * 36);
    std::memcpy(Hts.data(), l.H_target_source, sizeof(double) * 36);
    for (int i = 0; i < 6; i++) {
      bt[i] = -l.b_target[i];
      bs[i] = -l.b_source[i];
    }
    if (is_binary) return gtsam::make_shared<gtsam::HessianFactor>(keys()[0], keys()[1], Ht, Hts, bt, Hs, bs, l.error);
    return gtsam::make_shared<gtsam::HessianFactor>(keys()[0], Hs, bs, l.error);
  }

  size_t linearization_input_size() const override { return gp_vgicp_linearization_input_size(); }
  size_t linearization_output_size() const override { return gp_vgicp_linearization_output_size(); }
  size_t evaluation_input_size() const override { return gp_vgicp_evaluation_input_size(); }
  size_t evaluation_output_size() const override { return gp_vgicp_evaluation_output_size(); }

  // IntegratedVGICPDerivatives::touch_points (integrated_vgicp_derivatives.cu:63-78): with offloading enabled both operands are
  // brought back to the GPU; a source cloud whose device arrays moved hands its new pointers to the factor handle
  void touch_points() const {
    if (enable_offloading) {
      const_cast<GaussianVoxelMapGPU*>(target.get())->touch(nullptr);
      if (auto src = dynamic_cast<const OffloadableGPU*>(source.get())) const_cast<OffloadableGPU*>(src)->touch(nullptr);
    }
    if (auto src = dynamic_cast<const PointCloudGPU*>(source.get())) {
      if (src->generation != source_generation) {
        check_error << gp_vgicp_factor_set_source(h, as_floats(source->points_gpu), as_floats(source->covs_gpu), as_floats(source->normals_gpu));
        source_generation = src->generation;
      }
    }
  }

  void set_linearization_point(const gtsam::Values& values, void* lin_input_cpu) override {  // memcpy: no alignment assumed (:219-220)
    touch_points();  // reset_inliers -> touch_points upstream (integrated_vgicp_derivatives_inliers.cu:47)
    const Eigen::Matrix4d d = calc_delta(values).matrix();
    std::memcpy(lin_input_cpu, d.data(), sizeof(double) * 16);
  }
  void set_evaluation_point(const gtsam::Values& values, void* eval_input_cpu) override {
    const Eigen::Matrix4d d = calc_delta(values).matrix();
    std::memcpy(eval_input_cpu, d.data(), sizeof(double) * 16);
  }
  void issue_linearize(const void* lin_input_cpu, const void* lin_input_gpu, void* lin_output_gpu) override {
    double pose[16];
    std::memcpy(pose, lin_input_cpu, sizeof(pose));
    check_error << gp_vgicp_factor_issue_linearize(h, pose, static_cast<const double*>(lin_input_gpu), static_cast<gp_linearized6*>(lin_output_gpu));
  }
  void store_linearized(const void* lin_output_cpu) override {  // :239-245
    linearization_result.reset(new LinearizedSystem6);
    std::memcpy(linearization_result.get(), lin_output_cpu, sizeof(LinearizedSystem6));
    evaluation_result = linearization_result->error;
    num_inliers_ = static_cast<int>(linearization_result->num_inliers);
  }
  void issue_compute_error(const void* lin_input_cpu, const void* eval_input_cpu, const void* lin_input_gpu, const void* eval_input_gpu, void* eval_output_gpu) override {
    double pl[16], pe[16];
    std::memcpy(pl, lin_input_cpu, sizeof(pl));
    std::memcpy(pe, eval_input_cpu, sizeof(pe));
    check_error << gp_vgicp_factor_issue_compute_error(h, pl, pe, static_cast<const double*>(lin_input_gpu), static_cast<const double*>(eval_input_gpu), static_cast<double*>(eval_output_gpu));
  }
  void store_computed_error(const void* eval_output_cpu) override {
    double e;
    std::memcpy(&e, eval_output_cpu, sizeof(double));
    evaluation_result = e;
  }
  void sync() override { check_error << gp_vgicp_factor_sync(h); }

  /// T_target^-1 * T_source in double (integrated_vgicp_factor_gpu.cpp:152-164 keeps it in double up to the final cast; here it stays double)
  gtsam::Pose3 calc_delta(const gtsam::Values& values) const {
    if (!is_binary) return fixed_target_pose.inverse() * values.at<gtsam::Pose3>(keys()[0]);
    return values.at<gtsam::Pose3>(keys()[0]).inverse() * values.at<gtsam::Pose3>(keys()[1]);
  }

private:
  void init(CUstream_st* stream) {
    if (!source->points_gpu) {
      std::cerr << "error: GPU source points have not been allocated!!" << std::endl;  // :33-36
      abort();
    }
    if (!source->covs_gpu) {
      std::cerr << "error: GPU source covs have not been allocated!!" << std::endl;  // :38-41
      abort();
    }
    if (!target) {
      std::cerr << "error: GPU target voxels have not been created!!" << std::endl;  // :43-46
      abort();
    }
    check_error << gp_vgicp_factor_create(target->handle(), as_floats(source->points_gpu), as_floats(source->covs_gpu), as_floats(source->normals_gpu), static_cast<int>(source->size()), gp_stream(stream),
                                          temp_buffer ? temp_buffer->handle() : nullptr, &h);
    if (auto src = dynamic_cast<const PointCloudGPU*>(source.get())) source_generation = src->generation;
  }

  bool is_binary;
  gtsam::Pose3 fixed_target_pose;
  GaussianVoxelMapGPU::ConstPtr target;
  PointCloud::ConstPtr source;
  std::shared_ptr<TempBufferManager> temp_buffer;
  gp_vgicp_factor_t* h = nullptr;
  bool enable_offloading = false;
  mutable std::uint64_t source_generation = 0;
  mutable bool linearized = false;
  mutable gtsam::Pose3 linearization_point;
  mutable int num_inliers_ = 0;
  mutable std::optional<double> evaluation_result;
  mutable std::unique_ptr<LinearizedSystem6> linearization_result;
};

}  // namespace gtsam_points
