// nonlinear_factor_gpu.hpp -- NonlinearFactorGPU, the lower half of the plugin seam (factors/nonlinear_factor_gpu.hpp:49-121).
// Interface unchanged; third-party subclasses keep working through NonlinearFactorSetGPU's generic path.
#pragma once
#include "gtsam_stub.hpp"

namespace gtsam_points {

class NonlinearFactorGPU : public gtsam::NonlinearFactor {
public:
  using shared_ptr = gtsam_points::shared_ptr<NonlinearFactorGPU>;
  template <typename CONTAINER>
  explicit NonlinearFactorGPU(const CONTAINER& keys) : gtsam::NonlinearFactor(keys) {}
  ~NonlinearFactorGPU() override {}

  virtual size_t linearization_input_size() const = 0;
  virtual size_t linearization_output_size() const = 0;
  virtual size_t evaluation_input_size() const = 0;
  virtual size_t evaluation_output_size() const = 0;
  virtual void set_linearization_point(const gtsam::Values& values, void* lin_input_cpu) = 0;
  virtual void issue_linearize(const void* lin_input_cpu, const void* lin_input_gpu, void* lin_output_gpu) = 0;
  virtual void store_linearized(const void* lin_output_cpu) = 0;
  virtual void set_evaluation_point(const gtsam::Values& values, void* eval_input_cpu) = 0;
  virtual void issue_compute_error(const void* lin_input_cpu, const void* eval_input_cpu, const void* lin_input_gpu, const void* eval_input_gpu, void* eval_output_gpu) = 0;
  virtual void store_computed_error(const void* eval_output_cpu) = 0;
  virtual void sync() = 0;
};

}  // namespace gtsam_points
