// nonlinear_factor_set_gpu.hpp -- NonlinearFactorSetGPU (cuda/nonlinear_factor_set_gpu.{hpp,cpp}) over the C-ABI, a subclass of the
// reference's own NonlinearFactorSet (optimizers/linearization_hook.hpp:11-29); LinearizationHook itself is the reference's
// (optimizers/linearization_hook.cpp, compiled where it lies).
//   * an all-VGICP set on one device issues ONE batched launch per linearize() / error();
//   * an all-VGICP set whose factors live on several devices is sharded by device (gp_vgicp_multi_batch_*: per-device batched
//     launches + one RCCL all-reduce of the stacked records) -- the reference has no counterpart;
//   * any other NonlinearFactorGPU goes through the reference's staging-buffer protocol with the same byte cursors (:64-218).
#pragma once
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam_points/factors/nonlinear_factor_gpu.hpp>
#include <gtsam_points/optimizers/linearization_hook.hpp>

#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "check_error.hpp"
#include "integrated_vgicp_factor_gpu.hpp"

namespace gtsam_points {

class NonlinearFactorSetGPU : public NonlinearFactorSet {
public:
  NonlinearFactorSetGPU() { check_error << gp_stream_create(&stream); }
  ~NonlinearFactorSetGPU() override {
    drop_batch();
    check_error << gp_free(d_lin_in);
    check_error << gp_free(d_lin_out);
    check_error << gp_free(d_eval_in);
    check_error << gp_free(d_eval_out);
    check_error << gp_stream_destroy(stream);
  }

  int size() const override { return static_cast<int>(factors.size()); }
  void clear() override {
    factors.clear();
    drop_batch();
  }
  void clear_counts() override { num_linearizations = num_evaluations = 0; }
  int linearization_count() const override { return num_linearizations; }
  int evaluation_count() const override { return num_evaluations; }

  bool add(gtsam::NonlinearFactor::shared_ptr factor) override {  // keeps only NonlinearFactorGPU instances (:48-56)
    auto gpu_factor = gtsam_points::dynamic_pointer_cast<NonlinearFactorGPU>(factor);  // nonlinear_factor_set_gpu.cpp:48-56
    if (!gpu_factor) return false;
    factors.push_back(gpu_factor);
    drop_batch();
    return true;
  }
  void add(const gtsam::NonlinearFactorGraph& graph) override {
    for (const auto& f : graph) add(f);
  }

  void linearize(const gtsam::Values& values) override {  // :64-139
    if (factors.empty()) return;
    num_linearizations += size();
    size_t in_size = 0, out_size = 0;
    for (const auto& f : factors) {
      in_size += f->linearization_input_size();
      out_size += f->linearization_output_size();
    }
    lin_in_cpu.resize(in_size);
    lin_out_cpu.resize(out_size);
    size_t cur = 0;
    for (auto& f : factors) {
      f->set_linearization_point(values, lin_in_cpu.data() + cur);
      cur += f->linearization_input_size();
    }
    const unsigned char* results = lin_out_cpu.data();
    if (ensure_batch()) {  // fast path: every factor is an IntegratedVGICPFactorGPU (input = 128-B pose, output = 976-B record)
      if (multi) {
        check_error << gp_vgicp_multi_batch_linearize(multi, reinterpret_cast<const double*>(lin_in_cpu.data()), reinterpret_cast<gp_linearized6*>(lin_out_cpu.data()));
      } else {
        // the records are consumed where the finalize kernel stored them (the batch's pinned buffer): no copy into lin_out_cpu first
        const gp_linearized6* view = nullptr;
        check_error << gp_vgicp_batch_linearize_view(batch, reinterpret_cast<const double*>(lin_in_cpu.data()), &view);
        if (view) results = reinterpret_cast<const unsigned char*>(view);
      }
    } else {
      resize(&d_lin_in, &d_lin_in_size, in_size);
      resize(&d_lin_out, &d_lin_out_size, out_size);
      check_error << gp_memcpy_h2d(d_lin_in, lin_in_cpu.data(), in_size, stream);
      check_error << gp_stream_synchronize(stream);
      size_t ci = 0, co = 0;
      for (auto& f : factors) {
        f->issue_linearize(lin_in_cpu.data() + ci, static_cast<char*>(d_lin_in) + ci, static_cast<char*>(d_lin_out) + co);
        ci += f->linearization_input_size();
        co += f->linearization_output_size();
      }
      for (auto& f : factors) f->sync();
      check_error << gp_memcpy_d2h(lin_out_cpu.data(), d_lin_out, out_size, stream);
      check_error << gp_stream_synchronize(stream);
    }
    cur = 0;
    for (auto& f : factors) {
      f->store_linearized(results + cur);
      cur += f->linearization_output_size();
    }
  }

  void error(const gtsam::Values& values) override {  // :141-218; valid only after linearize() with the same factor order (:180-181)
    if (factors.empty()) return;
    num_evaluations += size();
    size_t in_size = 0, out_size = 0;
    for (const auto& f : factors) {
      in_size += f->evaluation_input_size();
      out_size += f->evaluation_output_size();
    }
    eval_in_cpu.resize(in_size);
    eval_out_cpu.resize(out_size);
    size_t cur = 0;
    for (auto& f : factors) {
      f->set_evaluation_point(values, eval_in_cpu.data() + cur);
      cur += f->evaluation_input_size();
    }
    if (ensure_batch()) {
      if (multi) {
        check_error << gp_vgicp_multi_batch_compute_error(multi, reinterpret_cast<const double*>(lin_in_cpu.data()), reinterpret_cast<const double*>(eval_in_cpu.data()),
                                                          reinterpret_cast<double*>(eval_out_cpu.data()));
      } else {
        check_error << gp_vgicp_batch_compute_error(batch, reinterpret_cast<const double*>(lin_in_cpu.data()), reinterpret_cast<const double*>(eval_in_cpu.data()),
                                                    reinterpret_cast<double*>(eval_out_cpu.data()));
      }
    } else {
      resize(&d_eval_in, &d_eval_in_size, in_size);
      resize(&d_eval_out, &d_eval_out_size, out_size);
      check_error << gp_memcpy_h2d(d_eval_in, eval_in_cpu.data(), in_size, stream);
      check_error << gp_stream_synchronize(stream);
      size_t cl = 0, ci = 0, co = 0;
      for (auto& f : factors) {
        f->issue_compute_error(lin_in_cpu.data() + cl, eval_in_cpu.data() + ci, static_cast<char*>(d_lin_in) + cl, static_cast<char*>(d_eval_in) + ci, static_cast<char*>(d_eval_out) + co);
        cl += f->linearization_input_size();
        ci += f->evaluation_input_size();
        co += f->evaluation_output_size();
      }
      for (auto& f : factors) f->sync();
      check_error << gp_memcpy_d2h(eval_out_cpu.data(), d_eval_out, out_size, stream);
      check_error << gp_stream_synchronize(stream);
    }
    cur = 0;
    for (auto& f : factors) {
      f->store_computed_error(eval_out_cpu.data() + cur);
      cur += f->evaluation_output_size();
    }
  }

  std::vector<gtsam::GaussianFactor::shared_ptr> calc_linear_factors(const gtsam::Values& linearization_point) override {  // :220-228
    linearize(linearization_point);
    std::vector<gtsam::GaussianFactor::shared_ptr> linear_factors(factors.size());
    for (size_t i = 0; i < factors.size(); i++) linear_factors[i] = factors[i]->linearize(linearization_point);
    return linear_factors;
  }

private:
  bool ensure_batch() {
    if (batch || multi) return true;
    if (batch_checked) return false;
    batch_checked = true;
    std::vector<gp_vgicp_factor_t*> handles;
    bool one_device = true;
    for (const auto& f : factors) {
      auto v = gtsam_points::dynamic_pointer_cast<IntegratedVGICPFactorGPU>(f);
      if (!v) return false;
      handles.push_back(v->handle());
      one_device = one_device && gp_vgicp_factor_device(v->handle()) == gp_vgicp_factor_device(handles[0]);
    }
    if (one_device && forced_shards.empty()) {
      check_error << gp_vgicp_batch_create(handles.data(), static_cast<int>(handles.size()), stream, &batch);
      return batch != nullptr;
    }
    // factors on several devices (or an explicit shard assignment: the single-GPU rehearsal of a multi-GPU plan):
    // one shard per device, RCCL all-reduce of the stacked records when the shards sit on distinct devices
    check_error << gp_vgicp_multi_batch_create(handles.data(), static_cast<int>(handles.size()), forced_shards.empty() ? nullptr : forced_shards.data(), forced_num_shards, -1,
                                               &multi);
    return multi != nullptr;
  }
  void drop_batch() {
    if (batch) check_error << gp_vgicp_batch_destroy(batch);
    if (multi) check_error << gp_vgicp_multi_batch_destroy(multi);
    batch = nullptr;
    multi = nullptr;
    batch_checked = false;
  }
  void resize(void** buf, size_t* cap, size_t size) {  // grow-only DeviceBuffer::resize (:20-28)
    if (*cap >= size) return;
    check_error << gp_free(*buf);
    check_error << gp_malloc(buf, size);
    *cap = size;
  }

  gp_stream_t stream = nullptr;
  int num_linearizations = 0, num_evaluations = 0;
  std::vector<NonlinearFactorGPU::shared_ptr> factors;
  std::vector<unsigned char> lin_in_cpu, lin_out_cpu, eval_in_cpu, eval_out_cpu;
  void *d_lin_in = nullptr, *d_lin_out = nullptr, *d_eval_in = nullptr, *d_eval_out = nullptr;
  size_t d_lin_in_size = 0, d_lin_out_size = 0, d_eval_in_size = 0, d_eval_out_size = 0;
  gp_vgicp_batch_t* batch = nullptr;
  gp_vgicp_multi_batch_t* multi = nullptr;
  bool batch_checked = false;
  std::vector<int> forced_shards;
  int forced_num_shards = 0;

public:
  /// rehearsal / tuning: assign the factors (in add() order) to `num_shards` shards explicitly; shards may share a device
  void set_shard_assignment(const std::vector<int>& shard_of_factor, int num_shards) {
    forced_shards = shard_of_factor;
    forced_num_shards = num_shards;
    drop_batch();
  }
  int num_shards() const { return multi ? gp_vgicp_multi_batch_num_shards(multi) : (batch ? 1 : 0); }
  bool uses_rccl() const { return multi && gp_vgicp_multi_batch_uses_rccl(multi) != 0; }
};

std::shared_ptr<NonlinearFactorSet> create_nonlinear_factor_set_gpu();  // cuda/nonlinear_factor_set_gpu_create.hpp:10; defined in gtsam_points_hip_host.cpp

}  // namespace gtsam_points
