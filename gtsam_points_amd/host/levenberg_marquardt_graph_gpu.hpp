// levenberg_marquardt_graph_gpu.hpp -- the optimizer's trial with the VALUES in device memory (gp_lm_graph_*, csrc/gp_lm.hip), over the C-ABI.
//
// No class of this name exists upstream: LevenbergMarquardtOptimizerExt (optimizers/levenberg_marquardt_ext.cpp) drives its GPU factors through the LinearizationHook --
// linearize(values) in iterate() (:352-392), buildDampedSystem / solve / retract on the host / error(newValues) in tryLambda() (:188-350) -- and that path is served by
// NonlinearFactorSetGPU (nonlinear_factor_set_gpu.hpp).  This class is the SAME cadence for a graph made only of IntegratedVGICPFactorGPU factors with the poses kept on
// the device: linearize() / try_lambda() / accept() are what iterate() and tryLambda() do per call, optimize() is optimize() (:394-430) with GTSAM's
// LevenbergMarquardtParams defaults; one wait per trial instead of two waits, two pose uploads and the host-side pose algebra of every factor.  A maintainer would call it
// from LevenbergMarquardtOptimizerExt when every factor of the graph is one of ours and no other factor touches the poses (else: the hook path, unchanged).
#pragma once
#include <gtsam/geometry/Pose3.h>
#include <gtsam/nonlinear/Values.h>

#include <map>
#include <memory>
#include <set>
#include <vector>

#include "check_error.hpp"
#include "integrated_vgicp_factor_gpu.hpp"

namespace gtsam_points {

class LevenbergMarquardtGraphGPU {
public:
  /// factors: binary factors relate keys()[0] (target) and keys()[1] (source); a unary factor's fixed target pose becomes a held pose of its own.
  /// fixed: keys held at their initial values (the reference pins them with a tight prior); every other key of the factors is a variable.
  LevenbergMarquardtGraphGPU(const std::vector<std::shared_ptr<IntegratedVGICPFactorGPU>>& factors, const std::set<gtsam::Key>& fixed, CUstream_st* stream = nullptr, int ordering = 4)
  : factors(factors) {
    std::vector<gp_vgicp_factor_t*> handles;
    std::vector<int> pairs;
    for (const auto& f : factors) {
      handles.push_back(f->handle());
      if (f->keys().size() == 2) {
        pairs.push_back(index_of(f->keys()[0]));
        pairs.push_back(index_of(f->keys()[1]));
      } else {
        unary_targets.emplace_back((int)unary_targets.size(), gtsam::Pose3(Eigen::Isometry3d(f->get_fixed_target_pose().cast<double>()).matrix()));
        pairs.push_back(-1 - (int)unary_targets.size());  // (patched below, once the number of keyed poses is known)
        pairs.push_back(index_of(f->keys()[0]));
      }
    }
    const int keyed = (int)keys.size();
    for (auto& p : pairs)
      if (p < 0) p = keyed + (-p - 2);
    std::vector<unsigned char> hold((size_t)keyed + unary_targets.size(), 0);
    for (int i = 0; i < keyed; i++) hold[(size_t)i] = fixed.count(keys[(size_t)i]) ? 1 : 0;
    for (size_t i = 0; i < unary_targets.size(); i++) hold[(size_t)keyed + i] = 1;
    check_error << gp_vgicp_batch_create(handles.data(), (int)handles.size(), gp_stream(stream), &batch);
    check_error << gp_lm_graph_create(batch, pairs.data(), (int)hold.size(), hold.data(), ordering, &h);
  }
  ~LevenbergMarquardtGraphGPU() {
    check_error << gp_lm_graph_destroy(h);
    check_error << gp_vgicp_batch_destroy(batch);
  }
  LevenbergMarquardtGraphGPU(const LevenbergMarquardtGraphGPU&) = delete;
  LevenbergMarquardtGraphGPU& operator=(const LevenbergMarquardtGraphGPU&) = delete;

  int dim() const { return gp_lm_graph_num_variables(h); }
  const std::vector<gtsam::Key>& ordered_keys() const { return keys; }  // pose index -> key; the free ones take the variable slots in this order

  void set_values(const gtsam::Values& values) {
    std::vector<double> v(16 * (keys.size() + unary_targets.size()));
    for (size_t i = 0; i < keys.size(); i++) std::memcpy(v.data() + 16 * i, pose16(Eigen::Isometry3d(values.at<gtsam::Pose3>(keys[i]).matrix())).data(), sizeof(double) * 16);
    for (size_t i = 0; i < unary_targets.size(); i++)
      std::memcpy(v.data() + 16 * (keys.size() + i), pose16(Eigen::Isometry3d(unary_targets[i].second.matrix())).data(), sizeof(double) * 16);
    check_error << gp_lm_graph_set_values(h, v.data());
  }
  gtsam::Values values() const {
    std::vector<double> v(16 * (keys.size() + unary_targets.size()));
    check_error << gp_lm_graph_get_values(h, v.data());
    gtsam::Values out;
    for (size_t i = 0; i < keys.size(); i++) {
      Eigen::Matrix4d m;
      std::memcpy(m.data(), v.data() + 16 * i, sizeof(double) * 16);
      out.insert(keys[i], gtsam::Pose3(m));
    }
    return out;
  }
  /// iterate()'s linearisation: asynchronous, the records stay in device memory
  void linearize() { check_error << gp_lm_graph_linearize(h); }
  /// tryLambda()'s device work: damped step, retract, error at the trial values; false = indeterminate system (b and c are valid, nothing to accept)
  /// dx, b: [dim()] in the order of ordered_keys()'s free poses, six entries (omega, v) each
  bool try_lambda(double lambda, std::vector<double>* dx, std::vector<double>* b, double* c, double* new_error) {
    const size_t n = (size_t)dim();
    if (dx) dx->resize(n);
    if (b) b->resize(n);
    const int rc = gp_lm_graph_try_lambda(h, lambda, 0, 1e-6, 1e32, dx ? dx->data() : nullptr, b ? b->data() : nullptr, c, new_error, nullptr);
    if (rc != GP_OK && rc != GP_ERROR_INDETERMINATE) check_error << rc;
    return rc == GP_OK;
  }
  void accept() { check_error << gp_lm_graph_accept(h); }
  /// optimize() (:394-430) from the values set last; GTSAM's LevenbergMarquardtParams defaults unless given
  gp_lm_summary optimize(const gp_lm_params* params = nullptr) {
    gp_lm_summary s{};
    check_error << gp_lm_graph_optimize(h, params, &s);
    return s;
  }

private:
  int index_of(gtsam::Key k) {
    auto it = index.find(k);
    if (it != index.end()) return it->second;
    const int i = (int)keys.size();
    keys.push_back(k);
    index[k] = i;
    return i;
  }
  std::vector<std::shared_ptr<IntegratedVGICPFactorGPU>> factors;  // kept alive: the batch holds their handles
  std::vector<gtsam::Key> keys;
  std::map<gtsam::Key, int> index;
  std::vector<std::pair<int, gtsam::Pose3>> unary_targets;
  gp_vgicp_batch_t* batch = nullptr;
  gp_lm_graph_t* h = nullptr;
};

}  // namespace gtsam_points
