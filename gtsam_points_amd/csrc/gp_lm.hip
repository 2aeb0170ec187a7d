// gp_lm.hip -- one Levenberg-Marquardt trial over a graph of pairwise VGICP factors WITHOUT the host in the middle (round 6).
//
// The reference's optimizer drives its GPU factors from the host (optimizers/levenberg_marquardt_ext.cpp): iterate() :352-392 hands `values` to
// linearization_hook_->linearize (cuda/nonlinear_factor_set_gpu.cpp:64-101: poses up, kernels, records down), tryLambda() :188-350 builds the damped system, solves,
// retracts ON THE HOST (:239), hands the new values to linearization_hook_->error (:245, nonlinear_factor_set_gpu.cpp:103-139: poses up, kernels, errors down) and
// decides.  Two waits, two uploads and the pose algebra of every factor per trial -- fine beside a 1.5 s CPU linearise, 0.13 ms of a 0.39 ms iteration here
// (BASELINE configs[2]: 256 factors over 64 poses).
//
// Here the VALUES live in device memory beside the records:
//   gp_lm_graph_linearize    the batch's linearise at the relative poses of the current values (a device table: gp_vgicp_batch_issue_linearize_dev) -> records in HBM
//   gp_lm_graph_try_lambda   the damped step (gp_sparse_system_issue_step / gp_dense_system_issue_step), then the retract of the current values by the step's x
//                            (Pose3::retract = T Expmap(xi), (omega, v) order), which writes the trial values and every factor's relative pose -- as the EPILOGUE of
//                            the step's own kernel where the step is one launch (gp_lm_poses.hpp), else one small kernel behind it (lm_poses_kernel) --, then the batch's error evaluation on the linearisation's correspondences at those (gp_vgicp_batch_issue_compute_error_dev):
//                            launches queued back to back, ONE wait -- a poll of the error evaluation's completion words, not hipStreamSynchronize -- with the
//                            linearise at the trial values already queued behind them (speculation: an accepted step finds it running); x, b, c, the trial's
//                            error and the trial values reach the host through pinned memory
//   gp_lm_graph_accept       the trial values become the current ones (their relative poses are already in place for the next linearise): a pointer swap
//   gp_lm_graph_optimize     the reference's cadence over those three (tryLambda's tests :262-292, decreaseLambda / increaseLambda with the GTSAM defaults)
// Pose arithmetic on the device = the formulas of gtsam::Pose3 (Expmap with the closed-form V, compose, inverse() * other) in f64; the host-side harness
// (bench_lm.py) computes the same with numpy, and the two agree to rounding (tests/test_lm_gpu.py).
// Device: the graph is created on the device that is current (the batch's: gp_set_device first in a multi-device process), like the batch and the systems it builds.
// Not thread-safe per handle, re-entrant across handles, like the rest of the library.
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "gp_host.hpp"
#include "gp_lm_poses.hpp"
#include "gp_vgicp_shared.hpp"

namespace gp {

__global__ void __launch_bounds__(256) lm_poses_kernel(const LmPoseView v) {
  lm_poses_thread(v, blockIdx.x * 256 + threadIdx.x, v.x, v.x != nullptr && *v.status != 0);
}

}  // namespace gp

struct gp_lm_graph {
  gp_vgicp_batch_t* batch = nullptr;  // not owned
  gp_sparse_system_t* sparse = nullptr;
  gp_dense_system_t* dense = nullptr;
  hipStream_t stream = nullptr;
  int F = 0, N = 0, slots = 0;
  std::vector<int> slot;
  gp::DeviceArray d_pairs, d_slot, d_values[2], d_deltas[2], d_records[2];
  gp::PinnedArray h_values[2];
  int cur = 0;                 // d_values[cur] / d_deltas[cur] / h_values[cur] are the current values; [1 - cur] the last trial's
  bool have_values = false, linearized = false, tried = false;
  bool rigid = true;           // every value handed to set_values was orthonormal to 1e-9 (retracts keep them so): the rigid kernels, as the host-pose entry points would choose
  // speculation: behind a trial's error evaluation the linearise AT THE TRIAL VALUES is queued into the other record buffer, so that an accepted step -- the common
  // case -- finds its linearisation already running while the host still decides; a rejected one leaves the records of the linearisation point untouched for the next
  // lambda (the queued linearise is 50 us of device time thrown away).  Same kernels on the same operands either way: results do not depend on it.
  int rec = 0;                 // d_records[rec]: the records of the current linearisation point
  bool speculate = true, spec_pending = false /* [1 - rec] is being written at the last trial's values */, spec_valid = false /* [rec] already holds the current values' records */;
  std::vector<double> errors;
  const double* x_dev = nullptr;
  const int* status_dev = nullptr;
};

namespace {

gp::LmPoseView pose_view(gp_lm_graph* g, int from, int to, bool step) {
  gp::LmPoseView v{};
  v.values = g->d_values[from].as<double>();
  v.pairs = g->d_pairs.as<int>();
  v.slot = g->d_slot.as<int>();
  v.x = step ? g->x_dev : nullptr;
  v.status = g->status_dev;
  v.values_out = step ? g->d_values[to].as<double>() : nullptr;
  v.values_host = step ? g->h_values[to].as<double>() : nullptr;
  v.deltas_out = g->d_deltas[to].as<double>();
  v.F = g->F, v.N = g->N;
  return v;
}

int launch_poses(gp_lm_graph* g, int from, int to, bool step) {
  const gp::LmPoseView v = pose_view(g, from, to, step);
  const int n = std::max(g->F, g->N);
  hipLaunchKernelGGL(gp::lm_poses_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, v);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

}  // namespace

extern "C" {

int gp_lm_graph_destroy(gp_lm_graph_t* g) {
  if (!g) return GP_OK;
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->sparse) gp_sparse_system_destroy(g->sparse);
  if (g->dense) gp_dense_system_destroy(g->dense);
  delete g;
  return GP_OK;
}

int gp_lm_graph_create(gp_vgicp_batch_t* batch, const int* pose_pairs, int num_poses, const unsigned char* pose_fixed, int ordering, gp_lm_graph_t** out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_create: null out");
  *out = nullptr;
  const int F = batch ? gp_vgicp_batch_size(batch) : 0;
  if (!batch || F <= 0 || !pose_pairs || num_poses < 2) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_create: a batch of >= 1 factors, pose_pairs [F][2], >= 2 poses");
  auto* g = new gp_lm_graph;
  g->batch = batch, g->F = F, g->N = num_poses;
  g->slot.assign((size_t)num_poses, -1);
  for (int i = 0; i < num_poses; i++)
    if (!pose_fixed || !pose_fixed[i]) g->slot[i] = g->slots++;
  std::vector<int> factor_slots(2 * (size_t)F);
  int rc = GP_OK;
  for (int f = 0; f < F && rc == GP_OK; f++) {
    const int t = pose_pairs[2 * f], s = pose_pairs[2 * f + 1];
    if (t < 0 || t >= num_poses || s < 0 || s >= num_poses || t == s) rc = gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_create: pose_pairs entries must be two different poses in [0, num_poses)");
    else factor_slots[2 * f] = g->slot[t], factor_slots[2 * f + 1] = g->slot[s];
  }
  if (rc == GP_OK && g->slots == 0) rc = gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_create: every pose is held");
  gp_stream_t st = nullptr;
  if (rc == GP_OK) rc = gp_vgicp_batch_stream(batch, &st);
  g->stream = (hipStream_t)st;
  // one free pose: the dense step (a 6 x 6 system); else the block-sparse one
  if (rc == GP_OK) rc = g->slots == 1 ? gp_dense_system_create(1, factor_slots.data(), F, st, &g->dense) : gp_sparse_system_create(g->slots, factor_slots.data(), F, ordering, st, &g->sparse);
  if (rc == GP_OK) rc = g->sparse ? gp_sparse_system_device_solution(g->sparse, &g->x_dev, &g->status_dev) : gp_dense_system_device_solution(g->dense, &g->x_dev, &g->status_dev);
  const size_t vb = sizeof(double) * 16 * (size_t)num_poses, db = sizeof(double) * 16 * (size_t)F;
  for (int k = 0; k < 2 && rc == GP_OK; k++) {
    if ((rc = g->d_values[k].alloc(vb)) || (rc = g->d_deltas[k].alloc(db)) || (rc = g->h_values[k].ensure(vb))) break;
  }
  if (rc == GP_OK) rc = g->d_pairs.alloc(sizeof(int) * 2 * (size_t)F);
  if (rc == GP_OK) rc = g->d_slot.alloc(sizeof(int) * (size_t)num_poses);
  for (int k = 0; k < 2 && rc == GP_OK; k++) rc = g->d_records[k].alloc(sizeof(gp_linearized6) * (size_t)F);
  g->errors.assign((size_t)F, 0.0);
  if (rc == GP_OK && hipMemcpy(g->d_pairs.ptr, pose_pairs, sizeof(int) * 2 * (size_t)F, hipMemcpyHostToDevice) != hipSuccess) rc = gp::fail(GP_ERROR_HIP, "gp_lm_graph_create: upload of the pairs");
  if (rc == GP_OK && hipMemcpy(g->d_slot.ptr, g->slot.data(), sizeof(int) * (size_t)num_poses, hipMemcpyHostToDevice) != hipSuccess) rc = gp::fail(GP_ERROR_HIP, "gp_lm_graph_create: upload of the slots");
  if (rc != GP_OK) {
    gp_lm_graph_destroy(g);
    return rc;
  }
  *out = g;
  return GP_OK;
}

// 0: no linearise is queued ahead of the host's decision (measurement / A-B; results are the same bits either way).  Returns the previous setting.
int gp_lm_graph_set_speculation(gp_lm_graph_t* g, int enable) {
  if (!g) return 0;
  const int was = g->speculate ? 1 : 0;
  g->speculate = enable != 0;
  return was;
}

// the graph's own damped system: 0 = its multi-launch step (the retract then runs as lm_poses_kernel behind it), 1 = the one-launch step where the system qualifies
// (default; the retract is its epilogue).  Bit-identical; for the test that says so and A/B timing.  Returns what the next trial runs (as gp_*_system_set_one_launch).
int gp_lm_graph_set_one_launch(gp_lm_graph_t* g, int enable) {
  if (!g) return 0;
  return g->sparse ? gp_sparse_system_set_one_launch(g->sparse, enable) : gp_dense_system_set_one_launch(g->dense, enable);
}

int gp_lm_graph_num_variables(const gp_lm_graph_t* g) { return g ? 6 * g->slots : 0; }

int gp_lm_graph_set_values(gp_lm_graph_t* g, const double* values_host) {
  if (!g || !values_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_set_values: null");
  GP_HIP(hipStreamSynchronize(g->stream));  // (a trial in flight still writes the pinned values)
  g->rigid = true;
  for (int i = 0; i < g->N; i++) g->rigid = g->rigid && gp::pose_is_rigid(values_host + 16 * (size_t)i);
  memcpy(g->h_values[g->cur].ptr, values_host, sizeof(double) * 16 * (size_t)g->N);
  GP_HIP(hipMemcpyAsync(g->d_values[g->cur].ptr, g->h_values[g->cur].ptr, sizeof(double) * 16 * (size_t)g->N, hipMemcpyHostToDevice, g->stream));
  GP_TRY(launch_poses(g, g->cur, g->cur, false));
  g->have_values = true, g->linearized = false, g->tried = false, g->spec_pending = false, g->spec_valid = false;
  return GP_OK;
}

int gp_lm_graph_get_values(gp_lm_graph_t* g, double* values_host) {
  if (!g || !values_host || !g->have_values) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_get_values: no values");
  memcpy(values_host, g->h_values[g->cur].ptr, sizeof(double) * 16 * (size_t)g->N);
  return GP_OK;
}

int gp_lm_graph_linearize(gp_lm_graph_t* g) {
  if (!g || !g->have_values) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_linearize: set the values first");
  if (!g->spec_valid)  // (else: queued behind the accepted trial's error evaluation)
    GP_TRY(gp_vgicp_batch_issue_linearize_dev(g->batch, g->d_deltas[g->cur].as<double>(), g->rigid ? 1 : 0, g->d_records[g->rec].as<gp_linearized6>()));
  g->spec_valid = false;
  g->linearized = true, g->tried = false;
  return GP_OK;
}

int gp_lm_graph_records(gp_lm_graph_t* g, const gp_linearized6** records_dev, const double** relative_poses_dev) {
  if (!g) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_records: null graph");
  if (records_dev) *records_dev = g->d_records[g->rec].as<gp_linearized6>();
  if (relative_poses_dev) *relative_poses_dev = g->d_deltas[g->cur].as<double>();
  return GP_OK;
}

int gp_lm_graph_try_lambda(gp_lm_graph_t* g, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal, double* x_host, double* b_host, double* c_host,
                           double* new_error, double* new_values_host) {
  if (!g || !g->linearized) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_try_lambda: linearize first");
  const gp_linearized6* rec = g->d_records[g->rec].as<gp_linearized6>();
  const int to = 1 - g->cur;
  g->spec_pending = false, g->tried = false;
  // (the one-launch steps retract the poses themselves, as their epilogue: gp_lm_poses.hpp; behind a multi-launch step lm_poses_kernel does)
  bool fused = false;
  const gp::LmPoseView pv = pose_view(g, g->cur, to, true);
  if (g->sparse) GP_TRY(gp::sparse_issue_step_with_poses(g->sparse, rec, lambda, diagonal_damping, min_diagonal, max_diagonal, pv, &fused));
  else GP_TRY(gp::dense_issue_step_with_poses(g->dense, rec, lambda, diagonal_damping, min_diagonal, max_diagonal, pv, &fused));
  int rc = fused ? GP_OK : launch_poses(g, g->cur, to, true);
  if (rc == GP_OK) rc = gp_vgicp_batch_issue_compute_error_dev_begin(g->batch, g->d_deltas[g->cur].as<double>(), g->d_deltas[to].as<double>());
  if (rc != GP_OK) {  // the step went out: collect it (its own wait) before the error is reported
    if (g->sparse) (void)gp_sparse_system_finish_step(g->sparse, nullptr, nullptr, nullptr);
    else (void)gp_dense_system_finish_step(g->dense, nullptr, nullptr, nullptr);
    return rc;
  }
  bool spec = false;
  if (g->speculate) spec = gp_vgicp_batch_issue_linearize_dev(g->batch, g->d_deltas[to].as<double>(), g->rigid ? 1 : 0, g->d_records[1 - g->rec].as<gp_linearized6>()) == GP_OK;
  // the call's ONE wait: the completion words of the error evaluation (everything in front of it on the stream -- the step, the trial values -- is then complete and
  // its pinned results are readable; the speculative linearise behind it is not waited for)
  rc = gp_vgicp_batch_compute_error_dev_end(g->batch, g->errors.data());
  const int rs = g->sparse ? (rc == GP_OK ? gp_sparse_system_collect_step(g->sparse, x_host, b_host, c_host) : gp_sparse_system_finish_step(g->sparse, x_host, b_host, c_host))
                           : (rc == GP_OK ? gp_dense_system_collect_step(g->dense, x_host, b_host, c_host) : gp_dense_system_finish_step(g->dense, x_host, b_host, c_host));
  if (rc != GP_OK) return rc;
  if (rs != GP_OK) return rs;  // GP_ERROR_INDETERMINATE: b, c valid; no trial (the kernel behind the step left the trial values = the current ones)
  g->tried = true;
  g->spec_pending = spec;
  if (new_error) {
    double e = 0.0;
    for (int f = 0; f < g->F; f++) e += g->errors[(size_t)f];
    *new_error = e;
  }
  if (new_values_host) memcpy(new_values_host, g->h_values[to].ptr, sizeof(double) * 16 * (size_t)g->N);
  return GP_OK;
}

int gp_lm_graph_accept(gp_lm_graph_t* g) {
  if (!g || !g->tried) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_accept: no successful trial to accept");
  g->cur = 1 - g->cur;
  if (g->spec_pending) g->rec = 1 - g->rec, g->spec_valid = true;
  g->spec_pending = false;
  g->tried = false, g->linearized = false;
  return GP_OK;
}

int gp_lm_graph_optimize(gp_lm_graph_t* g, const gp_lm_params* params, gp_lm_summary* summary) {
  if (!g || !g->have_values || !summary) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_optimize: set the values first; summary must not be null");
  gp_lm_params p;
  gp_lm_params_default(&p);
  if (params) p = *params;
  if (!(p.lambda_initial > 0.0) || !(p.lambda_factor > 1.0) || p.max_iterations < 0 || p.diagonal_damping)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_lm_graph_optimize: lambda_initial > 0, lambda_factor > 1, lambda I damping only (drive gp_lm_graph_try_lambda yourself for the diagonal form)");
  const size_t n = 6 * (size_t)g->slots;
  std::vector<double> x(n), b(n);
  double lam = p.lambda_initial, err = std::numeric_limits<double>::quiet_NaN();
  *summary = gp_lm_summary{};
  bool stop = false;
  for (int it = 0; it < p.max_iterations && !stop; it++) {
    summary->iterations++;
    GP_TRY(gp_lm_graph_linearize(g));
    for (;;) {  // tryLambda until a step is taken or the search ends (levenberg_marquardt_ext.cpp:188-350)
      summary->inner_iterations++;
      double c = 0.0, new_err = 0.0;
      const int rc = gp_lm_graph_try_lambda(g, lam, p.diagonal_damping, p.min_diagonal, p.max_diagonal, x.data(), b.data(), &c, &new_err, nullptr);
      if (rc != GP_OK && rc != GP_ERROR_INDETERMINATE) return rc;
      err = c;  // the cost at the linearisation point comes back with the step, solved or not (the GPU factor keeps the linearise's error)
      bool accepted = false;
      if (rc == GP_OK) {
        double bx = 0.0, xx = 0.0;
        for (size_t i = 0; i < n; i++) bx += b[i] * x[i], xx += x[i] * x[i];
        const double lin_change = 0.5 * bx + 0.5 * lam * xx;  // old - new linearised error of (A + lambda I) dx = b  (:226-230)
        if (lin_change >= 0.0) {
          const double change = err - new_err;
          accepted = lin_change > std::numeric_limits<double>::epsilon() * err && change / lin_change > p.min_model_fidelity;  // (:262-268)
          if (std::fabs(change) < p.relative_error_tol * err) stop = true;                                                     // (:271-278)
        }
        if (accepted) {
          const double prev = err;
          GP_TRY(gp_lm_graph_accept(g));
          err = new_err;
          lam /= p.lambda_factor;
          if (lam < p.lambda_lower_bound) lam = p.lambda_lower_bound;
          if (std::fabs(prev - err) < p.absolute_error_tol || std::fabs(prev - err) / std::max(prev, 1e-300) < p.relative_error_tol) stop = true;  // optimize() :394-430
          break;
        }
      }
      if (stop) break;
      lam *= p.lambda_factor;
      if (lam >= p.lambda_upper_bound) {
        stop = true;
        summary->gave_up = 1;
        break;
      }
    }
  }
  summary->final_error = err;
  summary->final_lambda = lam;
  return GP_OK;
}

void gp_lm_params_default(gp_lm_params* p) {
  if (!p) return;
  p->lambda_initial = 1e-5, p->lambda_factor = 10.0, p->lambda_upper_bound = 1e5, p->lambda_lower_bound = 0.0;
  p->relative_error_tol = 1e-5, p->absolute_error_tol = 1e-5, p->min_model_fidelity = 1e-3;
  p->min_diagonal = 1e-6, p->max_diagonal = 1e32;
  p->max_iterations = 100, p->diagonal_damping = 0;
}

}  // extern "C"
