// gp_lm_poses.hpp -- the pose arithmetic of the device-resident LM trial (gp_lm.hip), shared with the step kernels that run it as their EPILOGUE: the one-launch
// steps (sparse_small_step_kernel, dense_one_pose_step_kernel) end with x in their hands and one compute unit to themselves -- retracting the poses there saves the
// launch of lm_poses_kernel behind them (8.6 us by rocprofv3 on BASELINE configs[2]'s trial, of which the arithmetic is one).  Same functions, same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gtsam_points_hip.h"

namespace gp {

struct Rigid {
  double R[9];  // row-major
  double t[3];
};

__device__ __forceinline__ Rigid load_rigid(const double* __restrict__ p /*column-major 4x4*/) {
  Rigid T;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T.R[r * 3 + c] = p[c * 4 + r];
  T.t[0] = p[12], T.t[1] = p[13], T.t[2] = p[14];
  return T;
}

__device__ __forceinline__ void store_rigid(const Rigid& T, double* __restrict__ p) {
  for (int c = 0; c < 3; c++) {
    for (int r = 0; r < 3; r++) p[c * 4 + r] = T.R[r * 3 + c];
    p[c * 4 + 3] = 0.0;
  }
  p[12] = T.t[0], p[13] = T.t[1], p[14] = T.t[2], p[15] = 1.0;
}

// T * Expmap(xi), xi = (omega, v): gtsam::Pose3::Expmap (R = I + A W + B W^2, t = (I + B W + C W^2) v; series below theta = 1e-8) composed from the right
__device__ Rigid retract_rigid(const Rigid& T, const double* __restrict__ xi) {
  const double wx = xi[0], wy = xi[1], wz = xi[2];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double A, B, C;
  if (th < 1e-8) {
    A = 1.0 - th2 / 6.0, B = 0.5 - th2 / 24.0, C = 1.0 / 6.0 - th2 / 120.0;
  } else {
    const double s = sin(th), c = cos(th);
    A = s / th, B = (1.0 - c) / th2, C = (th - s) / (th2 * th);
  }
  const double W[9] = {0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0};
  double W2[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) W2[r * 3 + c] = W[r * 3] * W[c] + W[r * 3 + 1] * W[3 + c] + W[r * 3 + 2] * W[6 + c];
  double E[9], V[9];
  for (int i = 0; i < 9; i++) {
    const double id = (i % 4 == 0) ? 1.0 : 0.0;
    E[i] = id + A * W[i] + B * W2[i];
    V[i] = id + B * W[i] + C * W2[i];
  }
  const double te[3] = {V[0] * xi[3] + V[1] * xi[4] + V[2] * xi[5], V[3] * xi[3] + V[4] * xi[4] + V[5] * xi[5], V[6] * xi[3] + V[7] * xi[4] + V[8] * xi[5]};
  Rigid out;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out.R[r * 3 + c] = T.R[r * 3] * E[c] + T.R[r * 3 + 1] * E[3 + c] + T.R[r * 3 + 2] * E[6 + c];
    out.t[r] = T.R[r * 3] * te[0] + T.R[r * 3 + 1] * te[1] + T.R[r * 3 + 2] * te[2] + T.t[r];
  }
  return out;
}

// inverse(Tt) * Ts: the relative pose a pairwise factor is evaluated at (integrated_matching_cost_factor.cpp:28-31: delta = target^-1 source)
__device__ __forceinline__ Rigid between_rigid(const Rigid& Tt, const Rigid& Ts) {
  Rigid D;
  const double d[3] = {Ts.t[0] - Tt.t[0], Ts.t[1] - Tt.t[1], Ts.t[2] - Tt.t[2]};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) D.R[r * 3 + c] = Tt.R[r] * Ts.R[c] + Tt.R[3 + r] * Ts.R[3 + c] + Tt.R[6 + r] * Ts.R[6 + c];
    D.t[r] = Tt.R[r] * d[0] + Tt.R[3 + r] * d[1] + Tt.R[6 + r] * d[2];
  }
  return D;
}

struct LmPoseView {
  const double* values;   // [N][16] the current values
  const int* pairs;       // [F][2] (target pose, source pose)
  const int* slot;        // [N] variable slot of a pose, < 0 = held
  const double* x;        // [6 * slots] the step in slot order, or null = no step (relative poses of `values` themselves)
  const int* status;      // the step's status word: != 0 = indeterminate, the trial is the current values
  double* values_out;     // [N][16] device, may be null
  double* values_host;    // [N][16] pinned, may be null
  double* deltas_out;     // [F][16]
  int F, N;
};

// thread i of n >= max(F, N): factor i's relative pose at the (retracted) values, and pose i's (retracted) value.  x: the step in slot order (null: no step -- the relative
// poses of `values` themselves); failed: the step was indeterminate (the trial is the current values).  A pose shared by several factors is retracted by each of
// them from the same operands with the same instructions: the same bits everywhere.
__device__ __forceinline__ void lm_poses_thread(const LmPoseView& v, const int i, const double* __restrict__ x, const bool failed) {
  const bool step = x != nullptr && !failed;
  auto value = [&](int k) {
    Rigid T = load_rigid(v.values + 16 * (size_t)k);
    const int s = v.slot[k];
    return (step && s >= 0) ? retract_rigid(T, x + 6 * (size_t)s) : T;
  };
  if (i < v.F) {
    const Rigid D = between_rigid(value(v.pairs[2 * i]), value(v.pairs[2 * i + 1]));
    store_rigid(D, v.deltas_out + 16 * (size_t)i);
  }
  if (i < v.N && (v.values_out || v.values_host)) {
    const Rigid T = value(i);
    double p[16];
    store_rigid(T, p);
    for (int k = 0; k < 16; k++) {
      if (v.values_out) v.values_out[16 * (size_t)i + k] = p[k];
      if (v.values_host) v.values_host[16 * (size_t)i + k] = p[k];
    }
  }
}

// the damped step with the retract as its epilogue (gp_sparse.hip / gp_solver.hip): as gp_*_system_issue_step; *fused = the step ran as one launch and the epilogue
// with it (else the caller launches lm_poses_kernel behind the step itself)
int sparse_issue_step_with_poses(gp_sparse_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                                 const LmPoseView& poses, bool* fused);
int dense_issue_step_with_poses(gp_dense_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                                const LmPoseView& poses, bool* fused);

}  // namespace gp
