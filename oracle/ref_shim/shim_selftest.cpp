// shim_selftest.cpp -- TEST INFRASTRUCTURE.  C entry points straight onto the STAND-IN Eigen / GTSAM headers in ./include, so that their arithmetic can be held
// against an independent implementation (numpy) on its own: oracle/_ref/libref.so compiles the reference's control flow on top of these stand-ins, and every
// Matrix::inverse(), SelfAdjointEigenSolver, Pose3::inverse / compose / matrix, SO3::Hat, Isometry3d::inverse inside it resolves here.
// Call sites they stand in for: include/gtsam_points/factors/impl/integrated_vgicp_factor_impl.hpp:138-140,233-237;
// src/gtsam_points/features/covariance_estimation.cpp:49-53; src/gtsam_points/factors/integrated_matching_cost_factor.cpp:59-66.
// Needs no reference source: built by oracle/Makefile into oracle/libshimtest.so wherever the repository is.
#include <Eigen/Core>
#include <Eigen/Eigenvalues>
#include <Eigen/Geometry>
#include <gtsam/geometry/Pose3.h>

extern "C" {

// all matrices row-major on this boundary
__attribute__((visibility("default"))) void shim_inverse3(const double* a, int n, double* out) {
  for (int i = 0; i < n; i++) {
    Eigen::Matrix3d m;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) m(r, c) = a[9 * i + 3 * r + c];
    const Eigen::Matrix3d inv = m.inverse();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) out[9 * i + 3 * r + c] = inv(r, c);
  }
}

__attribute__((visibility("default"))) void shim_inverse4(const double* a, int n, double* out) {
  for (int i = 0; i < n; i++) {
    Eigen::Matrix4d m;
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) m(r, c) = a[16 * i + 4 * r + c];
    const Eigen::Matrix4d inv = m.inverse();
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) out[16 * i + 4 * r + c] = inv(r, c);
  }
}

// eigenvalues ascending, eigenvectors as columns of V (row-major out)
__attribute__((visibility("default"))) void shim_eig3(const double* a, int n, double* evals, double* evecs) {
  for (int i = 0; i < n; i++) {
    Eigen::Matrix3d m;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) m(r, c) = a[9 * i + 3 * r + c];
    Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> eig;
    eig.computeDirect(m);
    for (int k = 0; k < 3; k++) {
      evals[3 * i + k] = eig.eigenvalues()[k];
      for (int r = 0; r < 3; r++) evecs[9 * i + 3 * r + k] = eig.eigenvectors()(r, k);
    }
  }
}

// Pose3(A).inverse() * Pose3(B) as a 4x4 (integrated_matching_cost_factor.cpp:59-66), Pose3(A).inverse().matrix(), and Isometry3d(A).inverse() * Isometry3d(B)
__attribute__((visibility("default"))) void shim_pose_ops(const double* a, const double* b, int n, double* inv_a_times_b, double* inv_a, double* iso_inv_a_times_b) {
  for (int i = 0; i < n; i++) {
    Eigen::Matrix4d A, B;
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        A(r, c) = a[16 * i + 4 * r + c];
        B(r, c) = b[16 * i + 4 * r + c];
      }
    const Eigen::Matrix4d d = (gtsam::Pose3(A).inverse() * gtsam::Pose3(B)).matrix();
    const Eigen::Matrix4d ia = gtsam::Pose3(A).inverse().matrix();
    const Eigen::Matrix4d di = (Eigen::Isometry3d(A).inverse() * Eigen::Isometry3d(B)).matrix();
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        inv_a_times_b[16 * i + 4 * r + c] = d(r, c);
        inv_a[16 * i + 4 * r + c] = ia(r, c);
        iso_inv_a_times_b[16 * i + 4 * r + c] = di(r, c);
      }
  }
}

__attribute__((visibility("default"))) void shim_hat_expmap(const double* xi, int n, double* hat, double* expmap) {
  for (int i = 0; i < n; i++) {
    const Eigen::Matrix3d H = gtsam::SO3::Hat(Eigen::Vector3d(xi[6 * i], xi[6 * i + 1], xi[6 * i + 2]));
    Eigen::Matrix<double, 6, 1> v;
    for (int k = 0; k < 6; k++) v[k] = xi[6 * i + k];
    const Eigen::Matrix4d E = gtsam::Pose3::Expmap(v).matrix();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) hat[9 * i + 3 * r + c] = H(r, c);
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) expmap[16 * i + 4 * r + c] = E(r, c);
  }
}

// the products the factor forms around the inverse: R C R^T + B and J^T M J blocks use operator*, transpose, +: A * B * A^T + C for 3x3
__attribute__((visibility("default"))) void shim_sandwich3(const double* a, const double* b, const double* c, int n, double* out) {
  for (int i = 0; i < n; i++) {
    Eigen::Matrix3d A, B, Cm;
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 3; k++) {
        A(r, k) = a[9 * i + 3 * r + k];
        B(r, k) = b[9 * i + 3 * r + k];
        Cm(r, k) = c[9 * i + 3 * r + k];
      }
    const Eigen::Matrix3d S = Cm + A * B * A.transpose();
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 3; k++) out[9 * i + 3 * r + k] = S(r, k);
  }
}

}  // extern "C"
