// Stand-in for GTSAM 4.3a0 <gtsam/geometry/Pose3.h>: exactly what the VGICP path calls
// (Pose3(Matrix4), inverse(), operator*, matrix(), SO3::Hat).  Textbook SE(3); see oracle/vgicp_oracle.c header.
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace gtsam {
struct SO3 {
  static Eigen::Matrix3d Hat(const Eigen::Vector3d& v) {  // [0 -z y; z 0 -x; -y x 0]
    Eigen::Matrix3d m = Eigen::Matrix3d::Zero();
    m(0, 1) = -v[2];
    m(0, 2) = v[1];
    m(1, 0) = v[2];
    m(1, 2) = -v[0];
    m(2, 0) = -v[1];
    m(2, 1) = v[0];
    return m;
  }
  template <int R2, int C2>
  static Eigen::Matrix3d Hat(const Eigen::BlockRef<double, R2, C2, 3, 1>& v) { return Hat(Eigen::Vector3d(v)); }
};
class Pose3 {
public:
  Pose3() : R_(Eigen::Matrix3d::Identity()), t_(Eigen::Vector3d::Zero()) {}
  explicit Pose3(const Eigen::Matrix4d& T) : R_(T.block<3, 3>(0, 0)), t_(T.block<3, 1>(0, 3)) {}
  Pose3(const Eigen::Matrix3d& R, const Eigen::Vector3d& t) : R_(R), t_(t) {}
  Pose3 inverse() const {  // Pose3::inverse(): (R^T, -(R^T t))
    const Eigen::Matrix3d Rt = R_.transpose();
    return Pose3(Rt, -(Rt * t_));
  }
  Pose3 operator*(const Pose3& T) const { return Pose3(R_ * T.R_, t_ + R_ * T.t_); }  // Pose3::compose
  Eigen::Matrix4d matrix() const {
    Eigen::Matrix4d m = Eigen::Matrix4d::Identity();
    m.block<3, 3>(0, 0) = R_;
    m.block<3, 1>(0, 3) = t_;
    return m;
  }
  // Pose3::Expmap, xi = [omega, v] (Rodrigues + the SE(3) left Jacobian of omega applied to v)
  static Pose3 Expmap(const Eigen::Matrix<double, 6, 1>& xi) {
    const Eigen::Vector3d w(xi[0], xi[1], xi[2]), v(xi[3], xi[4], xi[5]);
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    const Eigen::Matrix3d W = SO3::Hat(w), W2 = W * W;
    const double A = th > 1e-10 ? std::sin(th) / th : 1.0, B = th > 1e-10 ? (1.0 - std::cos(th)) / th2 : 0.5, C = th > 1e-10 ? (th - std::sin(th)) / (th2 * th) : 1.0 / 6.0;
    const Eigen::Matrix3d R = Eigen::Matrix3d::Identity() + W * A + W2 * B;
    const Eigen::Matrix3d V = Eigen::Matrix3d::Identity() + W * B + W2 * C;
    return Pose3(R, V * v);
  }
  Pose3 retract(const Eigen::Matrix<double, 6, 1>& xi) const { return (*this) * Expmap(xi); }
  const Eigen::Matrix3d& rotationMatrix() const { return R_; }
  const Eigen::Vector3d& translation() const { return t_; }

private:
  Eigen::Matrix3d R_;
  Eigen::Vector3d t_;
};
}  // namespace gtsam
