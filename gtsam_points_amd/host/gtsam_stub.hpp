// gtsam_stub.hpp -- the handful of GTSAM types the VGICP GPU path touches, as a dependency-free stand-in.
//
// GTSAM is not installed on the build image, so the C++ mirror of the reference classes (this directory) is written
// against this stand-in.  With -DGTSAM_POINTS_HIP_WITH_GTSAM the real headers are used instead and this file is skipped.
// Semantics follow GTSAM 4.3a0: Pose3 right-multiplicative retract = compose(Expmap(xi)), tangent order [omega, v];
// HessianFactor(keys, G11, G12, g1, G22, g2, f).
#pragma once
#ifndef GTSAM_POINTS_HIP_WITH_GTSAM

#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace gtsam {

using Key = std::uint64_t;
using KeyVector = std::vector<Key>;
using KeyFormatter = std::function<std::string(Key)>;
inline std::string DefaultKeyFormatter(Key k) { return std::to_string(k); }

using Matrix4 = std::array<double, 16>;  // column-major
using Matrix6 = std::array<double, 36>;  // column-major
using Vector6 = std::array<double, 6>;

class Pose3 {
public:
  Pose3() { m_.fill(0.0); m_[0] = m_[5] = m_[10] = m_[15] = 1.0; }
  explicit Pose3(const Matrix4& m) : m_(m) {}
  const Matrix4& matrix() const { return m_; }
  double R(int r, int c) const { return m_[c * 4 + r]; }
  double t(int r) const { return m_[12 + r]; }

  Pose3 inverse() const {  // (R^T, -R^T t)
    Pose3 o;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) o.m_[c * 4 + r] = R(c, r);
      o.m_[12 + r] = -(R(0, r) * t(0) + R(1, r) * t(1) + R(2, r) * t(2));
    }
    return o;
  }
  Pose3 operator*(const Pose3& b) const {
    Pose3 o;
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += m_[k * 4 + r] * b.m_[c * 4 + k];
        o.m_[c * 4 + r] = s;
      }
    return o;
  }
  static Pose3 Expmap(const Vector6& xi) {
    const double wx = xi[0], wy = xi[1], wz = xi[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
    const double W[9] = {0, wz, -wy, -wz, 0, wx, wy, -wx, 0};  // col-major Hat(w)
    double W2[9];
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) W2[c * 3 + r] = W[0 * 3 + r] * W[c * 3 + 0] + W[1 * 3 + r] * W[c * 3 + 1] + W[2 * 3 + r] * W[c * 3 + 2];
    const double A = th > 1e-10 ? std::sin(th) / th : 1.0, B = th > 1e-10 ? (1 - std::cos(th)) / th2 : 0.5, Cc = th > 1e-10 ? (th - std::sin(th)) / (th2 * th) : 1.0 / 6;
    Pose3 o;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) o.m_[c * 4 + r] = (r == c ? 1.0 : 0.0) + A * W[c * 3 + r] + B * W2[c * 3 + r];
    for (int r = 0; r < 3; r++) {
      double s = 0.0;
      for (int c = 0; c < 3; c++) s += ((r == c ? 1.0 : 0.0) + B * W[c * 3 + r] + Cc * W2[c * 3 + r]) * xi[3 + c];
      o.m_[12 + r] = s;
    }
    return o;
  }
  Pose3 retract(const Vector6& xi) const { return (*this) * Expmap(xi); }

private:
  Matrix4 m_;
};

class Values {
public:
  template <typename T>
  const T& at(Key k) const { return poses_.at(k); }
  void insert(Key k, const Pose3& p) { poses_[k] = p; }
  void update(Key k, const Pose3& p) { poses_[k] = p; }
  bool exists(Key k) const { return poses_.count(k) > 0; }
  const std::map<Key, Pose3>& poses() const { return poses_; }

private:
  std::map<Key, Pose3> poses_;
};

class GaussianFactor {
public:
  using shared_ptr = std::shared_ptr<GaussianFactor>;
  virtual ~GaussianFactor() {}
};

class HessianFactor : public GaussianFactor {
public:
  using shared_ptr = std::shared_ptr<HessianFactor>;
  HessianFactor(Key j, const Matrix6& G, const Vector6& g, double f) : keys{j}, G11(G), g1(g), f(f), binary(false) {}
  HessianFactor(Key j1, Key j2, const Matrix6& G11, const Matrix6& G12, const Vector6& g1, const Matrix6& G22, const Vector6& g2, double f)
  : keys{j1, j2}, G11(G11), G12(G12), G22(G22), g1(g1), g2(g2), f(f), binary(true) {}
  KeyVector keys;
  Matrix6 G11{}, G12{}, G22{};
  Vector6 g1{}, g2{};
  double f;
  bool binary;
};

class NonlinearFactor {
public:
  using shared_ptr = std::shared_ptr<NonlinearFactor>;
  template <typename CONTAINER>
  explicit NonlinearFactor(const CONTAINER& keys) : keys_(keys.begin(), keys.end()) {}
  virtual ~NonlinearFactor() {}
  const KeyVector& keys() const { return keys_; }
  virtual size_t dim() const = 0;
  virtual double error(const Values& values) const = 0;
  virtual GaussianFactor::shared_ptr linearize(const Values& values) const = 0;
  virtual shared_ptr clone() const = 0;
  virtual void print(const std::string& s = "", const KeyFormatter& f = DefaultKeyFormatter) const { (void)f; std::cout << s << std::endl; }

protected:
  KeyVector keys_;
};

class NonlinearFactorGraph {
public:
  void add(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  template <typename T, typename... Args>
  std::shared_ptr<T> emplace_shared(Args&&... args) {
    auto f = std::make_shared<T>(std::forward<Args>(args)...);
    factors_.push_back(f);
    return f;
  }
  size_t size() const { return factors_.size(); }
  std::vector<NonlinearFactor::shared_ptr>::const_iterator begin() const { return factors_.begin(); }
  std::vector<NonlinearFactor::shared_ptr>::const_iterator end() const { return factors_.end(); }
  const NonlinearFactor::shared_ptr& operator[](size_t i) const { return factors_[i]; }

private:
  std::vector<NonlinearFactor::shared_ptr> factors_;
};

template <typename T, typename... Args>
std::shared_ptr<T> make_shared(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }

}  // namespace gtsam

namespace gtsam_points {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename T, typename U>
std::shared_ptr<T> dynamic_pointer_cast(const std::shared_ptr<U>& p) { return std::dynamic_pointer_cast<T>(p); }
}  // namespace gtsam_points

#endif  // GTSAM_POINTS_HIP_WITH_GTSAM
