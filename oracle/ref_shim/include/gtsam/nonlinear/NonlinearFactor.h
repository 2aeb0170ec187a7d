// Stand-in for <gtsam/nonlinear/NonlinearFactor.h> + Values (oracle/_ref only)
#pragma once
#include <iostream>
#include <map>
#include <memory>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Key.h>
#include <gtsam/linear/GaussianFactor.h>
#ifndef GTSAM_MAKE_ALIGNED_OPERATOR_NEW
#define GTSAM_MAKE_ALIGNED_OPERATOR_NEW  // gtsam/base/types.h
#endif
namespace gtsam {
class Values {
public:
  template <typename T>
  const T& at(Key k) const { return poses_.at(k); }
  void insert(Key k, const Pose3& p) { poses_[k] = p; }
  void update(Key k, const Pose3& p) { poses_[k] = p; }
  bool exists(Key k) const { return poses_.count(k) > 0; }
  size_t size() const { return poses_.size(); }
private:
  std::map<Key, Pose3> poses_;
};
class NonlinearFactor {
public:
  using shared_ptr = std::shared_ptr<NonlinearFactor>;
  NonlinearFactor() {}
  template <typename CONTAINER>
  explicit NonlinearFactor(const CONTAINER& keys) : keys_(keys.begin(), keys.end()) {}
  virtual ~NonlinearFactor() {}
  const KeyVector& keys() const { return keys_; }
  virtual size_t dim() const = 0;
  virtual double error(const Values& c) const = 0;
  virtual std::shared_ptr<GaussianFactor> linearize(const Values& c) const = 0;
  virtual shared_ptr clone() const { return shared_ptr(); }
  virtual void print(const std::string& s = "", const KeyFormatter& keyFormatter = DefaultKeyFormatter) const { (void)keyFormatter; std::cout << s; }
protected:
  KeyVector keys_;
};
}  // namespace gtsam
