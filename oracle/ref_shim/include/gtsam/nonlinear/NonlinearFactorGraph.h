// Stand-in for <gtsam/nonlinear/NonlinearFactorGraph.h>: an ordered container of factor pointers (add / emplace_shared / iteration),
// which is all the linearization hook and the GPU factor set use
#pragma once
#include <memory>
#include <vector>
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam {
class NonlinearFactorGraph {
public:
  using const_iterator = std::vector<NonlinearFactor::shared_ptr>::const_iterator;
  void add(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  void push_back(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  template <typename T, typename... Args>
  std::shared_ptr<T> emplace_shared(Args&&... args) {
    auto f = std::make_shared<T>(std::forward<Args>(args)...);
    factors_.push_back(f);
    return f;
  }
  size_t size() const { return factors_.size(); }
  const_iterator begin() const { return factors_.begin(); }
  const_iterator end() const { return factors_.end(); }
  const NonlinearFactor::shared_ptr& operator[](size_t i) const { return factors_[i]; }
  const NonlinearFactor::shared_ptr& at(size_t i) const { return factors_.at(i); }

private:
  std::vector<NonlinearFactor::shared_ptr> factors_;
};
}  // namespace gtsam
