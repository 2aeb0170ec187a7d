#pragma once
#include <memory>
#include <gtsam/inference/Key.h>
namespace gtsam {
class GaussianFactor {
public:
  using shared_ptr = std::shared_ptr<GaussianFactor>;
  virtual ~GaussianFactor() {}
};
}  // namespace gtsam
