// Stand-in for <gtsam/linear/HessianFactor.h>: container of the blocks the factor hands over
// (HessianFactor(j1, j2, G11, G12, g1, G22, g2, f) / (j, G, g, f)).
#pragma once
#include <Eigen/Core>
#include <gtsam/linear/GaussianFactor.h>
namespace gtsam {
class HessianFactor : public GaussianFactor {
public:
  using shared_ptr = std::shared_ptr<HessianFactor>;
  using M6 = Eigen::Matrix<double, 6, 6>;
  using V6 = Eigen::Matrix<double, 6, 1>;
  HessianFactor(Key j, const M6& G, const V6& g, double f) : keys{j}, G11(M6::Zero()), G12(M6::Zero()), G22(G), g1(V6::Zero()), g2(g), f(f), binary(false) {}
  HessianFactor(Key j1, Key j2, const M6& G11, const M6& G12, const V6& g1, const M6& G22, const V6& g2, double f)
  : keys{j1, j2}, G11(G11), G12(G12), G22(G22), g1(g1), g2(g2), f(f), binary(true) {}
  KeyVector keys;
  M6 G11, G12, G22;
  V6 g1, g2;
  double f;
  bool binary;
};
}  // namespace gtsam
