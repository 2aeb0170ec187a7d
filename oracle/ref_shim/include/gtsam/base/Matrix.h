#pragma once
#include <Eigen/Core>
#include <vector>
namespace gtsam {
struct Matrix { std::vector<double> data; int rows = 0, cols = 0; };  // dynamic matrices are not used on the VGICP path
using Vector = Matrix;
}  // namespace gtsam
