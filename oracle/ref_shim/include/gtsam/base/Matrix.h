#pragma once
#include <Eigen/Core>
#include <vector>
namespace gtsam {
struct Matrix { std::vector<double> data; int rows = 0, cols = 0; };  // dynamic matrices are not used on the VGICP path
using Vector = Matrix;
using Matrix6 = Eigen::Matrix<double, 6, 6>;
using Vector6 = Eigen::Matrix<double, 6, 1>;
using Matrix4 = Eigen::Matrix4d;
}  // namespace gtsam
