// check_error.hpp -- the reference's "log and continue" convention (cuda/check_error.cu:8-18) on top of the C-ABI status codes,
// and the two conversions every class of the mirror needs: the opaque stream type and Eigen poses -> the C-ABI's double[16].
//
// This directory is the C++ half of the drop-in: the classes below have the reference's names, constructors and virtuals and
// derive from the reference's OWN base classes (gtsam_points/factors/nonlinear_factor_gpu.hpp, types/point_cloud.hpp,
// types/gaussian_voxelmap.hpp, types/offloadable.hpp, optimizers/linearization_hook.hpp), so an application built against
// gtsam_points swaps its CUDA translation units for these headers + gtsam_points_hip_host.cpp + libgtsam_points_hip.so.
// They replace, and must not be mixed with, the reference's cuda/*.hpp, types/*_gpu.hpp and factors/*_gpu.hpp headers.
#pragma once
#include <gtsam_points_hip.h>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <iostream>

struct CUstream_st;  // the reference's opaque stream type (forward-declared in its headers); here it IS a hipStream_t

namespace gtsam_points {

class HIPCheckError {
public:
  void operator<<(int status) const {
    if (status == GP_OK) return;
    std::cerr << "warning: gtsam_points_hip status " << status << std::endl;
    std::cerr << "       : " << gp_last_error() << std::endl;
  }
};

static const HIPCheckError check_error;

inline gp_stream_t gp_stream(CUstream_st* stream) { return reinterpret_cast<gp_stream_t>(stream); }

// column-major 4x4 double, the pose format of the C-ABI
struct Pose16 {
  double m[16];
  const double* data() const { return m; }
};
inline Pose16 pose16(const Eigen::Isometry3d& T) {
  Pose16 p;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) p.m[c * 4 + r] = T.matrix()(r, c);
  return p;
}

template <typename T>
inline const float* as_floats(const T* p) {
  return reinterpret_cast<const float*>(p);
}

}  // namespace gtsam_points
