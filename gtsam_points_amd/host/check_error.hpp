// check_error.hpp -- the reference's "log and continue" convention (cuda/check_error.cu:8-18) on top of the C-ABI status codes
#pragma once
#include <gtsam_points_hip.h>

#include <iostream>

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

}  // namespace gtsam_points
