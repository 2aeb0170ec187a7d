// Stand-in for the cmake-generated gtsam_points/config.hpp (config.hpp.in): OpenMP backend, no TBB, no CUDA.
#pragma once
#define GTSAM_POINTS_VERSION_MAJOR 1
#define GTSAM_POINTS_VERSION_MINOR 2
#define GTSAM_POINTS_VERSION_PATCH 1
#define GTSAM_POINTS_VERSION_STRING "1.2.1"
#define GTSAM_POINTS_USE_OPENMP
