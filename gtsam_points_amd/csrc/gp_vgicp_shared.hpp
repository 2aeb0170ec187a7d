// gp_vgicp_shared.hpp -- descriptors and constants shared by the VGICP / GICP translation units
#pragma once

#include "gp_device.hpp"

namespace gp {

constexpr int kBlockThreads = 256;
constexpr int kPointsPerThread = 4;
constexpr int kTilePoints = kBlockThreads * kPointsPerThread;  // 1024 source points per workgroup
constexpr int kNumXCD = 8;

struct FactorDesc {
  const float* points;   // [n][3]
  const float* covs;     // [n][9]
  const float* normals;  // [n][3] or null
  double* posed;         // [map.num_voxels][10] this factor's voxel statistics in the SOURCE frame of its linearisation pose (or null)
  VoxelMapView map;
  int n;
  int surface_validation;
  int tile_begin;
  int tile_count;
};


// a single-factor launch carries its poses AND its factor descriptor in the kernel arguments: no H2D copy and no
// dependent descriptor loads on the latency path (2.8 us per workgroup in the timeline traces)
struct InlinePoses {
  double lin[16];
  double eval[16];
  FactorDesc factor;
  int use;
  int tile_points;
  int src_frame;  // the partial sums are in the source frame: the finalize kernel rotates them back
};

struct TileDesc {
  int factor;
  int begin;  // first point
  int count;  // <= kTilePoints
};

__device__ __forceinline__ int xcd_swizzle(int b, int num_tiles) {
  // workgroup b -> tile index; tiles [x*per, (x+1)*per) go to XCD x (dispatcher places workgroup b on XCD b % 8)
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  return (b % kNumXCD) * per + b / kNumXCD;
}

enum : int { MODE_LIN = 0, MODE_ERR = 1, MODE_LIN_GENERAL = 2 };

// host-side launchers of the finalize kernels (defined in gp_vgicp.hip; used by gp_knn.hip for the GICP factor, whose
// partial rows have the same layout): one factor, rigid pose given both on the host (kernel arguments) and on the device
int launch_finalize_single(hipStream_t stream, const double* pose_dev, const double* pose_host, const double* partials, int num_tiles, gp_linearized6* out_dev);
int launch_finalize_error_single(hipStream_t stream, const double* partials, int num_tiles, double* out_dev);

}  // namespace gp
