// gp_device.hpp -- device-side building blocks shared by the gfx950 kernels.
//
// Replaces the reference's device functors (include/gtsam_points/cuda/kernels/*.cuh):
//   vector3_hash.cuh:14-76   hash_combine / vector3i_hash / lookup_voxel
//   lookup_voxels.cuh:19-97  transform + optional surface validation + lookup
//   vgicp_derivatives.cuh    per-point residual / Mahalanobis / H,b terms
// Everything here is written for 64-lane wavefronts and double precision (the parity
// target is the CPU IntegratedVGICPFactor, <= 1e-5 relative on H and b).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gtsam_points_hip.h"

namespace gp {

// ---------------------------------------------------------------------------------------
// HBM layout of one voxel as the kernels gather it: ONE aligned 64-byte record
// (half a 128-B cache line) instead of the reference's three separate arrays
// (voxel_means 12 B + voxel_covs 36 B + num_points 4 B, gaussian_voxelmap_gpu.hpp:93-97).
//   mean_local : voxel mean minus the voxel centre ((coord + 0.5) * leaf); |mean_local| <= leaf,
//                so a float carries it to ~3e-8 * leaf absolute -- the double mean is
//                reconstructed as centre + mean_local.
//   cov        : upper triangle (xx,xy,xz,yy,yz,zz) of the mean input covariance, double.
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(64))) VoxelRecord {
  float mean_local[3];
  int num_points;
  double cov[6];
};
static_assert(sizeof(VoxelRecord) == 64, "VoxelRecord must be 64 B");

// Occupancy-block grid: the lookup structure of the VGICP pipeline kernel.  The map's bounding box is cut into blocks of
// 4 x 4 x 4 voxels; one 16-byte entry per block holds the 64 occupancy bits and the index of the block's first voxel.
// Voxels are numbered in (block, bit) order at build time, so  index = base + popcount(bits below this voxel's bit):
// ONE 16-B load per lookup (hits and misses alike, no hashing, no key compare, no probe loop), the whole structure is
// 16 B per 64 cells (0.6 MB for the 320 x 320 x 24-voxel bench map: it lives in L2 / L1), neighbouring points read the
// same entry, and neighbouring voxels have neighbouring records.  bit = (z & 3) << 4 | (y & 3) << 2 | (x & 3).
struct __attribute__((aligned(16))) GridBlock {
  unsigned long long bits;
  int base;
  int pad;
};
static_assert(sizeof(GridBlock) == 16, "GridBlock must be 16 B");

struct VoxelMapView {
  // block grid (null when the bounding box would need more than the block budget: the line table below is used then)
  const GridBlock* gblocks;
  int glo[3];   // block coordinate (voxel coordinate >> 2) of the box's low corner
  int gdim[3];  // blocks per axis
  // line table, private to the VGICP pipeline kernel (built from the voxel list after insert / assign / reload):
  // 64-B lines of 4 keys {coord, voxel_index or -1}, filled front to back, cheap 32-bit hash, power-of-two line count with
  // at most one key per 8 slots on average.  A lookup reads the whole home line in one round trip: match -> index into
  // `records`; a free slot in the line -> the voxel does not exist; only a full line without a match (P ~ 2e-3) sends the
  // lookup on to the next line.
  const gp_voxel_bucket* plines;
  uint32_t plmask;  // number of lines - 1
  uint32_t pad_;
  // reference-visible table (reference hash + max_bucket_scan_count probe rule) and compact records
  const gp_voxel_bucket* buckets;
  const VoxelRecord* records;
  const int* voxel_coords;  // [num_voxels][3] integer voxel coordinates (the source-frame pre-pass rebuilds the f64 mean from them)
  uint32_t num_buckets;
  uint32_t bucket_mask;  // num_buckets - 1 when num_buckets is a power of two, else 0
  int max_scan;
  int num_voxels;
  double inv_leaf;  // 1.0 / leaf, as incremental_voxelmap_impl.hpp:14
  double leaf;
};

// pose = rows of [R | t] in double (read from the column-major double[16] the host uploads)
struct Pose {
  double r00, r01, r02, r10, r11, r12, r20, r21, r22;
  double tx, ty, tz;
};

__device__ __forceinline__ Pose load_pose(const double* __restrict__ m /*col-major 4x4*/) {
  Pose p;
  p.r00 = m[0];
  p.r10 = m[1];
  p.r20 = m[2];
  p.r01 = m[4];
  p.r11 = m[5];
  p.r21 = m[6];
  p.r02 = m[8];
  p.r12 = m[9];
  p.r22 = m[10];
  p.tx = m[12];
  p.ty = m[13];
  p.tz = m[14];
  return p;
}

// boost-style hash_combine, cuda/kernels/vector3_hash.cuh:14-27 (kept bit-identical so that bucket
// tables written by either implementation are interchangeable, e.g. through save_compact / load)
__host__ __device__ __forceinline__ void hash_combine(uint64_t& h, uint64_t k) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  k *= m;
  k ^= k >> 47;
  k *= m;
  h ^= k;
  h *= m;
  h += 0xe6546b64ull;
}

// vector3i_hash, vector3_hash.cuh:33-39 (int -> uint64_t is a sign extension)
__host__ __device__ __forceinline__ uint64_t coord_hash(int x, int y, int z) {
  uint64_t seed = 0;
  hash_combine(seed, (uint64_t)(int64_t)x);
  hash_combine(seed, (uint64_t)(int64_t)y);
  hash_combine(seed, (uint64_t)(int64_t)z);
  return seed;
}

// private 32-bit hash of the kernels' own table (3 + 2 v_mul_lo_u32 instead of nine 64-bit multiplies)
__host__ __device__ __forceinline__ uint32_t coord_hash32(int x, int y, int z) {
  uint32_t h = (uint32_t)x * 73856093u ^ (uint32_t)y * 19349669u ^ (uint32_t)z * 83492791u;
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  h *= 0x297a2d39u;
  h ^= h >> 15;
  return h;
}

// fast_floor in DOUBLE, util/fast_floor.hpp:12-15 (the CPU map's rule; the reference GPU map floors
// float(x)/float(res), vector3_hash.cuh:41-50, which flips ~5e-5 of points across voxel faces)
// LiDAR clouds carry NaN / inf returns.  A non-finite coordinate has no voxel (the reference floors it into an undefined integer that no
// table holds); on the device the float -> int conversion of a NaN is 0, i.e. voxel (0, 0, 0), which may well exist: every lookup of a
// transformed point is guarded with one of these, so that such a point contributes nothing (no correspondence, not an inlier, no overlap)
__host__ __device__ __forceinline__ bool finite3(float x, float y, float z) {
  const float s = (x + y) + z;  // NaN if any of them is NaN or two infinities cancel, +-inf if one is infinite
  return s - s == 0.0f;
}
__host__ __device__ __forceinline__ bool finite3(double x, double y, double z) {
  const double s = (x + y) + z;
  return s - s == 0.0;
}

__host__ __device__ __forceinline__ int fast_floor(double x) {
  const int n = (int)x;
  return n - (x < (double)n);
}

__host__ __device__ __forceinline__ uint32_t bucket_index(uint64_t hash, int i, uint32_t num_buckets, uint32_t mask) {
  const uint64_t h = hash + (uint64_t)i;
  return mask ? (uint32_t)(h & (uint64_t)mask) : (uint32_t)(h % (uint64_t)num_buckets);
}

// lookup_voxel, vector3_hash.cuh:53-76: linear probing, stop at an empty bucket or a coordinate match
__device__ __forceinline__ int lookup_voxel(const VoxelMapView& m, int cx, int cy, int cz) {
  const uint64_t hash = coord_hash(cx, cy, cz);
  const int4* __restrict__ buckets = reinterpret_cast<const int4*>(m.buckets);
  for (int i = 0; i < m.max_scan; i++) {
    const int4 b = buckets[bucket_index(hash, i, m.num_buckets, m.bucket_mask)];
    if (b.w < 0) return -1;
    if (b.x == cx && b.y == cy && b.z == cz) return b.w;
  }
  return -1;
}

// ---------------------------------------------------------------------------------------
// per-block partial / per-factor reduced sums: 29 doubles, padded to 32.
// Only the TARGET-side normal equations are accumulated per point:
//   J_t = [-[q]x, I]  (vgicp_derivatives.cuh:53-55, integrated_vgicp_factor_impl.hpp:232-234)
//   H_t = J_t^T M J_t = [[-S K, -K^T], [-K, M]],  S = [q]x, K = M S
//   b_t = J_t^T M r   = [q x (M r); M r]
// The source-side blocks follow exactly from J_s = -J_t * Ad(delta) (Ad = adjoint of delta, [omega,v] order):
//   H_s = Ad^T H_t Ad,  H_ts = -H_t Ad,  b_s = -Ad^T b_t
// which the finalize kernel applies once per factor in double.  29 instead of 92 accumulators.
// ---------------------------------------------------------------------------------------
enum : int {
  ACC_COUNT = 0,
  ACC_ERR = 1,
  ACC_M = 2,     // 6: xx xy xz yy yz zz
  ACC_K = 8,     // 9: row-major K = M S
  ACC_TL = 17,   // 6: upper triangle of -S K
  ACC_QXMR = 23, // 3
  ACC_MR = 26,   // 3
  ACC_SIZE = 29,
  ACC_STRIDE = 32,
  // GENERAL mode (pose whose 3x3 block is not orthonormal to 1e-9, e.g. built from a 6-digit quaternion as
  // src/test/test_matching_cost_factors.cpp:50-55 does): the adjoint identity no longer holds exactly, so the
  // source-side and cross blocks are accumulated explicitly like the reference (92 sums).
  ACCG_HS_TL = 29,  // 6: upper triangle of G^T M G,  G = R [p]x
  ACCG_HS_BL = 35,  // 9: row-major -(R^T M G)
  ACCG_HS_BR = 44,  // 6: upper triangle of R^T M R
  ACCG_HTS = 50,    // 36: row-major J_t^T M J_s
  ACCG_BS = 86,     // 6
  ACCG_SIZE = 92,
  ACCG_STRIDE = 96
};

// symmetric 3x3 inverse by cofactors (Eigen's fixed-size 3x3 inverse; vgicp_derivatives.cuh:49,
// integrated_vgicp_factor_impl.hpp:140)
__device__ __forceinline__ void inverse_sym3(double c00, double c01, double c02, double c11, double c12, double c22, double* m /*6*/) {
  const double i00 = c11 * c22 - c12 * c12;
  const double i01 = c02 * c12 - c01 * c22;
  const double i02 = c01 * c12 - c02 * c11;
  const double det = c00 * i00 + c01 * i01 + c02 * i02;
  const double invdet = 1.0 / det;
  m[0] = i00 * invdet;
  m[1] = i01 * invdet;
  m[2] = i02 * invdet;
  m[3] = (c00 * c22 - c02 * c02) * invdet;
  m[4] = (c01 * c02 - c00 * c12) * invdet;
  m[5] = (c00 * c11 - c01 * c01) * invdet;
}

// fused covariance inverse M = (C_B + R C_A R^T)^-1 (vgicp_derivatives.cuh:48-49)
__device__ __forceinline__ void fused_mahalanobis(const Pose& T, const double* ca /*6 sym*/, const double* cb /*6 sym*/, double* m /*6*/) {
  // RC = R * C_A
  const double a00 = ca[0], a01 = ca[1], a02 = ca[2], a11 = ca[3], a12 = ca[4], a22 = ca[5];
  const double rc00 = T.r00 * a00 + T.r01 * a01 + T.r02 * a02;
  const double rc01 = T.r00 * a01 + T.r01 * a11 + T.r02 * a12;
  const double rc02 = T.r00 * a02 + T.r01 * a12 + T.r02 * a22;
  const double rc10 = T.r10 * a00 + T.r11 * a01 + T.r12 * a02;
  const double rc11 = T.r10 * a01 + T.r11 * a11 + T.r12 * a12;
  const double rc12 = T.r10 * a02 + T.r11 * a12 + T.r12 * a22;
  const double rc20 = T.r20 * a00 + T.r21 * a01 + T.r22 * a02;
  const double rc21 = T.r20 * a01 + T.r21 * a11 + T.r22 * a12;
  const double rc22 = T.r20 * a02 + T.r21 * a12 + T.r22 * a22;
  // (RC) R^T, upper triangle, plus C_B
  const double c00 = cb[0] + rc00 * T.r00 + rc01 * T.r01 + rc02 * T.r02;
  const double c01 = cb[1] + rc00 * T.r10 + rc01 * T.r11 + rc02 * T.r12;
  const double c02 = cb[2] + rc00 * T.r20 + rc01 * T.r21 + rc02 * T.r22;
  const double c11 = cb[3] + rc10 * T.r10 + rc11 * T.r11 + rc12 * T.r12;
  const double c12 = cb[4] + rc10 * T.r20 + rc11 * T.r21 + rc12 * T.r22;
  const double c22 = cb[5] + rc20 * T.r20 + rc21 * T.r21 + rc22 * T.r22;
  inverse_sym3(c00, c01, c02, c11, c12, c22, m);
}

__device__ __forceinline__ void voxel_center(const VoxelMapView& m, int cx, int cy, int cz, double& x, double& y, double& z) {
  x = ((double)cx + 0.5) * m.leaf;
  y = ((double)cy + 0.5) * m.leaf;
  z = ((double)cz + 0.5) * m.leaf;
}

// surface validation, lookup_voxels.cuh:41-50: reject when normalized(q) . (R n) > cos(80 deg)
__device__ __forceinline__ bool surface_rejected(const Pose& T, double qx, double qy, double qz, const float* __restrict__ n) {
  const double nx = (double)n[0], ny = (double)n[1], nz = (double)n[2];
  const double tnx = T.r00 * nx + T.r01 * ny + T.r02 * nz;
  const double tny = T.r10 * nx + T.r11 * ny + T.r12 * nz;
  const double tnz = T.r20 * nx + T.r21 * ny + T.r22 * nz;
  const double inv_norm = 1.0 / sqrt(qx * qx + qy * qy + qz * qz);
  return (qx * tnx + qy * tny + qz * tnz) * inv_norm > 0.174;
}

}  // namespace gp
