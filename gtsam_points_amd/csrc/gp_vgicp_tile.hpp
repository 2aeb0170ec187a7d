// gp_vgicp_tile.hpp -- the tuned tile kernel of the VGICP path (rigid poses; MODE_LIN / MODE_ERR).
//
// Same arithmetic as accumulate_point<> in gp_vgicp.hip, restructured for the memory system of gfx950:
//   phase A  all kPointsPerThread source points of a lane are loaded first (independent, coalesced across the wave:
//            64 lanes x 12 B points / 36 B covariances are contiguous in memory)
//   phase B  f64 transform + floor + hash, then the FIRST bucket probe of every point is issued back to back
//   phase C  probes are resolved (the rare collision chain walks on), then every hit's 64-B voxel record is requested
//   phase D  f64 fused-covariance inverse and the 29 target-side sums per hit
//   so a lane has up to 4 dependent-load chains in flight instead of one (the v1 kernel was latency-bound:
//   3 serial round trips per point).
//   All pointers are cast to the global address space (descriptors loaded from memory would otherwise make hipcc emit
//   flat_load, which also ties up lgkmcnt).
//   The 64-lane reduction is a transposing butterfly: 32 xor-shuffles of doubles instead of 29 x 6.
#pragma once

#include <type_traits>

#include "gp_vgicp_shared.hpp"

namespace gp {

#define GP_GLOBAL __attribute__((address_space(1)))

template <typename T>
__device__ __forceinline__ const GP_GLOBAL T* as_global(const T* p) {
  return (const GP_GLOBAL T*)p;
}

// builtin vector types: loads through an address-space-qualified pointer compile on the host pass too
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef double v2d __attribute__((ext_vector_type(2)));

struct f3 {
  float x, y, z;
};

// cheap reciprocal: v_rcp_f64 (~2^-23) + two Newton steps (-> ~1 ulp); replaces the ~30-instruction IEEE division
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = x * (2.0 - d * x);
  x = x * (2.0 - d * x);
  return x;
}

// transposing butterfly over a 64-lane wavefront: in v[0..31] per lane, out: lane L holds the wave-wide sum of
// component comp(L) = bitrev5(L >> 1) in v[0] (both lanes of a pair hold the same value).
__device__ __forceinline__ int butterfly_component(int lane) {
  return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

template <int D, int C>
__device__ __forceinline__ void butterfly_step(double* v, int lane) {
  const bool upper = (lane & D) != 0;
#pragma unroll
  for (int k = 0; k < C; k++) {
    const double send = upper ? v[k] : v[k + C];
    const double keep = upper ? v[k + C] : v[k];
    v[k] = keep + __shfl_xor(send, D, 64);
  }
}

__device__ __forceinline__ double butterfly_reduce32(double* v, int lane) {
  butterfly_step<32, 16>(v, lane);
  butterfly_step<16, 8>(v, lane);
  butterfly_step<8, 4>(v, lane);
  butterfly_step<4, 2>(v, lane);
  butterfly_step<2, 1>(v, lane);
  return v[0] + __shfl_xor(v[0], 1, 64);
}

template <int MODE, bool OUTER_F32, int PPT>
__global__ void __launch_bounds__(256) vgicp_tile_kernel2(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                          const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                          double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  const int tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;  // XCD-aware workgroup -> tile map
  if (tile_idx >= num_tiles) return;
  if (inl.stagger > 0 && ((blockIdx.x >> 3) & 1)) {
    for (int k = 0; k < inl.stagger; k++) __builtin_amdgcn_s_sleep(1);  // ~64 cycles each
  }
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
  } else {
    tile = tiles[tile_idx];  // kernel-argument pointers are already known to be global (uniform -> scalar loads)
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const GP_GLOBAL float* points = as_global(f.points);
  const GP_GLOBAL float* covs = as_global(f.covs);
  const GP_GLOBAL v4i* buckets = (const GP_GLOBAL v4i*)f.map.buckets;
  const GP_GLOBAL VoxelRecord* records = as_global(f.map.records);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // ---- phase A: source points and covariances ----
  f3 p[PPT];
  float cov[PPT][6];
  bool active[PPT];
#pragma unroll
  for (int it = 0; it < PPT; it++) {
    const int local = it * 256 + threadIdx.x;
    active[it] = local < tile.count;
    const size_t i = (size_t)tile.begin + (active[it] ? local : 0);
    p[it].x = points[3 * i];
    p[it].y = points[3 * i + 1];
    p[it].z = points[3 * i + 2];
    const GP_GLOBAL float* cp = covs + 9 * i;
    cov[it][0] = cp[0];  // xx
    cov[it][1] = cp[3];  // xy
    cov[it][3] = cp[4];  // yy
    cov[it][2] = cp[6];  // xz
    cov[it][4] = cp[7];  // yz
    cov[it][5] = cp[8];  // zz
  }

  // ---- phase B: transform, voxel coordinate, hash, first probe ----
  int cx[PPT], cy[PPT], cz[PPT];
  uint64_t hash[PPT];
  v4i bk[PPT];
#pragma unroll
  for (int it = 0; it < PPT; it++) {
    const double px = (double)p[it].x, py = (double)p[it].y, pz = (double)p[it].z;
    const double lx = Tl.r00 * px + Tl.r01 * py + Tl.r02 * pz + Tl.tx;
    const double ly = Tl.r10 * px + Tl.r11 * py + Tl.r12 * pz + Tl.ty;
    const double lz = Tl.r20 * px + Tl.r21 * py + Tl.r22 * pz + Tl.tz;
    cx[it] = fast_floor(lx * f.map.inv_leaf);
    cy[it] = fast_floor(ly * f.map.inv_leaf);
    cz[it] = fast_floor(lz * f.map.inv_leaf);
    if (f.surface_validation && active[it] && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * ((size_t)tile.begin + it * 256 + threadIdx.x))) active[it] = false;
    hash[it] = coord_hash(cx[it], cy[it], cz[it]);
    bk[it] = buckets[bucket_index(hash[it], 0, f.map.num_buckets, f.map.bucket_mask)];
  }

  // ---- phase C: resolve probes, request the voxel records ----
  int vid[PPT];
  v4f head[PPT];
  v2d c01[PPT], c23[PPT], c45[PPT];
#pragma unroll
  for (int it = 0; it < PPT; it++) {
    int v = -1;
    if (active[it]) {
      v4i b = bk[it];
      for (int i = 0;;) {
        if (b.w < 0) break;
        if (b.x == cx[it] && b.y == cy[it] && b.z == cz[it]) {
          v = b.w;
          break;
        }
        if (++i >= f.map.max_scan) break;
        b = buckets[bucket_index(hash[it], i, f.map.num_buckets, f.map.bucket_mask)];
      }
    }
    vid[it] = v;
    if (v >= 0) {
      const GP_GLOBAL char* rec = (const GP_GLOBAL char*)(records + v);
      head[it] = *(const GP_GLOBAL v4f*)rec;
      c01[it] = *(const GP_GLOBAL v2d*)(rec + 16);
      c23[it] = *(const GP_GLOBAL v2d*)(rec + 32);
      c45[it] = *(const GP_GLOBAL v2d*)(rec + 48);
    }
  }

  // ---- phase D: per-hit arithmetic ----
  double acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = 0.0;
  float accf[OUTER_F32 ? 32 : 1];
  if constexpr (OUTER_F32) {
#pragma unroll
    for (int k = 0; k < 32; k++) accf[k] = 0.0f;
  }
#pragma unroll
  for (int it = 0; it < PPT; it++) {
    if (vid[it] < 0) continue;
    const double px = (double)p[it].x, py = (double)p[it].y, pz = (double)p[it].z;
    const double ca[6] = {(double)cov[it][0], (double)cov[it][1], (double)cov[it][2], (double)cov[it][3], (double)cov[it][4], (double)cov[it][5]};
    const double cb[6] = {c01[it].x, c01[it].y, c23[it].x, c23[it].y, c45[it].x, c45[it].y};
    // M = (C_B + R C_A R^T)^-1 in f64 (condition number up to 1e3: f32 here would cost ~1e-4 per point, systematic)
    double m[6];
    {
      const double a00 = ca[0], a01 = ca[1], a02 = ca[2], a11 = ca[3], a12 = ca[4], a22 = ca[5];
      const double rc00 = Tl.r00 * a00 + Tl.r01 * a01 + Tl.r02 * a02, rc01 = Tl.r00 * a01 + Tl.r01 * a11 + Tl.r02 * a12, rc02 = Tl.r00 * a02 + Tl.r01 * a12 + Tl.r02 * a22;
      const double rc10 = Tl.r10 * a00 + Tl.r11 * a01 + Tl.r12 * a02, rc11 = Tl.r10 * a01 + Tl.r11 * a11 + Tl.r12 * a12, rc12 = Tl.r10 * a02 + Tl.r11 * a12 + Tl.r12 * a22;
      const double rc20 = Tl.r20 * a00 + Tl.r21 * a01 + Tl.r22 * a02, rc21 = Tl.r20 * a01 + Tl.r21 * a11 + Tl.r22 * a12, rc22 = Tl.r20 * a02 + Tl.r21 * a12 + Tl.r22 * a22;
      const double s00 = cb[0] + rc00 * Tl.r00 + rc01 * Tl.r01 + rc02 * Tl.r02;
      const double s01 = cb[1] + rc00 * Tl.r10 + rc01 * Tl.r11 + rc02 * Tl.r12;
      const double s02 = cb[2] + rc00 * Tl.r20 + rc01 * Tl.r21 + rc02 * Tl.r22;
      const double s11 = cb[3] + rc10 * Tl.r10 + rc11 * Tl.r11 + rc12 * Tl.r12;
      const double s12 = cb[4] + rc10 * Tl.r20 + rc11 * Tl.r21 + rc12 * Tl.r22;
      const double s22 = cb[5] + rc20 * Tl.r20 + rc21 * Tl.r21 + rc22 * Tl.r22;
      const double i00 = s11 * s22 - s12 * s12, i01 = s02 * s12 - s01 * s22, i02 = s01 * s12 - s02 * s11;
      const double invdet = fast_rcp(s00 * i00 + s01 * i01 + s02 * i02);
      m[0] = i00 * invdet;
      m[1] = i01 * invdet;
      m[2] = i02 * invdet;
      m[3] = (s00 * s22 - s02 * s02) * invdet;
      m[4] = (s01 * s02 - s00 * s12) * invdet;
      m[5] = (s00 * s11 - s01 * s01) * invdet;
    }
    // q at the evaluation pose (== linearisation pose for MODE_LIN), residual against centre + mean_local in f64
    const double qx = Te.r00 * px + Te.r01 * py + Te.r02 * pz + Te.tx;
    const double qy = Te.r10 * px + Te.r11 * py + Te.r12 * pz + Te.ty;
    const double qz = Te.r20 * px + Te.r21 * py + Te.r22 * pz + Te.tz;
    const double rx = (((double)cx[it] + 0.5) * f.map.leaf - qx) + (double)head[it].x;
    const double ry = (((double)cy[it] + 0.5) * f.map.leaf - qy) + (double)head[it].y;
    const double rz = (((double)cz[it] + 0.5) * f.map.leaf - qz) + (double)head[it].z;

    if constexpr (!OUTER_F32) {
      const double mrx = m[0] * rx + m[1] * ry + m[2] * rz;
      const double mry = m[1] * rx + m[3] * ry + m[4] * rz;
      const double mrz = m[2] * rx + m[4] * ry + m[5] * rz;
      acc[ACC_COUNT] += 1.0;
      acc[ACC_ERR] += rx * mrx + ry * mry + rz * mrz;
      if constexpr (MODE == MODE_LIN) {
#pragma unroll
        for (int k = 0; k < 6; k++) acc[ACC_M + k] += m[k];
        const double k00 = m[1] * qz - m[2] * qy, k01 = m[2] * qx - m[0] * qz, k02 = m[0] * qy - m[1] * qx;
        const double k10 = m[3] * qz - m[4] * qy, k11 = m[4] * qx - m[1] * qz, k12 = m[1] * qy - m[3] * qx;
        const double k20 = m[4] * qz - m[5] * qy, k21 = m[5] * qx - m[2] * qz, k22 = m[2] * qy - m[4] * qx;
        acc[ACC_K + 0] += k00;
        acc[ACC_K + 1] += k01;
        acc[ACC_K + 2] += k02;
        acc[ACC_K + 3] += k10;
        acc[ACC_K + 4] += k11;
        acc[ACC_K + 5] += k12;
        acc[ACC_K + 6] += k20;
        acc[ACC_K + 7] += k21;
        acc[ACC_K + 8] += k22;
        acc[ACC_TL + 0] += qz * k10 - qy * k20;
        acc[ACC_TL + 1] += qz * k11 - qy * k21;
        acc[ACC_TL + 2] += qz * k12 - qy * k22;
        acc[ACC_TL + 3] += qx * k21 - qz * k01;
        acc[ACC_TL + 4] += qx * k22 - qz * k02;
        acc[ACC_TL + 5] += qy * k02 - qx * k12;
        acc[ACC_QXMR + 0] += qy * mrz - qz * mry;
        acc[ACC_QXMR + 1] += qz * mrx - qx * mrz;
        acc[ACC_QXMR + 2] += qx * mry - qy * mrx;
        acc[ACC_MR + 0] += mrx;
        acc[ACC_MR + 1] += mry;
        acc[ACC_MR + 2] += mrz;
      }
    } else {
      // outer products in f32 on f64-accurate M, r, q (experimental variant)
      const float M0 = (float)m[0], M1 = (float)m[1], M2 = (float)m[2], M3 = (float)m[3], M4 = (float)m[4], M5 = (float)m[5];
      const float RX = (float)rx, RY = (float)ry, RZ = (float)rz, QX = (float)qx, QY = (float)qy, QZ = (float)qz;
      const float mrx = M0 * RX + M1 * RY + M2 * RZ, mry = M1 * RX + M3 * RY + M4 * RZ, mrz = M2 * RX + M4 * RY + M5 * RZ;
      accf[ACC_COUNT] += 1.0f;
      accf[ACC_ERR] += RX * mrx + RY * mry + RZ * mrz;
      if constexpr (MODE == MODE_LIN) {
        accf[ACC_M + 0] += M0;
        accf[ACC_M + 1] += M1;
        accf[ACC_M + 2] += M2;
        accf[ACC_M + 3] += M3;
        accf[ACC_M + 4] += M4;
        accf[ACC_M + 5] += M5;
        const float k00 = M1 * QZ - M2 * QY, k01 = M2 * QX - M0 * QZ, k02 = M0 * QY - M1 * QX;
        const float k10 = M3 * QZ - M4 * QY, k11 = M4 * QX - M1 * QZ, k12 = M1 * QY - M3 * QX;
        const float k20 = M4 * QZ - M5 * QY, k21 = M5 * QX - M2 * QZ, k22 = M2 * QY - M4 * QX;
        accf[ACC_K + 0] += k00;
        accf[ACC_K + 1] += k01;
        accf[ACC_K + 2] += k02;
        accf[ACC_K + 3] += k10;
        accf[ACC_K + 4] += k11;
        accf[ACC_K + 5] += k12;
        accf[ACC_K + 6] += k20;
        accf[ACC_K + 7] += k21;
        accf[ACC_K + 8] += k22;
        accf[ACC_TL + 0] += QZ * k10 - QY * k20;
        accf[ACC_TL + 1] += QZ * k11 - QY * k21;
        accf[ACC_TL + 2] += QZ * k12 - QY * k22;
        accf[ACC_TL + 3] += QX * k21 - QZ * k01;
        accf[ACC_TL + 4] += QX * k22 - QZ * k02;
        accf[ACC_TL + 5] += QY * k02 - QX * k12;
        accf[ACC_QXMR + 0] += QY * mrz - QZ * mry;
        accf[ACC_QXMR + 1] += QZ * mrx - QX * mrz;
        accf[ACC_QXMR + 2] += QX * mry - QY * mrx;
        accf[ACC_MR + 0] += mrx;
        accf[ACC_MR + 1] += mry;
        accf[ACC_MR + 2] += mrz;
      }
    }
  }
  if constexpr (OUTER_F32) {
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = (double)accf[k];
  }

  // ---- wavefront reduction (transposing butterfly), then LDS across the 4 waves ----
  __shared__ double lds[4][ACC_STRIDE];
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) lds[wave][k] = v;
    }
  } else {
    const double s = butterfly_reduce32(acc, lane);
    if ((lane & 1) == 0) lds[wave][butterfly_component(lane)] = s;
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double s = 0.0;
    if (threadIdx.x < NACC) s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = s;
  }
}


// =====================================================================================================================
// vgicp_tile_kernel3 -- as kernel2, plus:
//   * the map's PRIVATE slot table (VoxelMapView::pkeys/pfat): cheap 32-bit hash, and the 64-B record of the home slot is
//     requested together with its key -> 2 dependent round trips per point (source stream, table) instead of 3
//   * a tile is ITERS steps of 256*PPT points; the source points/covariances of step k+1 are requested before step k is
//     processed (register double buffer), with non-temporal loads so that the one-pass source stream does not evict the
//     voxel table from the XCD's L2
//   * OUTER_F32: f64 transform / floor / fused-covariance inverse / residual, f32 outer products and per-lane f32 partial
//     sums over the <= PPT*ITERS points of a lane, f64 from the wavefront reduction on.
// =====================================================================================================================
template <int PPT>
struct SourceRegs {
  float px[PPT], py[PPT], pz[PPT];
  float c[PPT][6];
};

template <int PPT>
__device__ __forceinline__ void load_source(SourceRegs<PPT>& r, const GP_GLOBAL float* points, const GP_GLOBAL float* covs, int tile_begin, int tile_count, int step) {
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    const int local = (step * PPT + j) * 256 + (int)threadIdx.x;
    const size_t i = (size_t)tile_begin + (local < tile_count ? local : 0);
    const GP_GLOBAL float* pp = points + 3 * i;
    const GP_GLOBAL float* cp = covs + 9 * i;
    r.px[j] = __builtin_nontemporal_load(pp);
    r.py[j] = __builtin_nontemporal_load(pp + 1);
    r.pz[j] = __builtin_nontemporal_load(pp + 2);
    r.c[j][0] = __builtin_nontemporal_load(cp);      // xx
    r.c[j][1] = __builtin_nontemporal_load(cp + 3);  // xy
    r.c[j][3] = __builtin_nontemporal_load(cp + 4);  // yy
    r.c[j][2] = __builtin_nontemporal_load(cp + 6);  // xz
    r.c[j][4] = __builtin_nontemporal_load(cp + 7);  // yz
    r.c[j][5] = __builtin_nontemporal_load(cp + 8);  // zz
  }
}

template <int MODE, bool OUTER_F32, int PPT, int ITERS, int ABLATE = 0>
__global__ void __launch_bounds__(256) vgicp_tile_kernel3(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                          const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                          double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  const int tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;  // XCD-aware workgroup -> tile map
  if (tile_idx >= num_tiles) return;
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
  } else {
    tile = tiles[tile_idx];
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const GP_GLOBAL float* points = as_global(f.points);
  const GP_GLOBAL float* covs = as_global(f.covs);
  const GP_GLOBAL v4i* pkeys = (const GP_GLOBAL v4i*)f.map.pkeys;
  const GP_GLOBAL char* pfat = (const GP_GLOBAL char*)f.map.pfat;
  const uint32_t pmask = f.map.pmask;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  // ABLATE (timing experiments only; results are wrong): 1 = no arithmetic, 2 = no table loads, 3 = no source loads, 4 = no butterfly
  SourceRegs<PPT> cur, nxt;
  if constexpr (ABLATE == 3) {
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      cur.px[j] = 0.01f * (float)(threadIdx.x + tile_idx);
      cur.py[j] = 0.02f * (float)threadIdx.x;
      cur.pz[j] = -1.7f;
      cur.c[j][0] = cur.c[j][3] = 1.f;
      cur.c[j][5] = 0.001f;
      cur.c[j][1] = cur.c[j][2] = cur.c[j][4] = 0.f;
    }
    nxt = cur;
  } else {
    load_source<PPT>(cur, points, covs, tile.begin, tile.count, 0);
  }

#pragma unroll
  for (int step = 0; step < ITERS; step++) {
    // ---- phase B: transform, voxel coordinate, cheap hash; key + record of the home slot requested together ----
    int cx[PPT], cy[PPT], cz[PPT];
    uint32_t slot[PPT];
    bool active[PPT];
    v4i key[PPT];
    v4f head[PPT];
    v2d c01[PPT], c23[PPT], c45[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      const int local = (step * PPT + j) * 256 + (int)threadIdx.x;
      active[j] = local < tile.count;
      const double px = (double)cur.px[j], py = (double)cur.py[j], pz = (double)cur.pz[j];
      const double lx = Tl.r00 * px + Tl.r01 * py + Tl.r02 * pz + Tl.tx;
      const double ly = Tl.r10 * px + Tl.r11 * py + Tl.r12 * pz + Tl.ty;
      const double lz = Tl.r20 * px + Tl.r21 * py + Tl.r22 * pz + Tl.tz;
      cx[j] = fast_floor(lx * f.map.inv_leaf);
      cy[j] = fast_floor(ly * f.map.inv_leaf);
      cz[j] = fast_floor(lz * f.map.inv_leaf);
      if (f.surface_validation && active[j] && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * ((size_t)tile.begin + local))) active[j] = false;
      slot[j] = coord_hash32(cx[j], cy[j], cz[j]) & pmask;
      if constexpr (ABLATE == 2) {
        key[j] = v4i{cx[j], cy[j], cz[j], (int)slot[j]};
        head[j] = v4f{0.1f, 0.05f, -0.02f, 1.f};
        c01[j] = v2d{1.0, 0.0};
        c23[j] = v2d{0.0, 1.0};
        c45[j] = v2d{0.0, 0.001};
      } else {
        key[j] = pkeys[slot[j]];
        const GP_GLOBAL char* rec = pfat + 64 * (size_t)slot[j];
        head[j] = *(const GP_GLOBAL v4f*)rec;
        c01[j] = *(const GP_GLOBAL v2d*)(rec + 16);
        c23[j] = *(const GP_GLOBAL v2d*)(rec + 32);
        c45[j] = *(const GP_GLOBAL v2d*)(rec + 48);
      }
    }
    // prefetch the NEXT step's source points now: vmcnt retires in issue order, so these (HBM-latency) loads must be
    // younger than the table loads above, or waiting for the table data would also wait for the prefetch
    if (step + 1 < ITERS && ABLATE != 3) load_source<PPT>(nxt, points, covs, tile.begin, tile.count, step + 1);
    // ---- phase C: resolve; the collision chain (rare at load factor <= 0.5) walks on and re-fetches the record ----
    bool hit[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      bool h = false;
      if (active[j]) {
        v4i k = key[j];
        uint32_t s = slot[j];
        bool moved = false;
        while (k.w >= 0) {
          if (k.x == cx[j] && k.y == cy[j] && k.z == cz[j]) {
            h = true;
            break;
          }
          s = (s + 1) & pmask;
          k = pkeys[s];
          moved = true;
        }
        if (h && moved) {
          const GP_GLOBAL char* rec = pfat + 64 * (size_t)s;
          head[j] = *(const GP_GLOBAL v4f*)rec;
          c01[j] = *(const GP_GLOBAL v2d*)(rec + 16);
          c23[j] = *(const GP_GLOBAL v2d*)(rec + 32);
          c45[j] = *(const GP_GLOBAL v2d*)(rec + 48);
        }
      }
      hit[j] = h;
    }
    // ---- phase D ----
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      if (!hit[j]) continue;
      if constexpr (ABLATE == 1) {
        acc[0] += (acc_t)(cur.px[j] + cur.c[j][0] + cur.c[j][1] + cur.c[j][2] + cur.c[j][3] + cur.c[j][4] + cur.c[j][5] + head[j].x) + (acc_t)(c01[j].x + c23[j].y + c45[j].y);
        continue;
      }
      const double px = (double)cur.px[j], py = (double)cur.py[j], pz = (double)cur.pz[j];
      double m[6];
      {
        const double a00 = (double)cur.c[j][0], a01 = (double)cur.c[j][1], a02 = (double)cur.c[j][2], a11 = (double)cur.c[j][3], a12 = (double)cur.c[j][4], a22 = (double)cur.c[j][5];
        const double rc00 = Tl.r00 * a00 + Tl.r01 * a01 + Tl.r02 * a02, rc01 = Tl.r00 * a01 + Tl.r01 * a11 + Tl.r02 * a12, rc02 = Tl.r00 * a02 + Tl.r01 * a12 + Tl.r02 * a22;
        const double rc10 = Tl.r10 * a00 + Tl.r11 * a01 + Tl.r12 * a02, rc11 = Tl.r10 * a01 + Tl.r11 * a11 + Tl.r12 * a12, rc12 = Tl.r10 * a02 + Tl.r11 * a12 + Tl.r12 * a22;
        const double rc20 = Tl.r20 * a00 + Tl.r21 * a01 + Tl.r22 * a02, rc21 = Tl.r20 * a01 + Tl.r21 * a11 + Tl.r22 * a12, rc22 = Tl.r20 * a02 + Tl.r21 * a12 + Tl.r22 * a22;
        const double s00 = c01[j].x + rc00 * Tl.r00 + rc01 * Tl.r01 + rc02 * Tl.r02;
        const double s01 = c01[j].y + rc00 * Tl.r10 + rc01 * Tl.r11 + rc02 * Tl.r12;
        const double s02 = c23[j].x + rc00 * Tl.r20 + rc01 * Tl.r21 + rc02 * Tl.r22;
        const double s11 = c23[j].y + rc10 * Tl.r10 + rc11 * Tl.r11 + rc12 * Tl.r12;
        const double s12 = c45[j].x + rc10 * Tl.r20 + rc11 * Tl.r21 + rc12 * Tl.r22;
        const double s22 = c45[j].y + rc20 * Tl.r20 + rc21 * Tl.r21 + rc22 * Tl.r22;
        const double i00 = s11 * s22 - s12 * s12, i01 = s02 * s12 - s01 * s22, i02 = s01 * s12 - s02 * s11;
        const double invdet = fast_rcp(s00 * i00 + s01 * i01 + s02 * i02);
        m[0] = i00 * invdet;
        m[1] = i01 * invdet;
        m[2] = i02 * invdet;
        m[3] = (s00 * s22 - s02 * s02) * invdet;
        m[4] = (s01 * s02 - s00 * s12) * invdet;
        m[5] = (s00 * s11 - s01 * s01) * invdet;
      }
      const double qx = Te.r00 * px + Te.r01 * py + Te.r02 * pz + Te.tx;
      const double qy = Te.r10 * px + Te.r11 * py + Te.r12 * pz + Te.ty;
      const double qz = Te.r20 * px + Te.r21 * py + Te.r22 * pz + Te.tz;
      const double rxd = (((double)cx[j] + 0.5) * f.map.leaf - qx) + (double)head[j].x;
      const double ryd = (((double)cy[j] + 0.5) * f.map.leaf - qy) + (double)head[j].y;
      const double rzd = (((double)cz[j] + 0.5) * f.map.leaf - qz) + (double)head[j].z;
      const acc_t M0 = (acc_t)m[0], M1 = (acc_t)m[1], M2 = (acc_t)m[2], M3 = (acc_t)m[3], M4 = (acc_t)m[4], M5 = (acc_t)m[5];
      const acc_t RX = (acc_t)rxd, RY = (acc_t)ryd, RZ = (acc_t)rzd, QX = (acc_t)qx, QY = (acc_t)qy, QZ = (acc_t)qz;
      const acc_t mrx = M0 * RX + M1 * RY + M2 * RZ, mry = M1 * RX + M3 * RY + M4 * RZ, mrz = M2 * RX + M4 * RY + M5 * RZ;
      acc[ACC_COUNT] += (acc_t)1;
      acc[ACC_ERR] += RX * mrx + RY * mry + RZ * mrz;
      if constexpr (MODE == MODE_LIN) {
        acc[ACC_M + 0] += M0;
        acc[ACC_M + 1] += M1;
        acc[ACC_M + 2] += M2;
        acc[ACC_M + 3] += M3;
        acc[ACC_M + 4] += M4;
        acc[ACC_M + 5] += M5;
        const acc_t k00 = M1 * QZ - M2 * QY, k01 = M2 * QX - M0 * QZ, k02 = M0 * QY - M1 * QX;
        const acc_t k10 = M3 * QZ - M4 * QY, k11 = M4 * QX - M1 * QZ, k12 = M1 * QY - M3 * QX;
        const acc_t k20 = M4 * QZ - M5 * QY, k21 = M5 * QX - M2 * QZ, k22 = M2 * QY - M4 * QX;
        acc[ACC_K + 0] += k00;
        acc[ACC_K + 1] += k01;
        acc[ACC_K + 2] += k02;
        acc[ACC_K + 3] += k10;
        acc[ACC_K + 4] += k11;
        acc[ACC_K + 5] += k12;
        acc[ACC_K + 6] += k20;
        acc[ACC_K + 7] += k21;
        acc[ACC_K + 8] += k22;
        acc[ACC_TL + 0] += QZ * k10 - QY * k20;
        acc[ACC_TL + 1] += QZ * k11 - QY * k21;
        acc[ACC_TL + 2] += QZ * k12 - QY * k22;
        acc[ACC_TL + 3] += QX * k21 - QZ * k01;
        acc[ACC_TL + 4] += QX * k22 - QZ * k02;
        acc[ACC_TL + 5] += QY * k02 - QX * k12;
        acc[ACC_QXMR + 0] += QY * mrz - QZ * mry;
        acc[ACC_QXMR + 1] += QZ * mrx - QX * mrz;
        acc[ACC_QXMR + 2] += QX * mry - QY * mrx;
        acc[ACC_MR + 0] += mrx;
        acc[ACC_MR + 1] += mry;
        acc[ACC_MR + 2] += mrz;
      }
    }
    if (step + 1 < ITERS) cur = nxt;
  }

  // ---- wavefront reduction in f64 (transposing butterfly), LDS across the 4 waves ----
  double accd[32];
#pragma unroll
  for (int k = 0; k < 32; k++) accd[k] = (double)acc[k];
  __shared__ double lds[4][ACC_STRIDE];
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = accd[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) lds[wave][k] = v;
    }
  } else {
    if constexpr (ABLATE == 4) {
      if (lane < 32) lds[wave][lane] = accd[0] + accd[lane & 1];
    } else {
      const double s = butterfly_reduce32(accd, lane);
      if ((lane & 1) == 0) lds[wave][butterfly_component(lane)] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double s = 0.0;
    if (threadIdx.x < NACC) s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = s;
  }
}


// =====================================================================================================================
// vgicp_tile_kernel4 -- software-pipelined version of kernel3 (one point per lane per step, ITERS steps per tile):
//   while step s is being computed, the table loads of step s+1 and the source loads of step s+2 are in flight.
//   vmcnt retires in issue order, so the issue order is pinned with compiler barriers: at the top of step s everything
//   outstanding (T_s, S_{s+1}) was issued a full compute phase earlier; then T_{s+1} and S_{s+2} are issued, then step s is
//   computed from registers.  Ablation of kernel3 showed source latency (9.0 us), table latency (8.1 us) and arithmetic
//   (2.7 us) adding up serially at 1 M points; this overlaps all three.
// =====================================================================================================================
#define GP_PIN_ORDER() asm volatile("" ::: "memory")

struct TableRegs {
  int cx, cy, cz;
  uint32_t slot;
  bool active;
  v4i key;
  v4f head;
  v2d c01, c23, c45;
};

template <int MODE, bool OUTER_F32, int ITERS>
__global__ void __launch_bounds__(256) vgicp_tile_kernel4(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                          const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                          double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  const int tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;  // XCD-aware workgroup -> tile map
  if (tile_idx >= num_tiles) return;
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
  } else {
    tile = tiles[tile_idx];
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const GP_GLOBAL float* points = as_global(f.points);
  const GP_GLOBAL float* covs = as_global(f.covs);
  const GP_GLOBAL v4i* pkeys = (const GP_GLOBAL v4i*)f.map.pkeys;
  const GP_GLOBAL char* pfat = (const GP_GLOBAL char*)f.map.pfat;
  const uint32_t pmask = f.map.pmask;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  // issue the table loads of one step from its (arrived) source registers
  auto issue_table = [&](const SourceRegs<1>& src, int step, TableRegs& t) {
    const int local = step * 256 + (int)threadIdx.x;
    t.active = local < tile.count;
    const double px = (double)src.px[0], py = (double)src.py[0], pz = (double)src.pz[0];
    const double lx = Tl.r00 * px + Tl.r01 * py + Tl.r02 * pz + Tl.tx;
    const double ly = Tl.r10 * px + Tl.r11 * py + Tl.r12 * pz + Tl.ty;
    const double lz = Tl.r20 * px + Tl.r21 * py + Tl.r22 * pz + Tl.tz;
    t.cx = fast_floor(lx * f.map.inv_leaf);
    t.cy = fast_floor(ly * f.map.inv_leaf);
    t.cz = fast_floor(lz * f.map.inv_leaf);
    if (f.surface_validation && t.active && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * ((size_t)tile.begin + local))) t.active = false;
    t.slot = coord_hash32(t.cx, t.cy, t.cz) & pmask;
    t.key = pkeys[t.slot];
    const GP_GLOBAL char* rec = pfat + 64 * (size_t)t.slot;
    t.head = *(const GP_GLOBAL v4f*)rec;
    t.c01 = *(const GP_GLOBAL v2d*)(rec + 16);
    t.c23 = *(const GP_GLOBAL v2d*)(rec + 32);
    t.c45 = *(const GP_GLOBAL v2d*)(rec + 48);
  };

  SourceRegs<1> s_cur, s_nxt, s_nn;
  TableRegs t_cur, t_nxt;
  // prologue: S_0 -> T_0 ; S_1
  load_source<1>(s_cur, points, covs, tile.begin, tile.count, 0);
  GP_PIN_ORDER();
  issue_table(s_cur, 0, t_cur);
  GP_PIN_ORDER();
  if (ITERS > 1) load_source<1>(s_nxt, points, covs, tile.begin, tile.count, 1);
  GP_PIN_ORDER();

#pragma unroll
  for (int step = 0; step < ITERS; step++) {
    // everything outstanding here (T_step, S_{step+1}) was issued one compute phase ago
    if (step + 1 < ITERS) {
      issue_table(s_nxt, step + 1, t_nxt);  // needs S_{step+1}; in-order vmcnt => T_step has arrived as well
      GP_PIN_ORDER();
      if (step + 2 < ITERS) load_source<1>(s_nn, points, covs, tile.begin, tile.count, step + 2);
      GP_PIN_ORDER();
    }
    // ---- resolve step `step` (collision chain is rare at load factor <= 0.5) ----
    bool hit = false;
    if (t_cur.active) {
      v4i k = t_cur.key;
      uint32_t s = t_cur.slot;
      bool moved = false;
      while (k.w >= 0) {
        if (k.x == t_cur.cx && k.y == t_cur.cy && k.z == t_cur.cz) {
          hit = true;
          break;
        }
        s = (s + 1) & pmask;
        k = pkeys[s];
        moved = true;
      }
      if (hit && moved) {
        const GP_GLOBAL char* rec = pfat + 64 * (size_t)s;
        t_cur.head = *(const GP_GLOBAL v4f*)rec;
        t_cur.c01 = *(const GP_GLOBAL v2d*)(rec + 16);
        t_cur.c23 = *(const GP_GLOBAL v2d*)(rec + 32);
        t_cur.c45 = *(const GP_GLOBAL v2d*)(rec + 48);
      }
    }
    if (hit) {
      const double px = (double)s_cur.px[0], py = (double)s_cur.py[0], pz = (double)s_cur.pz[0];
      double m[6];
      {
        const double a00 = (double)s_cur.c[0][0], a01 = (double)s_cur.c[0][1], a02 = (double)s_cur.c[0][2], a11 = (double)s_cur.c[0][3], a12 = (double)s_cur.c[0][4], a22 = (double)s_cur.c[0][5];
        const double rc00 = Tl.r00 * a00 + Tl.r01 * a01 + Tl.r02 * a02, rc01 = Tl.r00 * a01 + Tl.r01 * a11 + Tl.r02 * a12, rc02 = Tl.r00 * a02 + Tl.r01 * a12 + Tl.r02 * a22;
        const double rc10 = Tl.r10 * a00 + Tl.r11 * a01 + Tl.r12 * a02, rc11 = Tl.r10 * a01 + Tl.r11 * a11 + Tl.r12 * a12, rc12 = Tl.r10 * a02 + Tl.r11 * a12 + Tl.r12 * a22;
        const double rc20 = Tl.r20 * a00 + Tl.r21 * a01 + Tl.r22 * a02, rc21 = Tl.r20 * a01 + Tl.r21 * a11 + Tl.r22 * a12, rc22 = Tl.r20 * a02 + Tl.r21 * a12 + Tl.r22 * a22;
        const double s00 = t_cur.c01.x + rc00 * Tl.r00 + rc01 * Tl.r01 + rc02 * Tl.r02;
        const double s01 = t_cur.c01.y + rc00 * Tl.r10 + rc01 * Tl.r11 + rc02 * Tl.r12;
        const double s02 = t_cur.c23.x + rc00 * Tl.r20 + rc01 * Tl.r21 + rc02 * Tl.r22;
        const double s11 = t_cur.c23.y + rc10 * Tl.r10 + rc11 * Tl.r11 + rc12 * Tl.r12;
        const double s12 = t_cur.c45.x + rc10 * Tl.r20 + rc11 * Tl.r21 + rc12 * Tl.r22;
        const double s22 = t_cur.c45.y + rc20 * Tl.r20 + rc21 * Tl.r21 + rc22 * Tl.r22;
        const double i00 = s11 * s22 - s12 * s12, i01 = s02 * s12 - s01 * s22, i02 = s01 * s12 - s02 * s11;
        const double invdet = fast_rcp(s00 * i00 + s01 * i01 + s02 * i02);
        m[0] = i00 * invdet;
        m[1] = i01 * invdet;
        m[2] = i02 * invdet;
        m[3] = (s00 * s22 - s02 * s02) * invdet;
        m[4] = (s01 * s02 - s00 * s12) * invdet;
        m[5] = (s00 * s11 - s01 * s01) * invdet;
      }
      const double qx = Te.r00 * px + Te.r01 * py + Te.r02 * pz + Te.tx;
      const double qy = Te.r10 * px + Te.r11 * py + Te.r12 * pz + Te.ty;
      const double qz = Te.r20 * px + Te.r21 * py + Te.r22 * pz + Te.tz;
      const double rxd = (((double)t_cur.cx + 0.5) * f.map.leaf - qx) + (double)t_cur.head.x;
      const double ryd = (((double)t_cur.cy + 0.5) * f.map.leaf - qy) + (double)t_cur.head.y;
      const double rzd = (((double)t_cur.cz + 0.5) * f.map.leaf - qz) + (double)t_cur.head.z;
      const acc_t M0 = (acc_t)m[0], M1 = (acc_t)m[1], M2 = (acc_t)m[2], M3 = (acc_t)m[3], M4 = (acc_t)m[4], M5 = (acc_t)m[5];
      const acc_t RX = (acc_t)rxd, RY = (acc_t)ryd, RZ = (acc_t)rzd, QX = (acc_t)qx, QY = (acc_t)qy, QZ = (acc_t)qz;
      const acc_t mrx = M0 * RX + M1 * RY + M2 * RZ, mry = M1 * RX + M3 * RY + M4 * RZ, mrz = M2 * RX + M4 * RY + M5 * RZ;
      acc[ACC_COUNT] += (acc_t)1;
      acc[ACC_ERR] += RX * mrx + RY * mry + RZ * mrz;
      if constexpr (MODE == MODE_LIN) {
        acc[ACC_M + 0] += M0;
        acc[ACC_M + 1] += M1;
        acc[ACC_M + 2] += M2;
        acc[ACC_M + 3] += M3;
        acc[ACC_M + 4] += M4;
        acc[ACC_M + 5] += M5;
        const acc_t k00 = M1 * QZ - M2 * QY, k01 = M2 * QX - M0 * QZ, k02 = M0 * QY - M1 * QX;
        const acc_t k10 = M3 * QZ - M4 * QY, k11 = M4 * QX - M1 * QZ, k12 = M1 * QY - M3 * QX;
        const acc_t k20 = M4 * QZ - M5 * QY, k21 = M5 * QX - M2 * QZ, k22 = M2 * QY - M4 * QX;
        acc[ACC_K + 0] += k00;
        acc[ACC_K + 1] += k01;
        acc[ACC_K + 2] += k02;
        acc[ACC_K + 3] += k10;
        acc[ACC_K + 4] += k11;
        acc[ACC_K + 5] += k12;
        acc[ACC_K + 6] += k20;
        acc[ACC_K + 7] += k21;
        acc[ACC_K + 8] += k22;
        acc[ACC_TL + 0] += QZ * k10 - QY * k20;
        acc[ACC_TL + 1] += QZ * k11 - QY * k21;
        acc[ACC_TL + 2] += QZ * k12 - QY * k22;
        acc[ACC_TL + 3] += QX * k21 - QZ * k01;
        acc[ACC_TL + 4] += QX * k22 - QZ * k02;
        acc[ACC_TL + 5] += QY * k02 - QX * k12;
        acc[ACC_QXMR + 0] += QY * mrz - QZ * mry;
        acc[ACC_QXMR + 1] += QZ * mrx - QX * mrz;
        acc[ACC_QXMR + 2] += QX * mry - QY * mrx;
        acc[ACC_MR + 0] += mrx;
        acc[ACC_MR + 1] += mry;
        acc[ACC_MR + 2] += mrz;
      }
    }
    GP_PIN_ORDER();
    if (step + 1 < ITERS) {
      s_cur = s_nxt;
      t_cur = t_nxt;
      s_nxt = s_nn;
    }
  }

  double accd[32];
#pragma unroll
  for (int k = 0; k < 32; k++) accd[k] = (double)acc[k];
  __shared__ double lds[4][ACC_STRIDE];
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = accd[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) lds[wave][k] = v;
    }
  } else {
    const double s = butterfly_reduce32(accd, lane);
    if ((lane & 1) == 0) lds[wave][butterfly_component(lane)] = s;
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double s = 0.0;
    if (threadIdx.x < NACC) s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = s;
  }
}


// =====================================================================================================================
// vgicp_tile_kernel5 -- source stream staged through LDS with the gfx950 LDS-DMA (global_load_lds_dwordx4).
//
// Why: the ablations of kernel3 showed the pass is bound by memory-level parallelism, not arithmetic: in-flight loads
// cost VGPRs (9 per source point, 20 per voxel record), so a CU could keep only ~30 point-loads in flight (2.5 TB/s).
// LDS-DMA data never touches a VGPR while in flight: each wave requests its 256 points (3 KB) and covariances (9 KB) with
// 12 perfectly coalesced 1-KB instructions at kernel start, i.e. 36-48 KB outstanding per workgroup, ~10 MB per chip.
// The copy is wave-private (a wave DMAs exactly the 256 consecutive points its own lanes will process), so no barrier is
// needed: only the wave's own vmcnt.  Lanes then read their point (stride 3 dwords) and covariance (stride 9 dwords)
// from LDS -- both odd strides, bank-conflict free.  Table gathers (private slot table) stay register based, 4 in flight.
// A partial last wave of a factor falls back to per-lane loads + ds_write into the same LDS image (never reads past n).
// =====================================================================================================================
#define GP_LDS __attribute__((address_space(3)))

// per-correspondence algebra given the target mean mu_B in f64 (VGICP: voxel centre + offset; GICP: matched target point)
template <int MODE, typename acc_t>
__device__ __forceinline__ void accumulate_terms_mu(const Pose& Tl, const Pose& Te, float pxf, float pyf, float pzf, const float* cA, double mux, double muy, double muz,
                                                    const v2d& c01, const v2d& c23, const v2d& c45, acc_t* acc) {
  const double px = (double)pxf, py = (double)pyf, pz = (double)pzf;
  double m[6];
  {
    const double a00 = (double)cA[0], a01 = (double)cA[1], a02 = (double)cA[2], a11 = (double)cA[3], a12 = (double)cA[4], a22 = (double)cA[5];
    const double rc00 = Tl.r00 * a00 + Tl.r01 * a01 + Tl.r02 * a02, rc01 = Tl.r00 * a01 + Tl.r01 * a11 + Tl.r02 * a12, rc02 = Tl.r00 * a02 + Tl.r01 * a12 + Tl.r02 * a22;
    const double rc10 = Tl.r10 * a00 + Tl.r11 * a01 + Tl.r12 * a02, rc11 = Tl.r10 * a01 + Tl.r11 * a11 + Tl.r12 * a12, rc12 = Tl.r10 * a02 + Tl.r11 * a12 + Tl.r12 * a22;
    const double rc20 = Tl.r20 * a00 + Tl.r21 * a01 + Tl.r22 * a02, rc21 = Tl.r20 * a01 + Tl.r21 * a11 + Tl.r22 * a12, rc22 = Tl.r20 * a02 + Tl.r21 * a12 + Tl.r22 * a22;
    const double s00 = c01.x + rc00 * Tl.r00 + rc01 * Tl.r01 + rc02 * Tl.r02;
    const double s01 = c01.y + rc00 * Tl.r10 + rc01 * Tl.r11 + rc02 * Tl.r12;
    const double s02 = c23.x + rc00 * Tl.r20 + rc01 * Tl.r21 + rc02 * Tl.r22;
    const double s11 = c23.y + rc10 * Tl.r10 + rc11 * Tl.r11 + rc12 * Tl.r12;
    const double s12 = c45.x + rc10 * Tl.r20 + rc11 * Tl.r21 + rc12 * Tl.r22;
    const double s22 = c45.y + rc20 * Tl.r20 + rc21 * Tl.r21 + rc22 * Tl.r22;
    const double i00 = s11 * s22 - s12 * s12, i01 = s02 * s12 - s01 * s22, i02 = s01 * s12 - s02 * s11;
    const double invdet = fast_rcp(s00 * i00 + s01 * i01 + s02 * i02);
    m[0] = i00 * invdet;
    m[1] = i01 * invdet;
    m[2] = i02 * invdet;
    m[3] = (s00 * s22 - s02 * s02) * invdet;
    m[4] = (s01 * s02 - s00 * s12) * invdet;
    m[5] = (s00 * s11 - s01 * s01) * invdet;
  }
  const double qx = Te.r00 * px + Te.r01 * py + Te.r02 * pz + Te.tx;
  const double qy = Te.r10 * px + Te.r11 * py + Te.r12 * pz + Te.ty;
  const double qz = Te.r20 * px + Te.r21 * py + Te.r22 * pz + Te.tz;
  const double rxd = mux - qx, ryd = muy - qy, rzd = muz - qz;
  const acc_t M0 = (acc_t)m[0], M1 = (acc_t)m[1], M2 = (acc_t)m[2], M3 = (acc_t)m[3], M4 = (acc_t)m[4], M5 = (acc_t)m[5];
  const acc_t RX = (acc_t)rxd, RY = (acc_t)ryd, RZ = (acc_t)rzd, QX = (acc_t)qx, QY = (acc_t)qy, QZ = (acc_t)qz;
  const acc_t mrx = M0 * RX + M1 * RY + M2 * RZ, mry = M1 * RX + M3 * RY + M4 * RZ, mrz = M2 * RX + M4 * RY + M5 * RZ;
  acc[ACC_COUNT] += (acc_t)1;
  acc[ACC_ERR] += RX * mrx + RY * mry + RZ * mrz;
  if constexpr (MODE == MODE_LIN) {
    acc[ACC_M + 0] += M0;
    acc[ACC_M + 1] += M1;
    acc[ACC_M + 2] += M2;
    acc[ACC_M + 3] += M3;
    acc[ACC_M + 4] += M4;
    acc[ACC_M + 5] += M5;
    const acc_t k00 = M1 * QZ - M2 * QY, k01 = M2 * QX - M0 * QZ, k02 = M0 * QY - M1 * QX;
    const acc_t k10 = M3 * QZ - M4 * QY, k11 = M4 * QX - M1 * QZ, k12 = M1 * QY - M3 * QX;
    const acc_t k20 = M4 * QZ - M5 * QY, k21 = M5 * QX - M2 * QZ, k22 = M2 * QY - M4 * QX;
    acc[ACC_K + 0] += k00;
    acc[ACC_K + 1] += k01;
    acc[ACC_K + 2] += k02;
    acc[ACC_K + 3] += k10;
    acc[ACC_K + 4] += k11;
    acc[ACC_K + 5] += k12;
    acc[ACC_K + 6] += k20;
    acc[ACC_K + 7] += k21;
    acc[ACC_K + 8] += k22;
    acc[ACC_TL + 0] += QZ * k10 - QY * k20;
    acc[ACC_TL + 1] += QZ * k11 - QY * k21;
    acc[ACC_TL + 2] += QZ * k12 - QY * k22;
    acc[ACC_TL + 3] += QX * k21 - QZ * k01;
    acc[ACC_TL + 4] += QX * k22 - QZ * k02;
    acc[ACC_TL + 5] += QY * k02 - QX * k12;
    acc[ACC_QXMR + 0] += QY * mrz - QZ * mry;
    acc[ACC_QXMR + 1] += QZ * mrx - QX * mrz;
    acc[ACC_QXMR + 2] += QX * mry - QY * mrx;
    acc[ACC_MR + 0] += mrx;
    acc[ACC_MR + 1] += mry;
    acc[ACC_MR + 2] += mrz;
  }
}


template <int MODE, typename acc_t>
__device__ __forceinline__ void accumulate_terms(const Pose& Tl, const Pose& Te, double leaf, float pxf, float pyf, float pzf, const float* cA, int cx, int cy, int cz,
                                                 const v4f& head, const v2d& c01, const v2d& c23, const v2d& c45, acc_t* acc) {
  // mu_B = voxel centre + f32 offset; (centre - q) is formed first so that the large coordinates cancel in f64
  accumulate_terms_mu<MODE, acc_t>(Tl, Te, pxf, pyf, pzf, cA, ((double)cx + 0.5) * leaf + (double)head.x, ((double)cy + 0.5) * leaf + (double)head.y,
                                   ((double)cz + 0.5) * leaf + (double)head.z, c01, c23, c45, acc);
}

// optional per-workgroup phase timestamps (s_memtime) for timeline analysis: [num_tiles][8] uint64, enabled by the host
static __device__ unsigned long long* g_trace = nullptr;
#define GP_TRACE(slot)                                                                   \
  do {                                                                                   \
    if (trace && threadIdx.x == 0) trace[(size_t)tile_idx * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

constexpr int kWavePoints = 256;                      // points one wave stages and processes (4 per lane)
constexpr int kWaveLdsBytes = kWavePoints * (12 + 36);  // 12 KB: [256][3] floats then [256][9] floats

template <int MODE, bool OUTER_F32>
__global__ void __launch_bounds__(256) vgicp_tile_kernel5(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                          const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                          double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  constexpr int PPT = 4;
  __shared__ __attribute__((aligned(16))) char smem[4 * kWaveLdsBytes];
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  const int tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;  // XCD-aware workgroup -> tile map
  if (tile_idx >= num_tiles) return;
  unsigned long long* trace = g_trace;
  GP_TRACE(0);
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
  } else {
    tile = tiles[tile_idx];
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const GP_GLOBAL float* points = as_global(f.points);
  const GP_GLOBAL float* covs = as_global(f.covs);
  const size_t first = (size_t)tile.begin + (size_t)wave * kWavePoints;  // the wave's first point
  int wcount = tile.count - wave * kWavePoints;
  wcount = wcount < 0 ? 0 : (wcount > kWavePoints ? kWavePoints : wcount);
  char* wbase = smem + wave * kWaveLdsBytes;
  float* lds_p = reinterpret_cast<float*>(wbase);
  float* lds_c = reinterpret_cast<float*>(wbase + kWavePoints * 12);

  // ---- stage the wave's source slice into LDS ----
  if (wcount == kWavePoints) {
    const GP_GLOBAL char* gp = (const GP_GLOBAL char*)(points + 3 * first) + lane * 16;
    const GP_GLOBAL char* gc = (const GP_GLOBAL char*)(covs + 9 * first) + lane * 16;
#pragma unroll
    for (int k = 0; k < 3; k++) __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)(gp + k * 1024), (GP_LDS void*)(wbase + k * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 9; k++) __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)(gc + k * 1024), (GP_LDS void*)(wbase + 3072 + k * 1024), 16, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      const int idx = j * 64 + lane;
      if (idx < wcount) {
        const GP_GLOBAL float* pp = points + 3 * (first + idx);
        const GP_GLOBAL float* cp = covs + 9 * (first + idx);
        lds_p[3 * idx] = pp[0];
        lds_p[3 * idx + 1] = pp[1];
        lds_p[3 * idx + 2] = pp[2];
#pragma unroll
        for (int k = 0; k < 9; k++) lds_c[9 * idx + k] = cp[k];
      }
    }
  }
  GP_TRACE(1);  // descriptors loaded, DMA issued
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const GP_GLOBAL v4i* pkeys = (const GP_GLOBAL v4i*)f.map.pkeys;
  const GP_GLOBAL char* pfat = (const GP_GLOBAL char*)f.map.pfat;
  const uint32_t pmask = f.map.pmask;
  // the LDS-DMA writes retire on vmcnt; the data is wave-private, so no workgroup barrier is needed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  GP_TRACE(2);  // source slice in LDS

  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  // ---- phase B: read the 4 points of this lane from LDS, hash, request key + record of the home slot ----
  float px[PPT], py[PPT], pz[PPT];
  int cx[PPT], cy[PPT], cz[PPT];
  uint32_t slot[PPT];
  bool active[PPT];
  v4i key[PPT];
  v4f head[PPT];
  v2d c01[PPT], c23[PPT], c45[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    const int idx = j * 64 + lane;
    active[j] = idx < wcount;
    const int ri = active[j] ? idx : 0;
    px[j] = lds_p[3 * ri];
    py[j] = lds_p[3 * ri + 1];
    pz[j] = lds_p[3 * ri + 2];
    const double dx = (double)px[j], dy = (double)py[j], dz = (double)pz[j];
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    cx[j] = fast_floor(lx * f.map.inv_leaf);
    cy[j] = fast_floor(ly * f.map.inv_leaf);
    cz[j] = fast_floor(lz * f.map.inv_leaf);
    if (f.surface_validation && active[j] && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (first + idx))) active[j] = false;
    slot[j] = coord_hash32(cx[j], cy[j], cz[j]) & pmask;
    key[j] = pkeys[slot[j]];
    const GP_GLOBAL char* rec = pfat + 64 * (size_t)slot[j];
    head[j] = *(const GP_GLOBAL v4f*)rec;
    c01[j] = *(const GP_GLOBAL v2d*)(rec + 16);
    c23[j] = *(const GP_GLOBAL v2d*)(rec + 32);
    c45[j] = *(const GP_GLOBAL v2d*)(rec + 48);
  }
  GP_TRACE(3);  // table loads issued
  if (trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_TRACE(4);  // table data arrived
  }
  // ---- phase C/D ----
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    bool hit = false;
    if (active[j]) {
      v4i k = key[j];
      uint32_t s = slot[j];
      bool moved = false;
      while (k.w >= 0) {
        if (k.x == cx[j] && k.y == cy[j] && k.z == cz[j]) {
          hit = true;
          break;
        }
        s = (s + 1) & pmask;
        k = pkeys[s];
        moved = true;
      }
      if (hit && moved) {
        const GP_GLOBAL char* rec = pfat + 64 * (size_t)s;
        head[j] = *(const GP_GLOBAL v4f*)rec;
        c01[j] = *(const GP_GLOBAL v2d*)(rec + 16);
        c23[j] = *(const GP_GLOBAL v2d*)(rec + 32);
        c45[j] = *(const GP_GLOBAL v2d*)(rec + 48);
      }
    }
    if (hit) {
      const int idx = j * 64 + lane;
      const float* cp = lds_c + 9 * idx;
      const float cA[6] = {cp[0], cp[3], cp[6], cp[4], cp[7], cp[8]};
      accumulate_terms<MODE, acc_t>(Tl, Te, f.map.leaf, px[j], py[j], pz[j], cA, cx[j], cy[j], cz[j], head[j], c01[j], c23[j], c45[j], acc);
    }
  }

  GP_TRACE(5);  // arithmetic done
  // ---- reduction: butterfly within the wave, then across the 4 waves through LDS (the staging area is free now) ----
  double accd[32];
#pragma unroll
  for (int k = 0; k < 32; k++) accd[k] = (double)acc[k];
  __syncthreads();
  double(*lds)[ACC_STRIDE] = reinterpret_cast<double(*)[ACC_STRIDE]>(smem);
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = accd[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) lds[wave][k] = v;
    }
  } else {
    const double s = butterfly_reduce32(accd, lane);
    if ((lane & 1) == 0) lds[wave][butterfly_component(lane)] = s;
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double s = 0.0;
    if (threadIdx.x < NACC) s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = s;
  }
  GP_TRACE(6);
}


// =====================================================================================================================
// vgicp_tile_kernel6 -- occupancy-first variant: one point per lane per step, NO register accumulators.
// The 29 per-point terms (f32 outer products on f64-accurate M, r, q) are added straight into a per-thread column of
// an LDS accumulator image [32][256] floats with ds_add_f32 (no return value, conflict-free: consecutive lanes ->
// consecutive banks).  That frees the 58 VGPRs of the f64 accumulators (and the butterfly), so 5-6 waves per SIMD stay
// resident and plain thread-level parallelism hides the two dependent round trips (source point, slot table).
// =====================================================================================================================
template <int MODE, int STEPS>
__global__ void __launch_bounds__(256) vgicp_tile_kernel6(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                          const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                          double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  __shared__ float lacc[32][256];  // 32 KB
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  const int tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;
  if (tile_idx >= num_tiles) return;
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
  } else {
    tile = tiles[tile_idx];
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const GP_GLOBAL float* points = as_global(f.points);
  const GP_GLOBAL float* covs = as_global(f.covs);
  const GP_GLOBAL char* pkeys = (const GP_GLOBAL char*)f.map.pkeys;
  const GP_GLOBAL char* pfat = (const GP_GLOBAL char*)f.map.pfat;
  const int kshift = f.map.pwide ? 7 : 4, rshift = f.map.pwide ? 7 : 6;
  const uint32_t pmask = f.map.pmask;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NACC; k++) lacc[k][tid] = 0.0f;  // own column only: no barrier needed before the adds

  for (int step = 0; step < STEPS; step++) {
    const int local = step * 256 + tid;
    if (local >= tile.count) break;
    const size_t i = (size_t)tile.begin + local;
    const GP_GLOBAL float* pp = points + 3 * i;
    const GP_GLOBAL float* cp = covs + 9 * i;
    const float px = __builtin_nontemporal_load(pp), py = __builtin_nontemporal_load(pp + 1), pz = __builtin_nontemporal_load(pp + 2);
    const float cA[6] = {__builtin_nontemporal_load(cp),     __builtin_nontemporal_load(cp + 3), __builtin_nontemporal_load(cp + 6),
                         __builtin_nontemporal_load(cp + 4), __builtin_nontemporal_load(cp + 7), __builtin_nontemporal_load(cp + 8)};
    const double dx = (double)px, dy = (double)py, dz = (double)pz;
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    if (f.surface_validation && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * i)) continue;
    const int cx = fast_floor(lx * f.map.inv_leaf), cy = fast_floor(ly * f.map.inv_leaf), cz = fast_floor(lz * f.map.inv_leaf);
    uint32_t s = coord_hash32(cx, cy, cz) & pmask;
    v4i key = *(const GP_GLOBAL v4i*)(pkeys + ((size_t)s << kshift));
    const GP_GLOBAL char* rec = pfat + ((size_t)s << rshift);
    v4f head = *(const GP_GLOBAL v4f*)rec;
    v2d c01 = *(const GP_GLOBAL v2d*)(rec + 16), c23 = *(const GP_GLOBAL v2d*)(rec + 32), c45 = *(const GP_GLOBAL v2d*)(rec + 48);
    bool hit = false, moved = false;
    while (key.w >= 0) {
      if (key.x == cx && key.y == cy && key.z == cz) {
        hit = true;
        break;
      }
      s = (s + 1) & pmask;
      key = *(const GP_GLOBAL v4i*)(pkeys + ((size_t)s << kshift));
      moved = true;
    }
    if (!hit) continue;
    if (moved) {
      rec = pfat + ((size_t)s << rshift);
      head = *(const GP_GLOBAL v4f*)rec;
      c01 = *(const GP_GLOBAL v2d*)(rec + 16);
      c23 = *(const GP_GLOBAL v2d*)(rec + 32);
      c45 = *(const GP_GLOBAL v2d*)(rec + 48);
    }
    float t[32];
#pragma unroll
    for (int k = 0; k < NACC; k++) t[k] = 0.0f;
    accumulate_terms<MODE, float>(Tl, Te, f.map.leaf, px, py, pz, cA, cx, cy, cz, head, c01, c23, c45, t);
#pragma unroll
    for (int k = 0; k < NACC; k++) lacc[k][tid] += t[k];  // own column: plain read-modify-write
  }
  __syncthreads();
  // column sums: 8 lanes per component, 32 values each, then a 3-step shuffle
  const int comp = tid >> 3, part = tid & 7;
  double sum = 0.0;
  if (comp < NACC) {
#pragma unroll 8
    for (int j = 0; j < 32; j++) sum += (double)lacc[comp][part * 32 + j];
  }
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  sum += __shfl_xor(sum, 4, 64);
  if (part == 0) ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + comp] = comp < NACC ? sum : 0.0;
}


// =====================================================================================================================
// vgicp_tile_kernel7 -- rolling LDS-DMA source pipeline, one 64-point chunk per wave step.
//
// Measured background (scripts/stream_bench.py, scripts/alu_rate.py): reading the source (48 B/point) and gathering the
// L2-resident voxel records take 8 us + 3.6 us for 1 M points and ADD UP in every register-staged variant, as does the
// ~8-10 us of f64 arithmetic: a wave that is hashing or multiplying has no source bytes in flight, and with 16 waves per
// CU the HBM pipe only stays full while every one of them is waiting on it.  Here the source of chunk j+2 is requested
// (LDS-DMA: no VGPRs while in flight) before the gather and the arithmetic of chunk j are done, so a wave always has two
// chunks (6 KB) in flight -- ~100 KB per CU -- whatever else it is doing.  LDS: 3 stages x 3 KB per wave = 36 KB per
// workgroup, 4 workgroups per CU.  vmcnt retires in order; the issue order gather(j) -> DMA(j+2) makes "vmcnt <= 4" mean
// "gather(j) has landed" for the compiler's own wait and "chunk j+1 has landed" at the top of the next step.
// =====================================================================================================================
constexpr int kChunkPoints = 64;
constexpr int kChunkBytes = kChunkPoints * 48;  // [64][3] floats, then [64][9] floats
constexpr int kChunkDmaOps = 3;                 // 3 x 64 lanes x 16 B: pieces 0..47 are the points, 48..191 the covariances

// request one 64-point chunk (768 B of points + 2304 B of covariances) into an LDS stage: three full-wave 16-B DMA
// instructions, no masked lanes and no branches, so the in-order vmcnt arithmetic around them stays static
__device__ __forceinline__ void chunk_dma(const GP_GLOBAL float* points, const GP_GLOBAL float* covs, size_t first_point, char* stage, int lane) {
  const GP_GLOBAL char* gp = (const GP_GLOBAL char*)(points + 3 * first_point);
  const GP_GLOBAL char* gc = (const GP_GLOBAL char*)(covs + 9 * first_point);
  const GP_GLOBAL char* a0 = lane < 48 ? gp + lane * 16 : gc + (lane - 48) * 16;
  __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)a0, (GP_LDS void*)stage, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)(gc + (lane + 16) * 16), (GP_LDS void*)(stage + 1024), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)(gc + (lane + 80) * 16), (GP_LDS void*)(stage + 2048), 16, 0, 0);
}

// voxel gather with hand-placed waits: the five 16-B loads (key, then the 64-B record) are issued from inline asm so that the
// compiler does not track them; gather_wait() ties the destination registers to the s_waitcnt so no use can be scheduled early
__device__ __forceinline__ void gather_issue(const GP_GLOBAL v4i* kp, const GP_GLOBAL char* rec, v4i& key, v4f& head, v2d& c01, v2d& c23, v2d& c45) {
  asm volatile(
    "global_load_dwordx4 %0, %5, off\n\t"
    "global_load_dwordx4 %1, %6, off\n\t"
    "global_load_dwordx4 %2, %6, off offset:16\n\t"
    "global_load_dwordx4 %3, %6, off offset:32\n\t"
    "global_load_dwordx4 %4, %6, off offset:48"
    : "=&v"(key), "=&v"(head), "=&v"(c01), "=&v"(c23), "=&v"(c45)
    : "v"(kp), "v"(rec)
    : "memory");
}
__device__ __forceinline__ void gather_wait(v4i& key, v4f& head, v2d& c01, v2d& c23, v2d& c45) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(key), "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
}

template <int MODE, bool OUTER_F32, int PPT>
__global__ void __launch_bounds__(256, 4) vgicp_tile_kernel7(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                          const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                          double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  constexpr int STAGES = 3;
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGES * kChunkBytes];  // 36 KB
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  const int tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;  // XCD-aware workgroup -> tile map
  if (tile_idx >= num_tiles) return;
  unsigned long long* trace = g_trace;
  GP_TRACE(0);
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
  } else {
    tile = tiles[tile_idx];
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const GP_GLOBAL float* points = as_global(f.points);
  const GP_GLOBAL float* covs = as_global(f.covs);
  const size_t first = (size_t)tile.begin + (size_t)wave * (PPT * kChunkPoints);  // the wave's first point
  int wcount = __builtin_amdgcn_readfirstlane(tile.count) - wave * (PPT * kChunkPoints);
  wcount = wcount < 0 ? 0 : (wcount > PPT * kChunkPoints ? PPT * kChunkPoints : wcount);
  // the DMA ring needs 16-B aligned rows; anything else (a partial wave, an odd base pointer) reads its points directly
  const bool ring = wcount == PPT * kChunkPoints && (((uintptr_t)f.points | (uintptr_t)f.covs) & 15) == 0;
  char* wbase = smem + wave * (STAGES * kChunkBytes);

  if (ring) {
    chunk_dma(points, covs, first, wbase, lane);
    if (PPT > 1) chunk_dma(points, covs, first + kChunkPoints, wbase + kChunkBytes, lane);
  }

  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const GP_GLOBAL v4i* pkeys = (const GP_GLOBAL v4i*)f.map.pkeys;
  const GP_GLOBAL char* pfat = (const GP_GLOBAL char*)f.map.pfat;
  const uint32_t pmask = f.map.pmask;

  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  // one chunk: hash, gather (home slot speculatively, then the rare probe), [next DMA], algebra
  auto step = [&](auto ring_tag, int j, bool active, float px, float py, float pz, const float* cA) {
    constexpr bool RING = decltype(ring_tag)::value;
    const double dx = (double)px, dy = (double)py, dz = (double)pz;
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    const int cx = fast_floor(lx * f.map.inv_leaf), cy = fast_floor(ly * f.map.inv_leaf), cz = fast_floor(lz * f.map.inv_leaf);
    bool live = active;
    if (f.surface_validation && live && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (first + (size_t)j * kChunkPoints + lane))) live = false;
    uint32_t s = coord_hash32(cx, cy, cz) & pmask;
    // home slot: key + record requested together with hand-placed waits (the compiler's own vmcnt bookkeeping would fold the
    // younger DMA requests into the wait for these loads)
    v4i key;
    v4f head;
    v2d c01, c23, c45;
    gather_issue(pkeys + s, pfat + 64 * (size_t)s, key, head, c01, c23, c45);
    gather_wait(key, head, c01, c23, c45);
    if (j == 0) GP_TRACE(2);
    if (j == 1) GP_TRACE(4);
    bool hit = false, moved = false;
    if (live) {
      while (key.w >= 0) {
        if (key.x == cx && key.y == cy && key.z == cz) {
          hit = true;
          break;
        }
        s = (s + 1) & pmask;
        key = pkeys[s];
        moved = true;
      }
      if (hit && moved) {
        gather_issue(pkeys + s, pfat + 64 * (size_t)s, key, head, c01, c23, c45);
        gather_wait(key, head, c01, c23, c45);
      }
    }
    if constexpr (RING) {
      // the probe is settled (its waits are behind us): request chunk j+2 -- its stage held chunk j-1, consumed a step ago --
      // so that it travels while this chunk's algebra runs; it is older than gather(j+1), hence landed before step j+2
      __builtin_amdgcn_sched_barrier(0);
      if (j + 2 < PPT) chunk_dma(points, covs, first + (size_t)(j + 2) * kChunkPoints, wbase + ((j + 2) % STAGES) * kChunkBytes, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (hit) accumulate_terms<MODE, acc_t>(Tl, Te, f.map.leaf, px, py, pz, cA, cx, cy, cz, head, c01, c23, c45, acc);
  };

  if (ring) {
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      // only chunk j+1's request may be younger than chunk j (normally already satisfied: step j-1 waited for its gather,
      // which is younger than chunk j -- but a step whose lanes were all rejected waits for nothing)
      if (j + 1 < PPT) {
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (j == 0) GP_TRACE(1);
      if (j == 1) GP_TRACE(3);
      if (j == 2) GP_TRACE(5);
      const float* lp = reinterpret_cast<const float*>(wbase + (j % STAGES) * kChunkBytes);
      const float* lc = lp + kChunkPoints * 3;
      const float px = lp[3 * lane], py = lp[3 * lane + 1], pz = lp[3 * lane + 2];
      const float cA[6] = {lc[9 * lane], lc[9 * lane + 3], lc[9 * lane + 6], lc[9 * lane + 4], lc[9 * lane + 7], lc[9 * lane + 8]};
      step(std::true_type{}, j, true, px, py, pz, cA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int j = 0; j < PPT; j++) {
      const int nj = wcount - j * kChunkPoints;  // wave-uniform
      if (nj <= 0) break;
      const bool active = lane < nj;
      const size_t i = first + (size_t)j * kChunkPoints + (active ? lane : 0);
      const GP_GLOBAL float* pp = points + 3 * i;
      const GP_GLOBAL float* cp = covs + 9 * i;
      const float cA[6] = {cp[0], cp[3], cp[6], cp[4], cp[7], cp[8]};
      step(std::false_type{}, j, active, pp[0], pp[1], pp[2], cA);
    }
  }

  GP_TRACE(6);
  // ---- reduction: butterfly within the wave, then across the 4 waves through LDS (the ring is drained) ----
  double accd[32];
#pragma unroll
  for (int k = 0; k < 32; k++) accd[k] = (double)acc[k];
  __syncthreads();
  double(*lds)[ACC_STRIDE] = reinterpret_cast<double(*)[ACC_STRIDE]>(smem);
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = accd[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) lds[wave][k] = v;
    }
  } else {
    const double sum = butterfly_reduce32(accd, lane);
    if ((lane & 1) == 0) lds[wave][butterfly_component(lane)] = sum;
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double sum = 0.0;
    if (threadIdx.x < NACC) sum = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = sum;
  }
  GP_TRACE(7);
}

}  // namespace gp
