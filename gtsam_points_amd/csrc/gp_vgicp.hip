// gp_vgicp.hip -- fused VGICP linearisation / error evaluation for gfx950, single-factor and batched.
//
// Replaces (reference):
//   include/gtsam_points/cuda/kernels/{lookup_voxels,vgicp_derivatives,linearized_system}.cuh  (device functors)
//   src/gtsam_points/factors/integrated_vgicp_derivatives{,_linearize,_compute,_inliers}.cu       (CUB reduce / select)
//   the per-factor launch loop of src/gtsam_points/cuda/nonlinear_factor_set_gpu.cpp:64-218
//
// Shape of the computation:
//   main kernel    : one 256-thread workgroup per TILE of kTilePoints consecutive source points of one factor.
//                    per point: q = R p + t (f64) -> floor -> hash probe (16-B bucket) -> 64-B voxel record ->
//                    M = (C_B + R C_A R^T)^-1 -> 29 target-side sums; 64-lane shuffle reduce, LDS cross-wave
//                    reduce, ONE 32-double partial per tile.  No atomics, no inlier list, no CUB temp storage.
//   finalize kernel: one workgroup per factor sums its tiles' partials in a fixed order (deterministic) and
//                    expands them to the LinearizedSystem6 blocks through the adjoint identity
//                    J_s = -J_t Ad(delta):  H_s = Ad^T H_t Ad, H_ts = -H_t Ad, b_s = -Ad^T b_t.
//   The workgroup -> tile map is XCD-aware: workgroup b runs on XCD b % 8, so XCD x is handed the x-th contiguous
//   eighth of the tile list and the voxel tables of "its" factors stay in that XCD's 4 MiB L2.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "gp_host.hpp"
#include "gp_vgicp_tile.hpp"
#include "gp_vgicp_tile2.hpp"
#include "gp_vgicp_finalize.hpp"
#include "gp_vgicp_stream.hpp"

namespace gp {



template <int MODE>
struct ModeTraits {
  static constexpr int kAcc = MODE == MODE_ERR ? 2 : (MODE == MODE_LIN ? ACC_SIZE : ACCG_SIZE);
  static constexpr int kStride = MODE == MODE_LIN_GENERAL ? ACCG_STRIDE : ACC_STRIDE;
};

template <int MODE>
__device__ __forceinline__ void accumulate_point(const FactorDesc& f, const Pose& Tl, const Pose& Te, int i, double* acc) {
  const float* __restrict__ pp = f.points + 3 * (size_t)i;
  const double px = (double)pp[0], py = (double)pp[1], pz = (double)pp[2];
  if (!finite3(px, py, pz)) return;  // a NaN / inf return has no voxel
  // correspondence at the LINEARISATION pose (lookup kernel is built with d_xl, integrated_vgicp_derivatives_compute.cu:25)
  const double lx = Tl.r00 * px + Tl.r01 * py + Tl.r02 * pz + Tl.tx;
  const double ly = Tl.r10 * px + Tl.r11 * py + Tl.r12 * pz + Tl.ty;
  const double lz = Tl.r20 * px + Tl.r21 * py + Tl.r22 * pz + Tl.tz;
  if (f.surface_validation && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (size_t)i)) return;
  const int cx = fast_floor(lx * f.map.inv_leaf), cy = fast_floor(ly * f.map.inv_leaf), cz = fast_floor(lz * f.map.inv_leaf);
  const int v = lookup_voxel(f.map, cx, cy, cz);
  if (v < 0) return;

  // one 64-B gather: {mean_local f32x3, n, cov f64x6}
  const VoxelRecord* __restrict__ rec = f.map.records + v;
  const float4 head = *reinterpret_cast<const float4*>(rec);
  const double2 c01 = *reinterpret_cast<const double2*>(rec->cov);
  const double2 c23 = *reinterpret_cast<const double2*>(rec->cov + 2);
  const double2 c45 = *reinterpret_cast<const double2*>(rec->cov + 4);
  const double cb[6] = {c01.x, c01.y, c23.x, c23.y, c45.x, c45.y};

  const float* __restrict__ cp = f.covs + 9 * (size_t)i;
  // symmetric part of the column-major 3x3 (exactly the input when it is symmetric)
  const double ca[6] = {(double)cp[0], 0.5 * ((double)cp[3] + (double)cp[1]), 0.5 * ((double)cp[6] + (double)cp[2]),
                        (double)cp[4], 0.5 * ((double)cp[7] + (double)cp[5]), (double)cp[8]};

  double m[6];
  fused_mahalanobis(Tl, ca, cb, m);

  double ox, oy, oz;
  voxel_center(f.map, cx, cy, cz, ox, oy, oz);
  double qx, qy, qz;
  if constexpr (MODE == MODE_ERR) {
    qx = Te.r00 * px + Te.r01 * py + Te.r02 * pz + Te.tx;
    qy = Te.r10 * px + Te.r11 * py + Te.r12 * pz + Te.ty;
    qz = Te.r20 * px + Te.r21 * py + Te.r22 * pz + Te.tz;
  } else {
    qx = lx;
    qy = ly;
    qz = lz;
  }
  // r = mu_B - q, with mu_B = centre + mean_local
  const double rx = (ox - qx) + (double)head.x;
  const double ry = (oy - qy) + (double)head.y;
  const double rz = (oz - qz) + (double)head.z;
  accumulate_sums<MODE>(Tl, m, px, py, pz, qx, qy, qz, rx, ry, rz, acc);
}

// main kernel: one workgroup per tile; writes partials[tile][kStride]
template <int MODE>
__global__ void __launch_bounds__(kBlockThreads) vgicp_tile_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                                   const double* __restrict__ poses_lin, const double* __restrict__ poses_eval,
                                                                   const InlinePoses inl, double* __restrict__ partials) {
  constexpr int NACC = ModeTraits<MODE>::kAcc;
  constexpr int STRIDE = ModeTraits<MODE>::kStride;
  const int tile_idx = xcd_swizzle(blockIdx.x, num_tiles);
  if (tile_idx >= num_tiles) return;
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    if (inl.tile_points > 0) {
      tile.begin = tile_idx * inl.tile_points;
      tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
    } else {  // the table of a single-factor stream batch is its plan (gp_vgicp_stream.hpp): tiles of different sizes, XCD-major
      const int px = tile_idx / inl.plan.wgs_per_xcd, pq = tile_idx % inl.plan.wgs_per_xcd;
      plan_tile(plan_fields(inl.plan, px, pq), px, pq, &tile.begin, &tile.count);
    }
    tile.row = tile_idx;
  } else {
    tile = tiles[tile_idx];
  }
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;

  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.0;

  for (int local = threadIdx.x; local < tile.count; local += kBlockThreads) accumulate_point<MODE>(f, Tl, Te, tile.begin + local, acc);

  // 64-lane wavefront reduction
#pragma unroll
  for (int k = 0; k < NACC; k++) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    acc[k] = v;
  }
  __shared__ double lds[kBlockThreads / 64][STRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NACC; k++) lds[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < STRIDE) {
    double s = 0.0;
    if (threadIdx.x < NACC) {
#pragma unroll
      for (int w = 0; w < kBlockThreads / 64; w++) s += lds[w][threadIdx.x];
    }
    partials[(size_t)tile.row * STRIDE + threadIdx.x] = s;
  }
}

// finalize: one workgroup per factor; deterministic ordered sum of the factor's tile partials, then expansion to the
// LinearizedSystem6 blocks.  GENERAL = false: source-side blocks through the adjoint identity; true: read directly.
constexpr int kFinalizeThreads = 1024;

template <bool GENERAL>
__global__ void __launch_bounds__(kFinalizeThreads) vgicp_finalize_kernel(const FactorDesc* __restrict__ factors, const double* __restrict__ poses,
                                                                          const InlinePoses inl, const double* __restrict__ partials,
                                                                          gp_linearized6* __restrict__ out, const DoneFlags done) {
  constexpr int STRIDE = GENERAL ? ACCG_STRIDE : ACC_STRIDE;
  constexpr int NACC = GENERAL ? ACCG_SIZE : ACC_SIZE;
  constexpr int kSlices = kFinalizeThreads / STRIDE;  // 1024 threads = 32 slices x 32 sums, or 10 x 96 (+ idle)
  const int fi = blockIdx.x;
  const int tile_begin = inl.use ? inl.factor.tile_begin : factors[fi].tile_begin;
  const int tile_count = inl.use ? inl.factor.tile_count : factors[fi].tile_count;
  __shared__ double lds[kSlices][STRIDE];
  __shared__ double sum[STRIDE];
  __shared__ double Ht[6][6], Ad[6][6], HtA[6][6], bt[6];
  GP_FIN_TRACE(0);
  if (done.trace && threadIdx.x == 0 && blockIdx.x == 0) done.trace[9] = __builtin_amdgcn_s_getreg(GP_GETREG_XCC_ID);
  const int comp = threadIdx.x % STRIDE, slice = threadIdx.x / STRIDE;
  if (slice < kSlices) {
    // fixed summation order (deterministic).  All of a lane's rows are requested in ONE batch of up to 32 independent loads
    // (the rows were written by workgroups on other XCDs, so every load is an L2 miss: round trips, not bytes, set the time)
    const double* base = partials + (size_t)tile_begin * STRIDE + comp;
    double total = 0.0;
    for (int t0 = slice; t0 < tile_count; t0 += 32 * kSlices) {
      double v[32];
#pragma unroll
      for (int k = 0; k < 32; k++) {
        const int t = t0 + k * kSlices;
        v[k] = t < tile_count ? base[(size_t)t * STRIDE] : 0.0;
      }
#pragma unroll
      for (int w = 16; w > 0; w >>= 1) {
#pragma unroll
        for (int k = 0; k < w; k++) v[k] += v[k + w];
      }
      total += v[0];
    }
    lds[slice][comp] = total;
  }
  GP_FIN_TRACE(1);
  __syncthreads();
  if (threadIdx.x < STRIDE) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < kSlices; k++) a += lds[k][threadIdx.x];
    sum[threadIdx.x] = a;
  }
  __syncthreads();
  GP_FIN_TRACE(2);
  (void)NACC;
  const int t = threadIdx.x;
  const int r = t / 6, c = t % 6;  // t < 36: one 6x6 entry per lane
  // the record is assembled in LDS and leaves in one coalesced sweep of 8-byte stores at the end: when `out` is host-mapped
  // memory, ~130 scattered stores would each be their own PCIe write
  __shared__ double dst[122];
  double* out_rec = reinterpret_cast<double*>(out + fi);  // may be only 8-byte aligned (integrated_vgicp_factor_gpu.cpp:219-220)
  constexpr int OFF_HT = 2, OFF_HS = 38, OFF_HTS = 74, OFF_BT = 110, OFF_BS = 116;
  const Pose T = inl.use ? load_pose(inl.lin) : load_pose(poses + 16 * (size_t)fi);
  if (t < 36) {
    // H_t = [[TL, -K^T], [-K, M]]
    const int sym3[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    double h;
    if (r < 3 && c < 3) {
      h = sum[ACC_TL + sym3[r][c]];
    } else if (r >= 3 && c < 3) {
      h = -sum[ACC_K + (r - 3) * 3 + c];
    } else if (r < 3) {
      h = -sum[ACC_K + (c - 3) * 3 + r];
    } else {
      h = sum[ACC_M + sym3[r - 3][c - 3]];
    }
    Ht[r][c] = h;
    dst[OFF_HT + c * 6 + r] = h;
    // Ad(delta) = [[R, 0], [[t]x R, R]]   ([omega, v] ordering, GTSAM Pose3::AdjointMap)
    const double R[3][3] = {{T.r00, T.r01, T.r02}, {T.r10, T.r11, T.r12}, {T.r20, T.r21, T.r22}};
    const double tx[3][3] = {{0.0, -T.tz, T.ty}, {T.tz, 0.0, -T.tx}, {-T.ty, T.tx, 0.0}};
    double a;
    if (r < 3 && c < 3) {
      a = R[r][c];
    } else if (r < 3) {
      a = 0.0;
    } else if (c >= 3) {
      a = R[r - 3][c - 3];
    } else {
      a = tx[r - 3][0] * R[0][c] + tx[r - 3][1] * R[1][c] + tx[r - 3][2] * R[2][c];
    }
    Ad[r][c] = a;
  }
  if (t < 6) {
    const double b = t < 3 ? sum[ACC_QXMR + t] : sum[ACC_MR + t - 3];
    bt[t] = b;
    dst[OFF_BT + t] = b;
  }
  if (t == 0) {
    dst[0] = sum[ACC_COUNT];
    dst[1] = sum[ACC_ERR];
  }
  __syncthreads();
  if constexpr (!GENERAL) {
    if (t < 36) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) a += Ht[r][k] * Ad[k][c];
      HtA[r][c] = a;
      dst[OFF_HTS + c * 6 + r] = -a;  // H_ts = -H_t Ad
    }
    __syncthreads();
    if (t < 36) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) a += Ad[k][r] * HtA[k][c];
      dst[OFF_HS + c * 6 + r] = a;  // H_s = Ad^T H_t Ad
    }
    if (t < 6) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) a += Ad[k][t] * bt[k];
      dst[OFF_BS + t] = -a;  // b_s = -Ad^T b_t
    }
  } else {
    if (t < 36) {
      const int sym3[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
      double h;
      if (r < 3 && c < 3) {
        h = sum[ACCG_HS_TL + sym3[r][c]];
      } else if (r >= 3 && c < 3) {
        h = sum[ACCG_HS_BL + (r - 3) * 3 + c];
      } else if (r < 3) {
        h = sum[ACCG_HS_BL + (c - 3) * 3 + r];
      } else {
        h = sum[ACCG_HS_BR + sym3[r - 3][c - 3]];
      }
      dst[OFF_HS + c * 6 + r] = h;
      dst[OFF_HTS + c * 6 + r] = sum[ACCG_HTS + r * 6 + c];
    }
    if (t < 6) dst[OFF_BS + t] = sum[ACCG_BS + t];
  }
  __syncthreads();
  GP_FIN_TRACE(3);
  if (t < 122) out_rec[t] = dst[t];
  signal_done(done, fi, t < 122);
}

// finalize of the 29-sum (rigid pose) pass, round 2.  The timeline of the generic kernel above (scripts/trace_finalize.py) showed
// 2.7 us for the partials (250 KB through ONE compute unit's L1: bandwidth, not latency), and then 2.1 + 2.4 us for what should be
// nothing: sixteen waves meeting at four barriers, 6x6 tables indexed at run time (= scratch memory round trips) and a pose fetched
// when it was needed.  Here all sixteen waves load and tree-add their rows, the two slices of a wave meet through one cross-lane
// add, ONE barrier, and wave 0 alone does the rest out of LDS with wave-level synchronisation; the pose is requested first of all.
// THREADS: 1024 (one workgroup sums all the tiles of a factor) or 256 (the split form: an eighth of a large factor's tiles per workgroup --
// four waves meet at the barrier instead of sixteen, each lane requests 16 rows instead of 4, all of them in flight together)
template <int THREADS>
__global__ void __launch_bounds__(THREADS) vgicp_finalize_rigid_kernel(const FactorDesc* __restrict__ factors, const double* __restrict__ poses,
                                                                       const InlinePoses inl, const double* __restrict__ partials,
                                                                       gp_linearized6* __restrict__ out, const DoneFlags done, const int parts) {
  static_assert(ACC_STRIDE == 32 && (THREADS == 1024 || THREADS == 256), "lane = (slice parity, component); two slices per wave");
  constexpr int kSlices = THREADS / 32, kWaves = THREADS / 64;
  // parts > 1 (synchronous single-factor calls with many tiles): every record entry is LINEAR in the 29 sums, so `parts` workgroups
  // each expand their share of the tiles into a complete record of their own (slot fi * parts + part, own completion word) and the
  // host adds the records in slot order -- the 250 KB of partials of a 1 M-point factor then go through `parts` compute units'
  // L1s instead of one (that was 4.9 of the kernel's 7.1 us)
  // parts < 0: |parts| parts, and a part delivers its 32 SUMS instead of a record: the host adds the parts' sums and expands them
  // (expand_rigid_host) -- the expansion is the same for every part, so it runs once, off the device's critical path
  const bool sums_only = parts < 0;
  const int nparts = sums_only ? -parts : parts;
  const int fi = blockIdx.x / nparts, part = blockIdx.x - fi * nparts;
  const Pose T = inl.use ? load_pose(inl.lin) : load_pose(poses + 16 * (size_t)fi);  // in flight while the partials arrive
  int tile_begin = inl.use ? inl.factor.tile_begin : factors[fi].tile_begin;
  int tile_count = inl.use ? inl.factor.tile_count : factors[fi].tile_count;
  if (nparts > 1) {
    const int per = (tile_count + nparts - 1) / nparts;
    const int lo = min(part * per, tile_count), hi = min(lo + per, tile_count);
    tile_begin += lo;
    tile_count = hi - lo;
  }
  __shared__ double wsum[kWaves][32];
  __shared__ RigidScratch S;
  GP_FIN_TRACE(0);
  if (done.trace && threadIdx.x == 0 && blockIdx.x == 0) done.trace[9] = __builtin_amdgcn_s_getreg(GP_GETREG_XCC_ID);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int comp = threadIdx.x & 31, slice = threadIdx.x >> 5;
  {
    // fixed summation order (deterministic).  All of a lane's rows are requested in ONE batch of up to 32 independent loads
    double total = rigid_slice_total<kSlices, false>(partials + (size_t)tile_begin * ACC_STRIDE + comp, tile_count, slice);
    total += __shfl_xor(total, 32, 64);  // the wave's two slices
    if (lane < 32) wsum[wave][lane] = total;
  }
  GP_FIN_TRACE(1);
  __syncthreads();
  if (wave != 0) return;  // (a finished wave no longer takes part in barriers; none follow anyway)
  if (lane < 32) S.sum[lane] = rigid_wave_tree<kWaves>(&wsum[0][0], lane);
  GP_WAVE_SYNC();
  GP_FIN_TRACE(2);
  if (sums_only) {
    double* out_rec = reinterpret_cast<double*>(out + blockIdx.x);
    if (lane < 32) out_rec[lane] = S.sum[lane];
    GP_FIN_TRACE(3);
    if (done.flags) {
      GP_FIN_TRACE(4);
      __threadfence_system();
      GP_FIN_TRACE(5);
      if (lane == 0) __hip_atomic_store(done.flags + blockIdx.x, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      GP_FIN_TRACE(6);
    }
    return;
  }
  rigid_expand_wave(S, T, lane);
  const int t = lane;
  const double* dst = S.dst;
  GP_FIN_TRACE(3);
  // the record leaves in two coalesced sweeps of 8-byte stores (it may be only 8-byte aligned, integrated_vgicp_factor_gpu.cpp:219-220;
  // when `out` is host-mapped memory, scattered stores would each be their own PCIe write)
  double* out_rec = reinterpret_cast<double*>(out + blockIdx.x);
  out_rec[t] = dst[t];
  if (t + 64 < 122) out_rec[t + 64] = dst[t + 64];
  if (done.flags) {
    GP_FIN_TRACE(4);
    __threadfence_system();  // the record is visible to the host before the word that announces it (one wave: no barrier needed)
    GP_FIN_TRACE(5);
    if (t == 0) __hip_atomic_store(done.flags + blockIdx.x, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    GP_FIN_TRACE(6);
  }
}

__global__ void __launch_bounds__(kBlockThreads) vgicp_finalize_error_kernel(const FactorDesc* __restrict__ factors, const double* __restrict__ partials,
                                                                             double* __restrict__ out, int single_tile_count, const DoneFlags done) {
  const int fi = blockIdx.x;
  const int tile_begin = single_tile_count >= 0 ? 0 : factors[fi].tile_begin, tile_count = single_tile_count >= 0 ? single_tile_count : factors[fi].tile_count;
  static_assert(kBlockThreads == 256, "error_factor_total is written for 256 threads");
  __shared__ double lds[kBlockThreads / 64];
  const double a = error_factor_total<false>(partials, tile_begin, tile_count, lds);  // (shared with the tile kernel's fused tail: same order, same bits)
  if (threadIdx.x == 0) out[fi] = a;
  signal_done(done, fi, threadIdx.x == 0);
}

// split form of the error finalize for a synchronous single-factor evaluation with >= kFinalizeSplitTiles rows: part g sums rows [g * per, ...) and hands its
// sum to the host, which adds the parts in slot order.  Same code as the fused form runs in the part's last tile workgroup (finalize_part_error).
__global__ void __launch_bounds__(256) vgicp_finalize_error_parts_kernel(const double* __restrict__ partials, const int num_rows, const int rows_per_part, double* __restrict__ out,
                                                                         const int out_stride, const DoneFlags done) {
  __shared__ double wsum[4];
  const int part = blockIdx.x, row_begin = part * rows_per_part;
  finalize_part_error(partials, row_begin, min(rows_per_part, num_rows - row_begin), wsum, out + (size_t)part * out_stride, done.flags + part, done.seq);
}

int wait_done(const unsigned long long* flags_host, size_t count, unsigned long long seq, hipStream_t stream, long spin_us) {
  const volatile unsigned long long* fl = flags_host;
  const auto t0 = std::chrono::steady_clock::now();
  size_t next = 0;
  for (unsigned spins = 1;; spins++) {
    while (next < count && fl[next] == seq) next++;
    if (next == count) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return GP_OK;
    }
    __builtin_ia32_pause();
    // failed kernels, whose words never arrive, are left to the runtime once the spin budget of the call is used up (the caller sizes
    // it on the work: a 100 us budget sent every pass of the 256- and 512-factor batches into hipStreamSynchronize, whose wake-up
    // costs tens of microseconds)
    if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
  }
  GP_HIP(hipStreamSynchronize(stream));
  return GP_OK;
}

int launch_finalize_single(hipStream_t stream, const double* pose_dev, const double* pose_host, const double* partials, int num_tiles, gp_linearized6* out_dev,
                           bool general, DoneFlags done) {
  InlinePoses inl{};
  memcpy(inl.lin, pose_host, sizeof(double) * 16);
  inl.factor.tile_begin = 0;
  inl.factor.tile_count = num_tiles;
  inl.use = 1;
  if (general)
    hipLaunchKernelGGL(vgicp_finalize_kernel<true>, dim3(1), dim3(kFinalizeThreads), 0, stream, (const FactorDesc*)nullptr, pose_dev, inl, partials, out_dev, done);
  else
    hipLaunchKernelGGL(vgicp_finalize_rigid_kernel<kFinalizeThreads>, dim3(1), dim3(kFinalizeThreads), 0, stream, (const FactorDesc*)nullptr, pose_dev, inl, partials, out_dev, done, 1);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GP_OK : hip_fail(e, "vgicp_finalize_kernel", __FILE__, __LINE__);
}

int launch_finalize_error_single(hipStream_t stream, const double* partials, int num_tiles, double* out_dev, DoneFlags done) {
  hipLaunchKernelGGL(vgicp_finalize_error_kernel, dim3(1), dim3(kBlockThreads), 0, stream, (const FactorDesc*)nullptr, partials, out_dev, num_tiles, done);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GP_OK : hip_fail(e, "vgicp_finalize_error_kernel", __FILE__, __LINE__);
}

}  // namespace gp

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

struct gp_vgicp_batch;

// Per-batch tuning (gp_vgicp_batch_set_tuning / gp_vgicp_factor_set_tuning; keys GP_TUNE_* of gtsam_points_hip.h).  Nothing here is process-global:
// two batches on two threads may run different kernels side by side (re-entrant across handles).
constexpr int kDefaultSkewPermille = -1;  // automatic: by the mean share (make_stream_plan)
struct gp_vgicp_tuning {
  int kernel = GP_KERNEL_STREAM;  // GP_KERNEL_*: which tile kernel family the batch prefers (it falls back where that family does not apply)
  int source_policy = 0;          // cache policy of the source stream: 0 = per batch (non-temporal iff no two factors share a source cloud), 1 = default, 2 = non-temporal
  int xcd_chunk = 0;              // workgroup -> tile map: 0 = every XCD walks a contiguous eighth of the tile list, c > 0 = runs of c tiles dealt round robin
  int stagger = 0;                // round-2 kernels: odd wave slots start `stagger` x 512 clocks late
  int tile_interleave = 0;        // consecutive factors that share a source cloud take turns tile by tile
  int xcd_weights[8] = {1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000};  // GP_TUNE_XCD_WEIGHT_0 + x: share of XCD x in 1/1000 of the mean (unset: the library's table)
  bool xcd_weights_set = false;
  int max_wgs = 1024;             // stream family, one large factor: workgroups of the planned launch (<= one resident round of 1024)
  int tile_chunks = 0;            // stream family, batches: 64-point chunks per wave of a tile (tile = 256 x this many points); 0 = the largest of 4 / 2 / 1 that fills 3/4 of the chip
  int fused_finalize = 1;       // synchronous single-factor linearise of the stream family: the last tile workgroup of each part sums the part's rows
                                  // and hands them to the host (InlinePoses: fused finalize) -- no finalize launch.  (The first form of this knob, finalize
                                  // workgroups on a second stream waiting for the counters, cost +10 us per step: profiles/r03_overlap_finalize.jsonl)
  int balance = kDefaultSkewPermille;  // stream kernel, one large factor: how much more a dispatch round takes than the next, in 1/1000 of the mean share (0 = flat)
  int experiment = 0;             // GP_TUNE_EXPERIMENT: measurement instantiations of the stream kernel (gp_vgicp_stream.hpp, EXP), 0 = the product kernel
  int source_mirror = 1;          // stream family: stream the sources' packed mirrors (36 B per point, gp::SourceMirror) when every factor of the batch has one; 0 = the caller's arrays
};

struct gp_vgicp_factor {
  const gp_voxelmap* target = nullptr;
  const float* points = nullptr;
  const float* covs = nullptr;
  const float* normals = nullptr;
  int n = 0;
  int device = 0;  // the device the source arrays live on (hipPointerGetAttributes at creation)
  bool surface_validation = false;
  uint64_t generation = 0;  // bumped when the source pointers or flags change (tables that hold this factor go stale)
  double inlier_thresh_trans = 1e-6, inlier_thresh_angle = 1e-6;  // integrated_vgicp_derivatives.cu:26-27 (kept for API parity)
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  gp_temp_buffer* temp_buffer = nullptr;
  bool owns_temp_buffer = false;
  std::shared_ptr<gp::SourceMirror> mirror;  // packed mirror of (points, covs, n), shared with the other factors on this cloud; acquired by the first table build that wants it
  bool mirror_tried = false;                 // ... which happened (a cloud of < 64 points has none)
  gp_vgicp_batch* self_batch = nullptr;  // lazily built batch of one, used by the per-factor entry points
  gp_vgicp_tuning tuning;                // what the self batch is created with (gp_vgicp_factor_set_tuning)
};

struct gp_vgicp_batch {
  std::vector<gp_vgicp_factor*> factors;
  hipStream_t stream = nullptr;
  gp_temp_buffer* temp_buffer = nullptr;  // partials arena when provided (per-stream scratch), else own
  int num_tiles = 0;
  gp_vgicp_tuning tuning;
  int family = 0;         // GP_KERNEL_* the tile table was built for (tuning.kernel after the fallbacks)
  bool nt = false;        // source stream non-temporal (tuning.source_policy resolved)
  bool any_sv = false;    // some factor validates surfaces
  bool packed = false;    // every factor streams its packed mirror (vgicp_stream_kernel<PK>)
  bool planned = false;   // stream kernel, one large factor: the tile list is a balanced StreamPlan (else fixed tiles of tile_points)
  gp::StreamPlan plan{};  // ... how the chunks are dealt (also written into the tile table)
  unsigned long long* trace = nullptr;  // timeline build of the tile kernel: [2048][16] uint64 device buffer (gp_vgicp_batch_set_trace_buffer)
  // fused finalize (GP_TUNE_FUSED_FINALIZE; synchronous single-factor calls of the stream family)
  gp::DeviceArray d_arrive;               // 16 monotonic arrival counters, kArriveStride words apart
  gp::DeviceArray d_factor_arrive;        // fused finalize by factor: one counter per factor, kFactorArriveStride words apart, zero between launches
  size_t factor_arrive_count = 0;
  bool factor_arrive_dirty = false;       // a by-factor fused launch went out and has not been seen to complete: the zero-reset counters may be anywhere
  unsigned long long arrived[16] = {0};   // what the counters read once every launch issued so far has finished
  bool timing = false;                  // GP_TUNE_TIMING: the synchronous linearise brackets its two kernels with HIP events (gp_vgicp_batch_last_kernel_ms)
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  float last_tile_ms = 0.f, last_finalize_ms = 0.f;
  // fused steps: the kernel's own 100 MHz time stamps (gp_vgicp_batch_device_times)
  double dev_steps = 0.0, dev_stream_ticks = 0.0, dev_kernel_ticks = 0.0;
  int tile_points = 0;
  int64_t total_points = 0;
  gp::DeviceArray d_factors, d_tiles, d_partials, d_poses;  // d_poses: [2][F][16] (lin, eval)
  bool use_grid = false;  // every factor's map carries an occupancy-block grid (else the hashed line table is used)
  bool stream_once = false;  // no two factors of the batch read the same source cloud (and no sibling batch on this device does: gp_multi.hip
                             // clears shares_device_ok): the source stream may use the non-temporal policy
  bool shares_device_ok = true;
  gp_linearized6 view_store{};  // gp_vgicp_batch_linearize_view of a single large factor: the record the host combined from the finalize parts
  int ppt = 4;            // 64-point chunks per wave of the pipeline kernel (tile = 256 x ppt points)
  std::vector<gp::FactorDesc> h_descs;  // host copy of the factor table (a single factor rides in the kernel arguments)
  gp::PinnedArray h_poses;
  hipEvent_t h2d_done = nullptr;  // recorded behind every H2D copy of h_poses; the next staging waits on it
  gp::PinnedArray h_out;  // results land here straight from the finalize kernel (host-mapped, no D2H copy op)
  void* h_out_dev = nullptr;
  gp::PinnedArray h_done;  // one completion word per factor, written by the finalize kernel behind its record (synchronous calls poll it)
  void* h_done_dev = nullptr;
  bool dev_error_fused = false;  // gp_vgicp_batch_issue_compute_error_dev_begin went out in the by-factor fused form
  unsigned long long seq = 0;
  bool table_dirty = true;
  std::vector<uint64_t> seen;  // per factor: its generation + its target map's generation when the table was built
};

namespace {

// Tile-kernel families (GP_KERNEL_*, per batch: gp_vgicp_batch_set_tuning(GP_TUNE_KERNEL)):
//   GP_KERNEL_REFERENCE (0)   reference-shaped kernel (reference bucket table, one point per lane per stride, all modes): the in-library cross-check and
//                             the 92-sum path of non-orthonormal poses
//   GP_KERNEL_HASHED (2)      round-1/2 pipeline kernel over the hashed line table, f32 outer products: what a map without a block grid runs
//   GP_KERNEL_GRID_F64 (3)    round-2 pipeline kernel over the block grid, f64 throughout (parity 4e-10 / 1e-15 instead of 6e-9)
//   GP_KERNEL_LOOKAHEAD (8)   round-2 pipeline kernel, block grid, f32 outer products, look-ahead lookup: what maps with >= 2^26 voxels run
//   GP_KERNEL_STREAM (12)     third generation (gp_vgicp_stream.hpp): per-wave chunk streams, balanced single-factor launches, surface validation
//                             in the ring.  Default.
// A family that does not apply to a batch (no block grid, records beyond 32-bit offsets) falls back to LOOKAHEAD, and
// that to HASHED.  Measured and removed (numbers: DESIGN.md section 8 of rounds 1-3): the f64 hashed kernel, the non-lean start, forced 512- / 256-point
// tiles, the deep pipeline, the source-frame formulation.
// (the A/B switches of rounds 2-3 -- GP_POSES_ZERO_COPY, GP_FINALIZE_PARTS, GP_FINALIZE_NARROW, GP_FINALIZE_HOST_EXPAND: environment variables read once per
// process -- are gone: their measurements are in profiles/r02_*, r03_* and DESIGN.md; the winners are constants.  The library reads no environment variable that
// changes what it computes or launches.)
constexpr bool g_zero_copy_poses = true;  // synchronous batched calls: the kernels read the poses where the host staged them (stage_poses)
constexpr int kPipelineChunks = 4;  // 64-point chunks per wave: 1024-point tiles
constexpr int kFinalizePartsMax = 16;
// workgroups sharing the finalize of a synchronous single-factor call (A/B on C2, scripts/r02/r02_finalize_parts.sh: 1: 7.1 us, 2: 5.5, 4: 4.8, 8: 4.6, 16: 4.7 and the host step suffers;
// round 4, fused form, 8 against 16: C2 fused kernel 12.2-12.8 vs 12.1-12.9 us, step 19.9-20.2 vs 20.0-20.6; 125 k points: kernel 6.4-6.5 vs 6.2, step 13.7-14.4 vs 14.2-14.7)
static constexpr int finalize_parts() { return 8; }
constexpr int kFinalizeSplitTiles = 256;  // ... when the factor has at least this many tiles
constexpr int kResidentWorkgroups = 1024;  // 256 compute units x 4 workgroups of the tile kernels (34-40 KB of LDS, <= 128 VGPRs)

// where a launch takes its poses from
struct PoseSource {
  const double* d_lin = nullptr;
  const double* d_eval = nullptr;
  gp::InlinePoses inl{};
};

// Stream kernel, ONE large factor of n points: the launch geometry (number of workgroups, a multiple of 8) and how the chunks are dealt
// (StreamPlan, gp_vgicp_shared.hpp).  At most one resident round of workgroups whatever n is.
//   skew_permille   how much more a dispatch round takes than the next one, in 1/1000 of the mean share (0 = flat split)
//   xcd_weights     per-XCD share in 1/1000 of the mean share (1000 = equal), or null = kXcdWeightPermille
// Measured on MI355X (round 4, scripts/r04_instep_xcd.py: per-workgroup start / end stamps of the 1 M-point headline INSIDE synchronous steps, i.e. behind
// an idle queue -- the pattern every synchronous call runs in): the command processor hands the dispatch to the XCDs one after the other, in the order
// 0, 1, 2, 3, 7, 6, 5, 4: XCD 1 starts 0.17 us behind XCD 0, XCD 3 0.5 us, XCD 7 0.6-0.9, XCD 4 1.1-1.5 us (two boxes) -- and with equal shares they END
// as much later: XCDs 0-3 at 10.4-11.3 us, XCD 4 at 12.8.  (Back to back the offsets are 0.3-0.7 us, which is why round 3's weights, measured back
// to back, moved nothing.)  A workgroup's life is ~3.7 us of fill and drain + ~6 us that scale with its share, so the shares that equalise the ends are
// 1 + (mean offset - offset) / 6 us: the table below (mean of the two boxes' offsets).  Last end 12.8 -> 11.8-11.9 us in the traced build.
// GP_TUNE_XCD_WEIGHT_0 + x overrides.
constexpr int kXcdWeightPermille[gp::kNumXCD] = {1090, 1070, 1045, 1025, 905, 935, 950, 980};
// waves = waves per workgroup: 4 in the product; the 8- / 16-wave geometries of round 6 (gp_vgicp_stream.hpp, W) were planned through the same function
int make_stream_plan(int n, int skew_permille, const int* xcd_weights, gp::StreamPlan* p, int max_wgs = kResidentWorkgroups, int waves = 4) {
  const int C = n / gp::kChunkPoints;
  max_wgs = std::min(max_wgs, kResidentWorkgroups * 4 / waves);  // one resident round: 16 waves per compute unit
  const int G = std::min(std::max(gp::kNumXCD, max_wgs / gp::kNumXCD * gp::kNumXCD), (std::max((C + waves - 1) / waves, 1) + gp::kNumXCD - 1) / gp::kNumXCD * gp::kNumXCD);
  const int gx = G / gp::kNumXCD;
  *p = gp::StreamPlan{};
  p->tail = n % gp::kChunkPoints;
  p->wgs_per_xcd = gx;
  const double mean_chunks = (double)C / G;
  if (skew_permille < 0) {
    // automatic.  Round 3 needed ~250 at 15 chunks per workgroup because the rounds also had to absorb the XCDs' start offsets; with those in the XCD shares
    // (below) the in-step sweeps of round 4 (scripts/r04_sweep.py, 1 M / 3 M / 8 M points) put the best value at 100-150 for every size: 250 costs 0.3-0.5 us at
    // 1 M and 3 us at 8 M, 50 as much
    skew_permille = 150;
  }
  // shares of the XCDs: proportional to their weights, whole chunks, summing to C (largest remainders first; equal weights = cx or cx + 1).
  // The library's table compensates a FIXED delay (the XCD's dispatch offset), measured at the headline's 15.26 chunks per workgroup: its deviations from 1000
  // scale with 15.26 / (chunks per workgroup) -- an 8 M-point source gets an eighth of them (the unscaled table cost it 1.5 us of 60), a 100 k-point one twice.
  int scaled[gp::kNumXCD];
  if (!xcd_weights) {
    const double k = std::min(2.0, 15.26 / std::max(mean_chunks, 1.0));
    for (int x = 0; x < gp::kNumXCD; x++) scaled[x] = 1000 + (int)std::lround((kXcdWeightPermille[x] - 1000) * k);
  }
  const int* w = xcd_weights ? xcd_weights : scaled;
  int share[gp::kNumXCD];
  {
    int64_t wsum = 0;
    for (int x = 0; x < gp::kNumXCD; x++) wsum += std::max(w[x], 1);
    int given = 0;
    int64_t frac[gp::kNumXCD];
    for (int x = 0; x < gp::kNumXCD; x++) {
      const int64_t num = (int64_t)C * std::max(w[x], 1);
      share[x] = (int)(num / wsum);
      frac[x] = num % wsum;
      given += share[x];
    }
    for (int left = C - given; left > 0; left--) {
      int best = 0;
      for (int x = 1; x < gp::kNumXCD; x++)
        if (frac[x] > frac[best]) best = x;
      share[best]++;
      frac[best] = -1;
    }
  }
  for (int x = 0, at = 0; x < gp::kNumXCD; x++) {
    p->xbegin[x] = at;
    at += share[x];
  }
  const int rounds = (gx + gp::kStreamRound - 1) / gp::kStreamRound;  // <= 4
  const int late = gx - gp::kStreamRound * (rounds - 1);
  auto fill = [&](int x, double skew) {  // shares of the rounds of XCD x in front of its last one; returns what the last round's workgroups share
    const double mean = (double)share[x] / gx;
    int used = 0;
    for (int r = 0; r < 3; r++) p->n[x][r] = p->pre[x][r] = 0;
    for (int r = 0; r + 1 < rounds; r++) {
      p->n[x][r] = std::max(0, (int)std::ceil(mean * (1.0 + skew * (0.5 * (rounds - 1) - r)) - 1e-9));  // (rounded up: the last round never ends up above the one before it)
      p->pre[x][r] = used;
      used += gp::kStreamRound * p->n[x][r];
    }
    p->before_last[x] = used;
    return share[x] - used;
  };
  // the last round must get something sensible on every XCD: at least a third of the mean share per workgroup, never a negative rest; else
  // (and for skew 0) ONE flat round per XCD: every workgroup floor(share / gx) chunks, the first share % gx one more
  bool skewed = skew_permille > 0 && rounds > 1;
  for (int x = 0; x < gp::kNumXCD && skewed; x++) {
    const int rem = fill(x, skew_permille / 1000.0);
    if (rem < 0 || (int64_t)rem * 3 * gx < (int64_t)share[x] * late) skewed = false;
  }
  p->last_begin = skewed ? gp::kStreamRound * (rounds - 1) : 0;
  if (!skewed)
    for (int x = 0; x < gp::kNumXCD; x++) {
      for (int r = 0; r < 3; r++) p->n[x][r] = p->pre[x][r] = 0;
      p->before_last[x] = 0;
    }
  const int late_wgs = gx - p->last_begin;
  for (int x = 0; x < gp::kNumXCD; x++) {
    const int left = share[x] - p->before_last[x];
    p->lo[x] = left / late_wgs;
    p->extra[x] = left % late_wgs;
  }
  return G;
}
constexpr int kPlanMinPoints = 65536;  // single factors below this keep the fixed tiles of a batch (their records then do not depend on how they are batched)

int build_table(gp_vgicp_batch* b) {
  const int F = (int)b->factors.size();
  std::vector<gp::FactorDesc> descs((size_t)F);
  std::vector<gp::TileDesc> tiles;
  b->total_points = 0;
  b->use_grid = true;
  bool offsets32 = true;
  b->any_sv = false;
  {
    // the hashed family reads the maps' private line tables, which a map with a block grid builds only now (gp_voxelmap::ensure_private_table)
    bool all_grid = true;
    for (int i = 0; i < F; i++) {
      if (!b->factors[i]->target->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "VGICP factor: target voxel map is not loaded on the GPU");
      if (!b->factors[i]->target->has_grid) all_grid = false;
    }
    if (!all_grid || b->tuning.kernel == GP_KERNEL_HASHED)
      for (int i = 0; i < F; i++) GP_TRY(const_cast<gp_voxelmap*>(b->factors[i]->target)->ensure_private_table());  // (a cache of the map, not its state)
  }
  for (int i = 0; i < F; i++) {
    const gp_vgicp_factor* f = b->factors[i];
    if (!f->target->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "VGICP factor: target voxel map is not loaded on the GPU");
    gp::FactorDesc& d = descs[i];
    d.points = f->points;
    d.covs = f->covs;
    d.normals = f->normals;
    d.map = f->target->view();
    d.n = f->n;
    d.surface_validation = (f->surface_validation && f->normals) ? 1 : 0;
    if (!d.map.gblocks) b->use_grid = false;
    if ((int64_t)d.map.num_voxels * 64 >= (int64_t)1 << 32) offsets32 = false;
    if (d.surface_validation) b->any_sv = true;
    b->total_points += f->n;
  }
  // the family this batch really runs
  int fam = b->tuning.kernel;
  if (fam == GP_KERNEL_STREAM && (!b->use_grid || !offsets32)) fam = GP_KERNEL_LOOKAHEAD;
  if ((fam == GP_KERNEL_LOOKAHEAD || fam == GP_KERNEL_GRID_F64) && !b->use_grid) fam = GP_KERNEL_HASHED;
  b->family = fam;
  // the packed source mirrors (gp::SourceMirror): built once per cloud, here, where a table upload synchronises anyway.  The batch streams them when
  // EVERY factor with a full chunk has a usable one (a cloud with an unsymmetric covariance has none and keeps the whole batch on the caller's arrays)
  b->packed = false;
  if (fam == GP_KERNEL_STREAM && b->tuning.source_mirror) {
    bool all = F > 0;
    for (int i = 0; i < F; i++) {
      gp_vgicp_factor* f = b->factors[i];
      if (!f->mirror_tried) {
        GP_TRY(gp::acquire_source_mirror(f->points, f->covs, f->n, f->device, b->stream, &f->mirror));
        f->mirror_tried = true;
      }
      if (f->mirror && f->mirror->usable) descs[i].packed = f->mirror->data.as<char>();
      else if (f->n >= gp::kChunkPoints) all = false;
    }
    b->packed = all;
  }
  if (!b->packed)
    for (auto& d : descs) d.packed = nullptr;
  // tiles
  b->planned = fam == GP_KERNEL_STREAM && F == 1 && descs[0].n >= kPlanMinPoints;
  if (b->planned) {
    // one large factor: the balanced plan; the table holds the same tiles the in-argument launch derives from the plan (plan_tile)
    const int G = make_stream_plan(descs[0].n, b->tuning.balance, b->tuning.xcd_weights_set ? b->tuning.xcd_weights : nullptr, &b->plan, b->tuning.max_wgs);
    b->ppt = 4;
    b->tile_points = 0;
    descs[0].tile_begin = 0;
    for (int x = 0; x < gp::kNumXCD; x++)
      for (int q = 0; q < b->plan.wgs_per_xcd; q++) {
        int begin = 0, count = 0;
        gp::plan_tile(gp::plan_fields(b->plan, x, q), x, q, &begin, &count);
        tiles.push_back(gp::TileDesc{0, begin, count, (int)tiles.size()});
      }
    descs[0].tile_count = G;
  } else {
    int ppt = kPipelineChunks;  // the hashed-line-table, reference-shaped and f64 kernels exist for 1024-point tiles only
    if (fam == GP_KERNEL_STREAM) {  // (round 6: the LOOKAHEAD family keeps its 1024-point tiles -- it exists for maps of >= 2^26 voxels, where no batch is small;
                                    //  its 512- / 256-point instantiations went with that: VERDICT r05 #8)
      // per batch: the largest tile that still fills 3/4 of the chip's resident workgroups, so that a 15 k-point scan is not left to 15 workgroups
      ppt = 1;
      for (int cand : {4, 2}) {
        int64_t count = 0;
        for (const auto* f : b->factors) count += (f->n + 256 * cand - 1) / (256 * cand);
        if (count >= kResidentWorkgroups * 3 / 4) {
          ppt = cand;
          break;
        }
      }
    }
    if (fam == GP_KERNEL_STREAM) {
      // the stream kernel takes tiles of any whole number of chunks: 2048-point tiles (eight chunks per wave: the cold start of the pipeline is paid
      // half as often) where that still leaves two rounds of workgroups -- C3 56.7 -> 51.3 us, C4 shard 135 -> 115 us, whole C4 1.08 -> 1.00 ms
      // (profiles/r03_tile_chunks.jsonl; 4096-point tiles: C3 54.7, C4 shard 114)
      int64_t count8 = 0;
      for (const auto* f : b->factors) count8 += (f->n + 2047) / 2048;
      if (count8 >= 2 * kResidentWorkgroups) ppt = 8;
      // ... and 4096-point tiles from 4096 such tiles up (a C4 shard: 512 factors x 8): with the factors finalized inside the tile kernel every tile ends with a
      // write-through row and an arrival, so fewer, larger tiles win while there are still four rounds of them (C4 shard 0.171 -> 0.163 ms; C3, 1536 such tiles: no)
      int64_t count16 = 0;
      for (const auto* f : b->factors) count16 += (f->n + 4095) / 4096;
      if (count16 >= 4 * kResidentWorkgroups) ppt = 16;
      if (b->tuning.tile_chunks > 0) ppt = b->tuning.tile_chunks;
    }
    b->ppt = ppt;
    b->tile_points = 64 * 4 * ppt;
    for (int i = 0; i < F; i++) {
      gp::FactorDesc& d = descs[i];
      d.tile_begin = (int)tiles.size();
      for (int p = 0; p < d.n; p += b->tile_points) tiles.push_back(gp::TileDesc{i, p, std::min(b->tile_points, d.n - p), (int)tiles.size()});
      d.tile_count = (int)tiles.size() - d.tile_begin;
    }
  }
  b->num_tiles = (int)tiles.size();
  {
    std::vector<const float*> srcs;
    for (const auto& d : descs) srcs.push_back(d.points);
    std::sort(srcs.begin(), srcs.end());
    b->stream_once = b->shares_device_ok && std::adjacent_find(srcs.begin(), srcs.end()) == srcs.end();
    b->nt = b->tuning.source_policy == 2 || (b->tuning.source_policy == 0 && b->stream_once);
  }
  if (b->tuning.tile_interleave && !b->planned) {
    // execution order: consecutive factors that read the SAME source cloud (a submap matched against several targets, BASELINE
    // configs[3]) take turns tile by tile, so the workgroups that run side by side on an XCD read the same source bytes at the same
    // time -- one of them misses L2, the others hit.  Partial rows stay factor-major (TileDesc::row).
    std::vector<gp::TileDesc> order;
    order.reserve(tiles.size());
    for (int i = 0; i < F;) {
      int j = i + 1;
      while (j < F && descs[j].points == descs[i].points && descs[j].covs == descs[i].covs && descs[j].n == descs[i].n) j++;
      for (int t = 0; t < descs[i].tile_count; t++)
        for (int k = i; k < j; k++) order.push_back(tiles[(size_t)descs[k].tile_begin + t]);
      i = j;
    }
    tiles.swap(order);
  }
  b->h_descs = descs;
  GP_TRY(b->d_factors.ensure(sizeof(gp::FactorDesc) * (size_t)std::max(F, 1)));
  GP_TRY(b->d_tiles.ensure(sizeof(gp::TileDesc) * (size_t)std::max(b->num_tiles, 1)));
  GP_TRY(b->d_poses.ensure(sizeof(double) * 32 * (size_t)std::max(F, 1)));
  GP_TRY(b->h_poses.ensure(sizeof(double) * 32 * (size_t)std::max(F, 1)));
  GP_TRY(b->h_out.ensure(sizeof(gp_linearized6) * (size_t)std::max(F, kFinalizePartsMax)));
  GP_HIP(hipHostGetDevicePointer(&b->h_out_dev, b->h_out.ptr, 0));
  {
    const size_t before = b->h_done.bytes;
    GP_TRY(b->h_done.ensure(sizeof(unsigned long long) * (size_t)std::max(F, kFinalizePartsMax)));
    if (b->h_done.bytes != before) memset(b->h_done.ptr, 0, b->h_done.bytes);
    GP_HIP(hipHostGetDevicePointer(&b->h_done_dev, b->h_done.ptr, 0));
  }
  if (!b->temp_buffer) GP_TRY(b->d_partials.ensure(sizeof(double) * gp::ACCG_STRIDE * (size_t)std::max(b->num_tiles, 1)));
  // the table upload is synchronous (pageable source); it happens once per factor-set change, not per linearise
  if (F) GP_HIP(hipMemcpy(b->d_factors.ptr, descs.data(), sizeof(gp::FactorDesc) * (size_t)F, hipMemcpyHostToDevice));
  if (b->num_tiles) GP_HIP(hipMemcpy(b->d_tiles.ptr, tiles.data(), sizeof(gp::TileDesc) * (size_t)b->num_tiles, hipMemcpyHostToDevice));
  b->seen.resize((size_t)F);
  for (int i = 0; i < F; i++) b->seen[i] = b->factors[i]->generation + b->factors[i]->target->generation;
  b->table_dirty = false;
  return GP_OK;
}

// the factor table caches device pointers: rebuild it when the tuning changed (set_tuning marks it dirty), a flag or source pointer of a
// factor changed, or a target map was re-inserted / offloaded / reloaded since (OffloadableGPU protocol)
bool table_is_stale(const gp_vgicp_batch* b) {
  if (b->table_dirty || b->seen.size() != b->factors.size()) return true;
  for (size_t i = 0; i < b->factors.size(); i++)
    if (b->seen[i] != b->factors[i]->generation + b->factors[i]->target->generation) return true;
  return false;
}

int partials_ptr(gp_vgicp_batch* b, double** out) {
  if (b->temp_buffer) {
    void* p = nullptr;
    GP_TRY(gp_temp_buffer_get(b->temp_buffer, sizeof(double) * gp::ACCG_STRIDE * (size_t)std::max(b->num_tiles, 1), &p));
    *out = reinterpret_cast<double*>(p);
  } else {
    *out = b->d_partials.as<double>();
  }
  return GP_OK;
}

inline int grid_tiles(int num_tiles, int xcd_chunk = 0) {
  if (xcd_chunk > 0) {
    const int group = gp::kNumXCD * xcd_chunk;
    return (num_tiles + group - 1) / group * group;
  }
  const int per = (num_tiles + gp::kNumXCD - 1) / gp::kNumXCD;
  return per * gp::kNumXCD;
}

// is the 3x3 block of every pose orthonormal to 1e-9?  (gp::pose_is_rigid, gp_vgicp_shared.hpp)
bool poses_are_rigid(const double* poses_host, size_t F) {
  if (!poses_host) return false;
  for (size_t i = 0; i < F; i++)
    if (!gp::pose_is_rigid(poses_host + 16 * i)) return false;
  return true;
}

template <int MODE>
int launch_tiles(gp_vgicp_batch* b, const PoseSource& ps, double* partials) {
  if (b->num_tiles <= 0) return GP_OK;
  const int fam = MODE == gp::MODE_LIN_GENERAL ? GP_KERNEL_REFERENCE : b->family;
  const bool single_plan = b->planned;  // the tile list IS the plan: the contiguous map, whatever xcd_chunk says
  // the reference-shaped kernels (the 92-sum path included) keep the contiguous map
  // ... and so does every in-argument launch of the stream kernel (its fixed-tile branch knows no other map)
  const int chunk = (fam == GP_KERNEL_REFERENCE || single_plan || (fam == GP_KERNEL_STREAM && ps.inl.use)) ? 0 : b->tuning.xcd_chunk;
  const dim3 grid_dim(grid_tiles(b->num_tiles, chunk)), block(gp::kBlockThreads);
  const gp::FactorDesc* fd = b->d_factors.as<gp::FactorDesc>();
  const gp::TileDesc* td = b->d_tiles.as<gp::TileDesc>();
  gp::InlinePoses inl = ps.inl;
  inl.stagger = b->tuning.stagger;
  inl.xcd_chunk = chunk;
  inl.tile_points = b->tile_points;
  inl.plan = b->plan;
  inl.trace = b->trace;
  const bool traced = b->trace != nullptr && inl.use && MODE == gp::MODE_LIN;  // the timeline builds exist for single-factor linearise launches
#define GP_ARGS grid_dim, block, 0, b->stream, fd, td, b->num_tiles, ps.d_lin, ps.d_eval, inl, partials
  if constexpr (MODE == gp::MODE_LIN_GENERAL) {
    hipLaunchKernelGGL(gp::vgicp_tile_kernel<MODE>, GP_ARGS);
  } else if (fam == GP_KERNEL_REFERENCE) {
    hipLaunchKernelGGL(gp::vgicp_tile_kernel<MODE>, GP_ARGS);
  } else if (fam == GP_KERNEL_STREAM) {
#define GP_LAUNCH_STREAM(NT, INL, SV, PK, TRACE) hipLaunchKernelGGL((gp::vgicp_stream_kernel<MODE, NT, INL, SV, PK, TRACE>), GP_ARGS)
#define GP_LAUNCH_STREAM_S(INL, SV)                                  \
  do {                                                               \
    if (b->packed) {                                                 \
      if (b->nt) GP_LAUNCH_STREAM(true, INL, SV, true, false);       \
      else GP_LAUNCH_STREAM(false, INL, SV, true, false);            \
    } else {                                                         \
      if (b->nt) GP_LAUNCH_STREAM(true, INL, SV, false, false);      \
      else GP_LAUNCH_STREAM(false, INL, SV, false, false);           \
    }                                                                \
  } while (0)
    if (traced && !b->any_sv) {
      if constexpr (MODE == gp::MODE_LIN) {
        if (b->packed) {
          if (b->nt) GP_LAUNCH_STREAM(true, true, false, true, true);
          else GP_LAUNCH_STREAM(false, true, false, true, true);
        } else {
          if (b->nt) GP_LAUNCH_STREAM(true, true, false, false, true);
          else GP_LAUNCH_STREAM(false, true, false, false, true);
        }
      }
    } else if (inl.use && MODE == gp::MODE_LIN && b->tuning.experiment && b->planned && b->packed && b->nt && !b->any_sv) {
      // measurement instantiations (GP_TUNE_EXPERIMENT): the headline's shape only -- planned single factor, packed non-temporal stream, no surface validation
      if constexpr (MODE == gp::MODE_LIN) {
        if (b->tuning.experiment == 1) hipLaunchKernelGGL((gp::vgicp_stream_kernel<MODE, true, true, false, true, false, 1>), GP_ARGS);
        else hipLaunchKernelGGL((gp::vgicp_stream_kernel<MODE, true, true, false, true, false, 2>), GP_ARGS);
      }
    } else if (inl.use) {
      if (b->any_sv) GP_LAUNCH_STREAM_S(true, true);
      else GP_LAUNCH_STREAM_S(true, false);
    } else {
      if (b->any_sv) GP_LAUNCH_STREAM_S(false, true);
      else GP_LAUNCH_STREAM_S(false, false);
    }
#undef GP_LAUNCH_STREAM_S
#undef GP_LAUNCH_STREAM
  } else {
    // the round-2 family: <MODE, f32 outer products, chunks per wave, block grid, TRACE, lean start, look-ahead lookup>
#define GP_LAUNCH_PIPE(F32, PPT, GRID, LEAN, AHEAD) hipLaunchKernelGGL((gp::vgicp_pipeline_kernel<MODE, F32, PPT, GRID, false, LEAN, AHEAD>), GP_ARGS)
    if (fam == GP_KERNEL_HASHED) {
      GP_LAUNCH_PIPE(true, 4, false, false, false);
    } else if (fam == GP_KERNEL_GRID_F64) {
      GP_LAUNCH_PIPE(false, 4, true, false, false);
    } else {  // GP_KERNEL_LOOKAHEAD: 1024-point tiles; the linearise with the look-ahead lookup, the error evaluation without
      if constexpr (MODE == gp::MODE_LIN) GP_LAUNCH_PIPE(true, 4, true, true, true);
      else GP_LAUNCH_PIPE(true, 4, true, true, false);
    }
#undef GP_LAUNCH_PIPE
  }
#undef GP_ARGS
  GP_HIP(hipGetLastError());
  return GP_OK;
}

template <bool GENERAL>
int launch_finalize(gp_vgicp_batch* b, const PoseSource& ps, const double* partials, gp_linearized6* out_dev, gp::DoneFlags done = {}, int parts = 1) {
  if constexpr (GENERAL) {
    hipLaunchKernelGGL(gp::vgicp_finalize_kernel<true>, dim3((int)b->factors.size()), dim3(gp::kFinalizeThreads), 0, b->stream, b->d_factors.as<gp::FactorDesc>(),
                       ps.d_lin, ps.inl, partials, out_dev, done);
  } else {
    constexpr bool narrow = true;  // 256-thread parts (1024-thread ones measured slower, round 2)
    const int nparts = parts < 0 ? -parts : parts;  // (parts < 0: the parts deliver their sums, the host expands)
    if (nparts > 1 && narrow)
      hipLaunchKernelGGL(gp::vgicp_finalize_rigid_kernel<256>, dim3((int)b->factors.size() * nparts), dim3(256), 0, b->stream, b->d_factors.as<gp::FactorDesc>(), ps.d_lin,
                         ps.inl, partials, out_dev, done, parts);
    else
      hipLaunchKernelGGL(gp::vgicp_finalize_rigid_kernel<gp::kFinalizeThreads>, dim3((int)b->factors.size() * nparts), dim3(gp::kFinalizeThreads), 0, b->stream,
                         b->d_factors.as<gp::FactorDesc>(), ps.d_lin, ps.inl, partials, out_dev, done, parts);
  }
  GP_HIP(hipGetLastError());
  return GP_OK;
}

// device work of one linearisation pass.  rigid == true: 29-sum kernel + adjoint expansion; false: 92-sum kernel
// (exact for any 3x3 block, like the reference's explicit J_s).
int launch_linearize(gp_vgicp_batch* b, const PoseSource& ps, gp_linearized6* out_dev, bool rigid, gp::DoneFlags done = {}, int parts = 1, bool timed = false) {
  if (b->factors.empty()) return GP_OK;
  double* partials = nullptr;
  GP_TRY(partials_ptr(b, &partials));
  if (timed) GP_HIP(hipEventRecord(b->ev[0], b->stream));
  if (rigid) GP_TRY(launch_tiles<gp::MODE_LIN>(b, ps, partials));
  else GP_TRY(launch_tiles<gp::MODE_LIN_GENERAL>(b, ps, partials));
  if (timed) GP_HIP(hipEventRecord(b->ev[1], b->stream));
  if (rigid) GP_TRY(launch_finalize<false>(b, ps, partials, out_dev, done, parts));
  else GP_TRY(launch_finalize<true>(b, ps, partials, out_dev, done));
  if (timed) GP_HIP(hipEventRecord(b->ev[2], b->stream));
  return GP_OK;
}

int launch_error(gp_vgicp_batch* b, const PoseSource& ps, double* out_dev, gp::DoneFlags done = {}) {
  if (b->factors.empty()) return GP_OK;
  double* partials = nullptr;
  GP_TRY(partials_ptr(b, &partials));
  GP_TRY(launch_tiles<gp::MODE_ERR>(b, ps, partials));
  hipLaunchKernelGGL(gp::vgicp_finalize_error_kernel, dim3((int)b->factors.size()), dim3(gp::kBlockThreads), 0, b->stream, b->d_factors.as<gp::FactorDesc>(),
                     partials, out_dev, -1, done);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

// poses for a launch: a single factor carries them in the kernel arguments; a batch uploads them with one H2D
// how long a synchronous call polls its completion words before it hands over to hipStreamSynchronize: 100 us + 50 ps per source point
// (4x what the tile kernel takes), so that a healthy pass never gets there
long spin_budget_us(const gp_vgicp_batch* b) { return 100 + (long)(b->total_points / 20000); }

// zero_copy (the SYNCHRONOUS entry points, which return only after the kernels have finished): the kernels read the poses straight
// out of the pinned staging buffer over the fabric -- one 128-B scalar read per workgroup, hidden behind the other workgroups -- instead
// of behind a copy-engine transfer (an API call, an event and ~10 us of SDMA start-up in front of the first kernel).  The asynchronous
// issue_* entry points keep the H2D copy: the caller may re-stage poses while the kernels of the previous pass are still running.
int stage_poses(gp_vgicp_batch* b, const double* lin, const double* eval, PoseSource* ps, bool zero_copy = false) {
  const size_t F = b->factors.size();
  if (F == 1) {
    memcpy(ps->inl.lin, lin, sizeof(double) * 16);
    if (eval) memcpy(ps->inl.eval, eval, sizeof(double) * 16);
    ps->inl.factor = b->h_descs[0];
    ps->inl.tile_points = b->tile_points;
    ps->inl.use = 1;
    return GP_OK;
  }
  // the previous pass's H2D copy may still be reading the pinned staging buffer (the issue_* entry points are asynchronous):
  // wait for it before the buffer is overwritten
  if (b->h2d_done) {
    GP_HIP(hipEventSynchronize(b->h2d_done));
  } else {
    GP_HIP(hipEventCreateWithFlags(&b->h2d_done, hipEventDisableTiming));
  }
  double* h = b->h_poses.as<double>();
  memcpy(h, lin, sizeof(double) * 16 * F);
  if (eval) memcpy(h + 16 * F, eval, sizeof(double) * 16 * F);
  if (zero_copy && g_zero_copy_poses) {
    void* hd = nullptr;
    GP_HIP(hipHostGetDevicePointer(&hd, h, 0));
    ps->d_lin = static_cast<const double*>(hd);
    ps->d_eval = eval ? static_cast<const double*>(hd) + 16 * F : nullptr;
    ps->inl.use = 0;
    return GP_OK;
  }
  GP_HIP(hipMemcpyAsync(b->d_poses.ptr, h, sizeof(double) * 16 * F * (eval ? 2 : 1), hipMemcpyHostToDevice, b->stream));
  GP_HIP(hipEventRecord(b->h2d_done, b->stream));
  ps->d_lin = b->d_poses.as<double>();
  ps->d_eval = eval ? b->d_poses.as<double>() + 16 * F : nullptr;
  ps->inl.use = 0;
  return GP_OK;
}

int ensure_self_batch(gp_vgicp_factor* f) {
  if (!f->self_batch) {
    // a batch of one on the factor's own stream whose partials live in the factor's TempBufferManager arena
    // (set before the table is built, so that no second partials array is allocated)
    auto* b = new gp_vgicp_batch;
    b->factors.push_back(f);
    b->stream = f->stream;
    b->temp_buffer = f->temp_buffer;
    b->tuning = f->tuning;
    const int rc = build_table(b);
    if (rc != GP_OK) {
      delete b;
      return rc;
    }
    f->self_batch = b;
  }
  if (table_is_stale(f->self_batch)) GP_TRY(build_table(f->self_batch));
  return GP_OK;
}

}  // namespace

extern "C" {

// ---- per-batch tuning (nothing process-global: SURVEY.md 8(b) "re-entrant across handles") --------------------------------------------------
static int apply_tuning(gp_vgicp_tuning* t, int key, int value) {
  switch (key) {
    case GP_TUNE_KERNEL:
      if (value != GP_KERNEL_REFERENCE && value != GP_KERNEL_HASHED && value != GP_KERNEL_GRID_F64 && value != GP_KERNEL_LOOKAHEAD && value != GP_KERNEL_STREAM)
        return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_KERNEL: one of GP_KERNEL_REFERENCE (0), _HASHED (2), _GRID_F64 (3), _LOOKAHEAD (8), _STREAM (12)");
      t->kernel = value;
      return GP_OK;
    case GP_TUNE_SOURCE_POLICY:
      if (value < 0 || value > 2) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_SOURCE_POLICY: 0 (per batch), 1 (default cache policy), 2 (non-temporal)");
      t->source_policy = value;
      return GP_OK;
    case GP_TUNE_XCD_CHUNK:
      if (value < 0 || value > 4096) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_XCD_CHUNK: 0 (contiguous eighths) .. 4096 tiles per run");
      t->xcd_chunk = value;
      return GP_OK;
    case GP_TUNE_STAGGER:
      if (value < 0 || value > 64) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_STAGGER: 0..64 (x 512 clocks)");
      t->stagger = value;
      return GP_OK;
    case GP_TUNE_TILE_INTERLEAVE:
      t->tile_interleave = value ? 1 : 0;
      return GP_OK;
    case GP_TUNE_FUSED_FINALIZE:
      t->fused_finalize = value ? 1 : 0;
      return GP_OK;
    case GP_TUNE_MAX_WORKGROUPS:
      if (value < 8 || value > 1024) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_MAX_WORKGROUPS: 8 .. 1024");
      t->max_wgs = value;
      return GP_OK;
    case GP_TUNE_TILE_CHUNKS:
      if (value < 0 || value > 64) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_TILE_CHUNKS: 0 (automatic) .. 64 chunks per wave");
      t->tile_chunks = value;
      return GP_OK;
    case GP_TUNE_SOURCE_MIRROR:
      t->source_mirror = value ? 1 : 0;
      return GP_OK;
    case GP_TUNE_EXPERIMENT:
      if (value < 0 || value > 2) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_EXPERIMENT: 0 (off), 1 (block-grid warm-up), 2 (f32 covariance rotation: breaks parity, timing only)");
      t->experiment = value;
      return GP_OK;
    case GP_TUNE_BALANCE:
      if (value < -1 || value > 600) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_BALANCE: -1 (automatic), 0 (flat) .. 600 (per mille of the mean share per dispatch round)");
      t->balance = value;
      return GP_OK;
    default:
      if (key >= GP_TUNE_XCD_WEIGHT_0 && key < GP_TUNE_XCD_WEIGHT_0 + 8) {
        if (value < 500 || value > 1500) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "GP_TUNE_XCD_WEIGHT_x: 500..1500 (per mille of the mean share)");
        t->xcd_weights[key - GP_TUNE_XCD_WEIGHT_0] = value;
        t->xcd_weights_set = true;
        return GP_OK;
      }
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "unknown GP_TUNE_* key");
  }
}

int gp_vgicp_batch_set_tuning(gp_vgicp_batch_t* b, int key, int value) {
  if (!b) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_set_tuning: null batch");
  if (key == GP_TUNE_TIMING) {
    b->timing = value != 0;
    return GP_OK;
  }
  if (key == GP_TUNE_TEST_ARRIVAL_SKEW) {  // test hook: the host's idea of arrival counter 0 runs `value` ahead of the device's, as after a launch that was lost
    b->arrived[0] += (unsigned long long)(value > 0 ? value : 0);
    if (b->d_factor_arrive.ptr && value > 0) {  // ... and factor counter 0 is far out of range: that factor's record cannot complete
      const unsigned long long v = 1ull << 40;
      GP_HIP(hipStreamSynchronize(b->stream));
      GP_HIP(hipMemcpy(b->d_factor_arrive.ptr, &v, sizeof(v), hipMemcpyHostToDevice));
    }
    return GP_OK;
  }
  GP_TRY(apply_tuning(&b->tuning, key, value));
  b->table_dirty = true;
  return GP_OK;
}

int gp_vgicp_batch_get_tuning(const gp_vgicp_batch_t* b, int key, int* value) {
  if (!b || !value) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_get_tuning: null");
  switch (key) {
    case GP_TUNE_KERNEL: *value = b->tuning.kernel; return GP_OK;
    case GP_TUNE_SOURCE_POLICY: *value = b->tuning.source_policy; return GP_OK;
    case GP_TUNE_XCD_CHUNK: *value = b->tuning.xcd_chunk; return GP_OK;
    case GP_TUNE_STAGGER: *value = b->tuning.stagger; return GP_OK;
    case GP_TUNE_TILE_INTERLEAVE: *value = b->tuning.tile_interleave; return GP_OK;
    case GP_TUNE_BALANCE: *value = b->tuning.balance; return GP_OK;
    case GP_TUNE_FUSED_FINALIZE: *value = b->tuning.fused_finalize; return GP_OK;
    case GP_TUNE_TILE_CHUNKS: *value = b->tuning.tile_chunks; return GP_OK;
    case GP_TUNE_EFFECTIVE_KERNEL: *value = b->table_dirty ? -1 : b->family; return GP_OK;  // what the last table build resolved GP_TUNE_KERNEL to
    case GP_TUNE_SOURCE_MIRROR: *value = b->tuning.source_mirror; return GP_OK;
    case GP_TUNE_EXPERIMENT: *value = b->tuning.experiment; return GP_OK;
    case GP_TUNE_EFFECTIVE_MIRROR: *value = b->table_dirty ? -1 : (b->packed ? 1 : 0); return GP_OK;  // does the built table stream the packed mirrors?
    default: return gp::fail(GP_ERROR_INVALID_ARGUMENT, "unknown GP_TUNE_* key");
  }
}

// the per-factor entry points run a batch of one: its tuning is the factor's
int gp_vgicp_factor_set_tuning(gp_vgicp_factor_t* f, int key, int value) {
  if (!f) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_set_tuning: null factor");
  GP_TRY(apply_tuning(&f->tuning, key, value));
  if (f->self_batch) {
    f->self_batch->tuning = f->tuning;
    f->self_batch->table_dirty = true;
  }
  return GP_OK;
}

// timeline hook (measurement): the traced build of the tile kernel of THIS batch's single-factor linearise stores 8 s_memtime stamps + HW_ID /
// XCC_ID + two s_memrealtime stamps per workgroup into dev_buffer ([2048][16] uint64, row = tile index); row 2047 receives the stamps of the
// finalize kernel of synchronous calls.  NULL disables.
int gp_vgicp_batch_set_trace_buffer(gp_vgicp_batch_t* b, void* dev_buffer) {
  if (!b) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_set_trace_buffer: null batch");
  b->trace = static_cast<unsigned long long*>(dev_buffer);
  return GP_OK;
}

size_t gp_vgicp_linearization_input_size(void) { return sizeof(double) * 16; }
size_t gp_vgicp_linearization_output_size(void) { return sizeof(gp_linearized6); }
size_t gp_vgicp_evaluation_input_size(void) { return sizeof(double) * 16; }
size_t gp_vgicp_evaluation_output_size(void) { return sizeof(double); }

int gp_vgicp_factor_create(const gp_voxelmap_t* target, const float* points_dev, const float* covs_dev, const float* normals_dev, int num_points,
                           gp_stream_t stream, gp_temp_buffer_t* temp_buffer, gp_vgicp_factor_t** out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_create: null out");
  // the reference abort()s on these (integrated_vgicp_factor_gpu.cpp:33-46); the C++ mirror keeps that, the C-ABI reports
  if (!points_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: GPU source points have not been allocated!!");
  if (!covs_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: GPU source covs have not been allocated!!");
  if (!target) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: GPU target voxels have not been created!!");
  if (num_points < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_create: negative size");
  auto* f = new gp_vgicp_factor;
  f->target = target;
  f->points = points_dev;
  f->covs = covs_dev;
  f->normals = normals_dev;
  f->n = num_points;
  {
    hipPointerAttribute_t attr{};
    int cur = 0;
    (void)hipGetDevice(&cur);
    f->device = (hipPointerGetAttributes(&attr, points_dev) == hipSuccess) ? attr.device : cur;
    (void)hipGetLastError();  // a pointer the runtime does not know leaves a sticky error behind
  }
  if (stream) {
    f->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);  // integrated_vgicp_derivatives.cu:36-39
    if (e != hipSuccess) {
      delete f;
      return gp::hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
    }
    f->owns_stream = true;
  }
  if (temp_buffer) {
    f->temp_buffer = temp_buffer;
  } else {
    int rc = gp_temp_buffer_create(0, &f->temp_buffer);  // :41-43
    if (rc != GP_OK) {
      if (f->owns_stream) (void)hipStreamDestroy(f->stream);
      delete f;
      return rc;
    }
    f->owns_temp_buffer = true;
  }
  *out = f;
  return GP_OK;
}

int gp_vgicp_factor_destroy(gp_vgicp_factor_t* f) {
  if (!f) return GP_OK;
  if (f->self_batch) gp_vgicp_batch_destroy(f->self_batch);
  if (f->owns_stream) {
    (void)hipStreamSynchronize(f->stream);
    (void)hipStreamDestroy(f->stream);
  }
  if (f->owns_temp_buffer) gp_temp_buffer_destroy(f->temp_buffer);
  delete f;
  return GP_OK;
}

int gp_vgicp_factor_set_surface_validation(gp_vgicp_factor_t* f, int enable) {
  if (!f) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "null factor");
  if (enable && !f->normals) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "surface validation needs source normals (integrated_vgicp_factor_gpu.hpp:84-86)");
  f->surface_validation = enable != 0;
  f->generation++;
  return GP_OK;
}

// the source cloud was offloaded and reloaded (PointCloudGPU::reload_gpu re-allocates): hand the factor the new arrays
int gp_vgicp_factor_set_source(gp_vgicp_factor_t* f, const float* points_dev, const float* covs_dev, const float* normals_dev) {
  if (!f || !points_dev || !covs_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_set_source: null factor / points / covs");
  if (f->surface_validation && !normals_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_set_source: surface validation is on but no normals given");
  f->points = points_dev;
  f->covs = covs_dev;
  f->normals = normals_dev;
  f->mirror.reset();  // (also the way to say "the arrays were rewritten in place": the next table build packs afresh, or joins the mirror of the new arrays)
  f->mirror_tried = false;
  f->generation++;
  return GP_OK;
}

int gp_vgicp_factor_set_inlier_update_thresh(gp_vgicp_factor_t* f, double trans, double angle) {
  if (!f) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "null factor");
  f->inlier_thresh_trans = trans;
  f->inlier_thresh_angle = angle;
  return GP_OK;
}

int gp_vgicp_factor_num_points(const gp_vgicp_factor_t* f) { return f ? f->n : 0; }
int gp_vgicp_factor_device(const gp_vgicp_factor_t* f) { return f ? f->device : 0; }
gp_stream_t gp_vgicp_factor_stream(const gp_vgicp_factor_t* f) { return f ? (gp_stream_t)f->stream : nullptr; }

int gp_vgicp_factor_issue_linearize(gp_vgicp_factor_t* f, const double* pose_host, const double* pose_dev, gp_linearized6* out_dev) {
  if (!f || (!pose_dev && !pose_host) || !out_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_issue_linearize: null");
  GP_TRY(ensure_self_batch(f));
  PoseSource ps;
  if (pose_host) {  // the host copy rides in the kernel arguments; the device copy is not even read
    memcpy(ps.inl.lin, pose_host, sizeof(double) * 16);
    ps.inl.factor = f->self_batch->h_descs[0];
    ps.inl.tile_points = f->self_batch->tile_points;
    ps.inl.use = 1;
  } else {
    ps.d_lin = pose_dev;
  }
  return launch_linearize(f->self_batch, ps, out_dev, poses_are_rigid(pose_host, 1));
}

int gp_vgicp_factor_issue_compute_error(gp_vgicp_factor_t* f, const double* pose_lin_host, const double* pose_eval_host, const double* pose_lin_dev,
                                        const double* pose_eval_dev, double* out_dev) {
  if (!f || !out_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_issue_compute_error: null");
  GP_TRY(ensure_self_batch(f));
  PoseSource ps;
  if (pose_lin_host && pose_eval_host) {
    memcpy(ps.inl.lin, pose_lin_host, sizeof(double) * 16);
    memcpy(ps.inl.eval, pose_eval_host, sizeof(double) * 16);
    ps.inl.factor = f->self_batch->h_descs[0];
    ps.inl.tile_points = f->self_batch->tile_points;
    ps.inl.use = 1;
  } else if (pose_lin_dev && pose_eval_dev) {
    ps.d_lin = pose_lin_dev;
    ps.d_eval = pose_eval_dev;
  } else {
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_issue_compute_error: poses missing");
  }
  return launch_error(f->self_batch, ps, out_dev);
}

int gp_vgicp_factor_sync(gp_vgicp_factor_t* f) {
  if (!f) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "null factor");
  GP_HIP(hipStreamSynchronize(f->stream));
  return GP_OK;
}

int gp_vgicp_factor_linearize(gp_vgicp_factor_t* f, const double pose[16], gp_linearized6* out_host) {
  if (!f || !pose || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_linearize: null");
  GP_TRY(ensure_self_batch(f));
  return gp_vgicp_batch_linearize(f->self_batch, pose, out_host);
}

int gp_vgicp_factor_compute_error(gp_vgicp_factor_t* f, const double pose_lin[16], const double pose_eval[16], double* out_host) {
  if (!f || !pose_lin || !pose_eval || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_factor_compute_error: null");
  GP_TRY(ensure_self_batch(f));
  return gp_vgicp_batch_compute_error(f->self_batch, pose_lin, pose_eval, out_host);
}

// ---- batch ----------------------------------------------------------------------------------------------------

int gp_vgicp_batch_create(gp_vgicp_factor_t* const* factors, int num_factors, gp_stream_t stream, gp_vgicp_batch_t** out) {
  if (!out || num_factors < 0 || (num_factors > 0 && !factors)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_create: bad arguments");
  auto* b = new gp_vgicp_batch;
  for (int i = 0; i < num_factors; i++) {
    if (!factors[i]) {
      delete b;
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_create: null factor");
    }
    b->factors.push_back(factors[i]);
  }
  b->stream = (hipStream_t)stream;
  int rc = build_table(b);
  if (rc != GP_OK) {
    delete b;
    return rc;
  }
  *out = b;
  return GP_OK;
}

// gp_multi.hip: several shards of one multi-batch run on this device and read the same source clouds one after the other, so the source
// stream of this batch must stay cacheable (no non-temporal policy)
extern "C++" {
namespace gp {
void batch_set_sources_shared(gp_vgicp_batch* batch, bool shared) {
  if (!batch || batch->shares_device_ok == !shared) return;
  batch->shares_device_ok = !shared;
  batch->table_dirty = true;
}
}  // namespace gp
}

// measurement: durations of the tile kernel and the finalize kernel of the last synchronous linearise (GP_TUNE_TIMING = 1), HIP events on the batch's stream
int gp_vgicp_batch_device_times(gp_vgicp_batch_t* batch, int reset, double* steps, double* stream_us_mean, double* kernel_us_mean) {
  if (!batch) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_device_times: null batch");
  if (steps) *steps = batch->dev_steps;
  if (stream_us_mean) *stream_us_mean = batch->dev_steps > 0 ? batch->dev_stream_ticks / batch->dev_steps / 100.0 : 0.0;
  if (kernel_us_mean) *kernel_us_mean = batch->dev_steps > 0 ? batch->dev_kernel_ticks / batch->dev_steps / 100.0 : 0.0;
  if (reset) batch->dev_steps = batch->dev_stream_ticks = batch->dev_kernel_ticks = 0.0;
  return GP_OK;
}

int gp_vgicp_batch_last_kernel_ms(const gp_vgicp_batch_t* batch, float* tile_ms, float* finalize_ms) {
  if (!batch) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_last_kernel_ms: null batch");
  if (tile_ms) *tile_ms = batch->last_tile_ms;
  if (finalize_ms) *finalize_ms = batch->last_finalize_ms;
  return GP_OK;
}

int gp_vgicp_batch_destroy(gp_vgicp_batch_t* batch) {
  if (!batch) return GP_OK;
  for (auto& e : batch->ev)
    if (e) (void)hipEventDestroy(e);
  // the staging buffers are about to be freed: the last H2D copy must have finished.  The stream itself is the caller's and may
  // already be gone (the reference's clone() drops it, integrated_vgicp_factor_gpu.cpp:122-134), so it is not synchronised here.
  if (batch->h2d_done) {
    (void)hipEventSynchronize(batch->h2d_done);
    (void)hipEventDestroy(batch->h2d_done);
  }
  delete batch;
  return GP_OK;
}

int gp_vgicp_batch_size(const gp_vgicp_batch_t* batch) { return batch ? (int)batch->factors.size() : 0; }
int64_t gp_vgicp_batch_total_points(const gp_vgicp_batch_t* batch) { return batch ? batch->total_points : 0; }

int64_t gp_vgicp_batch_algorithmic_bytes(const gp_vgicp_batch_t* batch) {
  if (!batch) return 0;
  int64_t bytes = 0;
  for (const auto* f : batch->factors)
    bytes += 48ll * f->n + 16ll * f->target->info.num_buckets + 52ll * f->target->info.num_voxels + 560ll + (f->surface_validation ? 12ll * f->n : 0ll);
  return bytes;
}

// what one pass of the batch's built table really requests from memory with perfect reuse of the lookup structures: the source stream as the kernel reads it
// (36 B per point through the packed mirrors, else 48; + 12 with surface validation), every DISTINCT map's block grid (16 B per 4x4x4 voxels of its box) or
// bucket table and its 64-B records once, pose in and record out.  Beside gp_vgicp_batch_algorithmic_bytes (the reference-layout accounting,
// which repacking does not change) in bench.py's roofline object.
int64_t gp_vgicp_batch_actual_bytes(gp_vgicp_batch_t* batch) {
  if (!batch) return 0;
  if (table_is_stale(batch) && build_table(batch) != GP_OK) return 0;
  int64_t bytes = 0;
  std::vector<const gp_voxelmap*> maps;
  for (const auto* f : batch->factors) {
    const int full = batch->packed ? f->n / gp::kChunkPoints * gp::kChunkPoints : 0;  // (the points behind the last full chunk come from the caller's arrays)
    bytes += 36ll * full + 48ll * (f->n - full) + 560ll + (f->surface_validation && f->normals ? 12ll * f->n : 0ll);
    if (std::find(maps.begin(), maps.end(), f->target) == maps.end()) maps.push_back(f->target);
  }
  for (const auto* m : maps)
    bytes += 64ll * m->info.num_voxels + (m->has_grid && batch->use_grid ? 16ll * m->gdim[0] * m->gdim[1] * m->gdim[2] : 16ll * m->info.num_buckets);
  return bytes;
}

int gp_vgicp_batch_issue_linearize(gp_vgicp_batch_t* b, const double* poses_host, gp_linearized6* out_dev) {
  if (!b || !poses_host || !out_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_issue_linearize: null");
  if (table_is_stale(b)) GP_TRY(build_table(b));
  if (b->factors.empty()) return GP_OK;
  PoseSource ps;
  GP_TRY(stage_poses(b, poses_host, nullptr, &ps));
  return launch_linearize(b, ps, out_dev, poses_are_rigid(poses_host, b->factors.size()));
}

int gp_vgicp_batch_issue_compute_error(gp_vgicp_batch_t* b, const double* poses_lin_host, const double* poses_eval_host, double* out_dev) {
  if (!b || !poses_lin_host || !poses_eval_host || !out_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_issue_compute_error: null");
  if (table_is_stale(b)) GP_TRY(build_table(b));
  if (b->factors.empty()) return GP_OK;
  PoseSource ps;
  GP_TRY(stage_poses(b, poses_lin_host, poses_eval_host, &ps));
  return launch_error(b, ps, out_dev);
}

static int clean_factor_arrivals(gp_vgicp_batch_t* b);
static bool words_arrived(const gp_vgicp_batch_t* b, int count, unsigned long long seq);

// the same two passes with the poses ALREADY in device memory (double[F][16] per table, column-major -- what a device-side retract produces, gp_lm.hip): no staging,
// no H2D copy, nothing for the host to wait on before the next call.  rigid: the caller vouches that every 3x3 block is orthonormal to 1e-9 (what the host-pose entry
// points test for themselves, poses_are_rigid): the 29-sum kernel + adjoint expansion; 0 = the 92-sum kernel, exact for any 3x3 block.  Any F (a single factor reads its descriptor and pose from the
// tables like the batch's other members: the in-argument form needs the pose on the host).
int gp_vgicp_batch_issue_linearize_dev(gp_vgicp_batch_t* b, const double* poses_dev, int rigid, gp_linearized6* out_dev) {
  if (!b || !poses_dev || !out_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_issue_linearize_dev: null");
  if (table_is_stale(b)) GP_TRY(build_table(b));
  if (b->factors.empty()) return GP_OK;
  PoseSource ps;
  ps.d_lin = poses_dev;
  ps.inl.use = 0;
  // (the by-factor fused finalize of the synchronous call was tried here too -- records written by the factors' last tile workgroups, no finalize launch -- and measured
  //  4 us SLOWER on BASELINE configs[2]'s batch: 59 us against 49.6 + 5.0, the 256 factors' tails end later than one finalize kernel over all of them; the error
  //  evaluation below keeps its fused form: 52 us against 49.5 + 9.2)
  return launch_linearize(b, ps, out_dev, rigid != 0);
}

int gp_vgicp_batch_issue_compute_error_dev(gp_vgicp_batch_t* b, const double* poses_lin_dev, const double* poses_eval_dev, double* out_dev) {
  if (!b || !poses_lin_dev || !poses_eval_dev || !out_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_issue_compute_error_dev: null");
  if (table_is_stale(b)) GP_TRY(build_table(b));
  if (b->factors.empty()) return GP_OK;
  PoseSource ps;
  ps.d_lin = poses_lin_dev;
  ps.d_eval = poses_eval_dev;
  ps.inl.use = 0;
  return launch_error(b, ps, out_dev);
}

// ... and the error evaluation as a SYNCHRONOUS call: the finalize kernel hands the F sums and a completion word each to the host (pinned), and the call polls the words
// (wait_done) instead of going through hipStreamSynchronize -- everything queued in front of it on the batch's stream is complete when it returns, whatever else was
// queued BEHIND it in the meantime is not waited for (gp_lm.hip queues the next linearise there)
int gp_vgicp_batch_compute_error_dev(gp_vgicp_batch_t* b, const double* poses_lin_dev, const double* poses_eval_dev, double* out_host) {
  if (!b || !poses_lin_dev || !poses_eval_dev || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_compute_error_dev: null");
  GP_TRY(gp_vgicp_batch_issue_compute_error_dev_begin(b, poses_lin_dev, poses_eval_dev));
  return gp_vgicp_batch_compute_error_dev_end(b, out_host);
}

// the two halves of the call above: begin launches (and returns the moment the kernels are queued), end polls and copies.  One begin at a time per batch.
int gp_vgicp_batch_issue_compute_error_dev_begin(gp_vgicp_batch_t* b, const double* poses_lin_dev, const double* poses_eval_dev) {
  if (!b || !poses_lin_dev || !poses_eval_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_issue_compute_error_dev_begin: null");
  if (table_is_stale(b)) GP_TRY(build_table(b));
  if (b->factors.empty()) return GP_OK;
  PoseSource ps;
  ps.d_lin = poses_lin_dev;
  ps.d_eval = poses_eval_dev;
  ps.inl.use = 0;
  const gp::DoneFlags done{static_cast<unsigned long long*>(b->h_done_dev), ++b->seq};
  const size_t F = b->factors.size();
  b->dev_error_fused = false;
  if (b->family == GP_KERNEL_STREAM && b->tuning.fused_finalize && !b->trace && !b->planned) {
    // ONE launch, as the synchronous call's by-factor form (gp_vgicp_batch_compute_error): the factor's last tile workgroup adds its rows up and hands sum and word over
    if (b->factor_arrive_count < F) {
      const size_t bytes = sizeof(unsigned long long) * gp::kFactorArriveStride * F;
      GP_TRY(b->d_factor_arrive.alloc(bytes));
      GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, bytes));
      b->factor_arrive_count = F;
    }
    double* partials = nullptr;
    GP_TRY(partials_ptr(b, &partials));
    GP_TRY(clean_factor_arrivals(b));
    ps.inl.arrive = b->d_factor_arrive.as<unsigned long long>();
    ps.inl.rows_per_part = 0;
    ps.inl.num_rows = b->num_tiles;
    ps.inl.fin_out = static_cast<double*>(b->h_out_dev);
    ps.inl.fin_stride = 1;
    ps.inl.fin_flags = done.flags;
    ps.inl.fin_seq = done.seq;
    b->factor_arrive_dirty = true;
    GP_TRY(launch_tiles<gp::MODE_ERR>(b, ps, partials));
    for (size_t i = 0; i < F; i++)
      if (b->h_descs[i].tile_count == 0) {
        static_cast<volatile double*>(b->h_out.ptr)[i] = 0.0;
        static_cast<volatile unsigned long long*>(b->h_done.ptr)[i] = done.seq;
      }
    b->dev_error_fused = true;
    return GP_OK;
  }
  return launch_error(b, ps, static_cast<double*>(b->h_out_dev), done);
}

int gp_vgicp_batch_compute_error_dev_end(gp_vgicp_batch_t* b, double* out_host) {
  if (!b || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_compute_error_dev_end: null");
  const size_t F = b->factors.size();
  if (F == 0) return GP_OK;
  GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, b->seq, b->stream, spin_budget_us(b) + 400));  // (+ the damped step queued in front of it)
  if (b->dev_error_fused) {
    if (!words_arrived(b, (int)F, b->seq)) {  // (as gp_vgicp_batch_compute_error: a counter left dirty by a launch that did not run to its end -- the finalize kernel does the sums)
      double* partials = nullptr;
      GP_TRY(partials_ptr(b, &partials));
      GP_HIP(hipStreamSynchronize(b->stream));
      GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, sizeof(unsigned long long) * gp::kFactorArriveStride * F));
      const gp::DoneFlags again{static_cast<unsigned long long*>(b->h_done_dev), ++b->seq};
      hipLaunchKernelGGL(gp::vgicp_finalize_error_kernel, dim3((int)F), dim3(gp::kBlockThreads), 0, b->stream, b->d_factors.as<gp::FactorDesc>(), (const double*)partials,
                         static_cast<double*>(b->h_out_dev), -1, again);
      GP_HIP(hipGetLastError());
      GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, again.seq, b->stream, spin_budget_us(b)));
    }
    b->factor_arrive_dirty = false;
    b->dev_error_fused = false;
  }
  memcpy(out_host, b->h_out.ptr, sizeof(double) * F);
  return GP_OK;
}

int gp_vgicp_batch_stream(const gp_vgicp_batch_t* b, gp_stream_t* out) {
  if (!b || !out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_stream: null");
  *out = (gp_stream_t)b->stream;
  return GP_OK;
}

int gp_vgicp_batch_sync(gp_vgicp_batch_t* b) {
  if (!b) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "null batch");
  GP_HIP(hipStreamSynchronize(b->stream));
  return GP_OK;
}

// the 6x6 expansion of the rigid finalize kernel on the host (same formulas: H_t from the 29 sums, Ad(delta), H_ts = -H_t Ad, H_s = Ad^T H_t Ad,
// b_s = -Ad^T b_t): used by the synchronous single-factor call, whose finalize parts then deliver only their sums
static void expand_rigid_host(const double* sum, const double* pose /*col-major 4x4*/, double* dst /*[122]*/) {
  constexpr int OFF_HT = 2, OFF_HS = 38, OFF_HTS = 74, OFF_BT = 110, OFF_BS = 116;
  const double Rl[9] = {pose[0], pose[4], pose[8], pose[1], pose[5], pose[9], pose[2], pose[6], pose[10]};  // row-major R
  const double tx = pose[12], ty = pose[13], tz = pose[14];
  const double Xl[9] = {0.0, -tz, ty, tz, 0.0, -tx, -ty, tx, 0.0};  // [t]x, row-major
  auto sym3 = [](int a, int b) {
    const int i = a < b ? a : b, j = a < b ? b : a;
    return (i * (5 - i)) / 2 + j;
  };
  double Ht[6][6], Ad[6][6], HtA[6][6], bt[6];
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      double h;
      if (r < 3 && c < 3) h = sum[gp::ACC_TL + sym3(r, c)];
      else if (r >= 3 && c < 3) h = -sum[gp::ACC_K + (r - 3) * 3 + c];
      else if (r < 3) h = -sum[gp::ACC_K + (c - 3) * 3 + r];
      else h = sum[gp::ACC_M + sym3(r - 3, c - 3)];
      Ht[r][c] = h;
      dst[OFF_HT + c * 6 + r] = h;
      double a;
      if (r < 3 && c < 3) a = Rl[r * 3 + c];
      else if (r < 3) a = 0.0;
      else if (c >= 3) a = Rl[(r - 3) * 3 + (c - 3)];
      else a = Xl[(r - 3) * 3] * Rl[c] + Xl[(r - 3) * 3 + 1] * Rl[3 + c] + Xl[(r - 3) * 3 + 2] * Rl[6 + c];
      Ad[r][c] = a;
    }
  for (int k = 0; k < 6; k++) {
    bt[k] = k < 3 ? sum[gp::ACC_QXMR + k] : sum[gp::ACC_MR + k - 3];
    dst[OFF_BT + k] = bt[k];
  }
  dst[0] = sum[gp::ACC_COUNT];
  dst[1] = sum[gp::ACC_ERR];
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      double a = 0.0;
      for (int k = 0; k < 6; k++) a += Ht[r][k] * Ad[k][c];
      HtA[r][c] = a;
      dst[OFF_HTS + c * 6 + r] = -a;
    }
  for (int k6 = 0; k6 < 6; k6++) {
    double a = 0.0;
    for (int k = 0; k < 6; k++) a += Ad[k][k6] * bt[k];
    dst[OFF_BS + k6] = -a;
  }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      double a = 0.0;
      for (int k = 0; k < 6; k++) a += Ad[k][r] * HtA[k][c];
      dst[OFF_HS + c * 6 + r] = a;
    }
}

// fused finalize: arrival counters (monotonic, one per part, kArriveStride words apart) and where the parts' last workgroups deliver
static int arm_arrival(gp_vgicp_batch_t* b, PoseSource* ps, int parts, int per, const gp::DoneFlags& done) {
  if (!b->d_arrive.ptr) {
    GP_TRY(b->d_arrive.alloc(sizeof(unsigned long long) * 16 * gp::kArriveStride));
    GP_HIP(hipMemset(b->d_arrive.ptr, 0, sizeof(unsigned long long) * 16 * gp::kArriveStride));
    memset(b->arrived, 0, sizeof(b->arrived));
  }
  ps->inl.arrive = b->d_arrive.as<unsigned long long>();
  ps->inl.rows_per_part = per;
  ps->inl.num_rows = b->num_tiles;
  for (int g = 0; g < parts; g++)  // (committed to b->arrived by fused_launched() once the launch has gone out)
    ps->inl.arrive_target[g] = b->arrived[g] + (unsigned long long)std::max(0, std::min(per, b->num_tiles - g * per));
  ps->inl.fin_out = static_cast<double*>(b->h_out_dev);
  ps->inl.fin_stride = (int)(sizeof(gp_linearized6) / sizeof(double));
  ps->inl.fin_flags = done.flags;
  ps->inl.fin_seq = done.seq;
  return GP_OK;
}

static void fused_launched(gp_vgicp_batch_t* b, const PoseSource& ps, int parts) {
  for (int g = 0; g < parts; g++) b->arrived[g] = ps.inl.arrive_target[g];
}
// did every part's completion word arrive?  (after wait_done, which hands over to hipStreamSynchronize when its spin budget is used up: in the two-kernel
// form a finished stream means finished records, in the fused form the records exist only if the parts' last workgroups saw their counters complete)
static bool words_arrived(const gp_vgicp_batch_t* b, int count, unsigned long long seq) {
  const volatile unsigned long long* fl = static_cast<const volatile unsigned long long*>(b->h_done.ptr);
  for (int g = 0; g < count; g++)
    if (fl[g] != seq) return false;
  return true;
}
// the arrival counters and the host's idea of them have come apart (a launch that failed half way, a kernel that was torn down): start over from zero
static int reset_arrival(gp_vgicp_batch_t* b) {
  GP_HIP(hipStreamSynchronize(b->stream));
  if (b->d_arrive.ptr) GP_HIP(hipMemset(b->d_arrive.ptr, 0, sizeof(unsigned long long) * 16 * gp::kArriveStride));
  memset(b->arrived, 0, sizeof(b->arrived));
  return GP_OK;
}

// The by-factor fused finalize counts arrivals in zero-reset counters (the last arriver of a factor stores 0): a launch that went out and was never seen to complete
// (an error return between launch and the last completion word, a torn-down kernel) may have left small positive values behind, which would let the NEXT launch's
// factors finalize early on stale rows without anybody noticing.  Such a batch cleans its counters, behind the stream, before it counts again.
static int clean_factor_arrivals(gp_vgicp_batch_t* b) {
  if (!b->factor_arrive_dirty || !b->d_factor_arrive.ptr) return GP_OK;
  GP_HIP(hipStreamSynchronize(b->stream));
  GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, sizeof(unsigned long long) * gp::kFactorArriveStride * b->factor_arrive_count));
  b->factor_arrive_dirty = false;
  return GP_OK;
}

// synchronous: the finalize kernel stores the records straight into host-mapped pinned memory (no D2H copy op).
// out_host != nullptr: the records are copied there; view != nullptr: *view points at them where they lie (the batch's own pinned
// buffer, or `view_store` for the single large factor whose parts the host combines) until the next call on the batch.
static int batch_linearize_sync(gp_vgicp_batch_t* b, const double* poses_host, gp_linearized6* out_host, const gp_linearized6** view) {
  const size_t F = b->factors.size();
  if (view) *view = nullptr;
  if (F == 0) return GP_OK;
  gp_linearized6 local;
  if (!out_host && F == 1) out_host = view ? &b->view_store : &local;  // (the combined record of a split finalize needs a home)
  if (table_is_stale(b)) GP_TRY(build_table(b));
  PoseSource ps;
  GP_TRY(stage_poses(b, poses_host, nullptr, &ps, true));
  const gp::DoneFlags done{static_cast<unsigned long long*>(b->h_done_dev), ++b->seq, b->trace ? b->trace + 2047 * 16 : nullptr};
  const bool rigid = poses_are_rigid(poses_host, F);
  const int parts = (F == 1 && rigid && b->num_tiles >= kFinalizeSplitTiles) ? finalize_parts() : 1;
  constexpr bool host_expand = true;  // the parts deliver their 32 sums, the host expands once (the parts expanding: measured slower, round 2)
  const bool sums_only = parts > 1 && host_expand;
  if (b->timing && !b->ev[0])
    for (auto& e : b->ev) GP_HIP(hipEventCreate(&e));
  const bool fused = sums_only && b->family == GP_KERNEL_STREAM && ps.inl.use && b->tuning.fused_finalize && !b->timing && parts <= 16;
  if (fused) {
    // ONE launch: the last workgroup of each part of the tile list finalizes the part (gp_vgicp_stream.hpp: finalize_part_rows)
    double* partials = nullptr;
    GP_TRY(partials_ptr(b, &partials));
    GP_TRY(arm_arrival(b, &ps, parts, (b->num_tiles + parts - 1) / parts, done));
    GP_TRY(launch_tiles<gp::MODE_LIN>(b, ps, partials));
    fused_launched(b, ps, parts);
    GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), (size_t)parts, done.seq, b->stream, spin_budget_us(b)));
    if (!words_arrived(b, parts, done.seq)) {
      // the stream is idle and the words are not there: the counters were out of step.  The rows are complete (the kernel has finished): reset the
      // counters and let the finalize kernel do this call's sums
      GP_TRY(reset_arrival(b));
      const gp::DoneFlags again{done.flags, ++b->seq, nullptr};
      GP_TRY(launch_finalize<false>(b, ps, partials, reinterpret_cast<gp_linearized6*>(b->h_out_dev), again, -parts));
      GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), (size_t)parts, again.seq, b->stream, spin_budget_us(b)));
    } else {
      // the step as the device saw it: first workgroup started .. last row in .. last sums out (words 34 / 32 / 33 of the parts' slots)
      const unsigned long long* w = static_cast<const unsigned long long*>(b->h_out.ptr);
      constexpr size_t N = sizeof(gp_linearized6) / sizeof(double);
      unsigned long long t0 = ~0ull, t_rows = 0, t_out = 0;
      for (int g = 0; g < parts; g++) {
        if (g < gp::kNumXCD) t0 = std::min(t0, w[g * N + 34]);
        t_rows = std::max(t_rows, w[g * N + 32]);
        t_out = std::max(t_out, w[g * N + 33]);
      }
      if (t0 <= t_rows && t_rows <= t_out && t_out - t0 < 100000000ull) {
        b->dev_steps += 1.0;
        b->dev_stream_ticks += (double)(t_rows - t0);
        b->dev_kernel_ticks += (double)(t_out - t0);
      }
    }
  } else if (parts == 1 && rigid && b->family == GP_KERNEL_STREAM && b->tuning.fused_finalize && !b->timing && !b->trace) {
    // ONE launch for a batch (or a small single factor): the workgroup that stores a factor's last row finalizes the factor (gp_vgicp_stream.hpp), so
    // the records leave for the host while the other factors' tiles are still running
    if (b->factor_arrive_count < F) {
      const size_t bytes = sizeof(unsigned long long) * gp::kFactorArriveStride * F;
      GP_TRY(b->d_factor_arrive.alloc(bytes));
      GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, bytes));
      b->factor_arrive_count = F;
    }
    double* partials = nullptr;
    GP_TRY(partials_ptr(b, &partials));
    GP_TRY(clean_factor_arrivals(b));
    ps.inl.arrive = b->d_factor_arrive.as<unsigned long long>();
    ps.inl.rows_per_part = 0;
    ps.inl.num_rows = b->num_tiles;
    ps.inl.fin_out = static_cast<double*>(b->h_out_dev);
    ps.inl.fin_stride = (int)(sizeof(gp_linearized6) / sizeof(double));
    ps.inl.fin_flags = done.flags;
    ps.inl.fin_seq = done.seq;
    b->factor_arrive_dirty = true;  // (cleared below once every completion word of THIS launch has been seen: an error return in between leaves it set)
    GP_TRY(launch_tiles<gp::MODE_LIN>(b, ps, partials));
    for (size_t i = 0; i < F; i++)
      if (b->h_descs[i].tile_count == 0) {  // a factor without points has no workgroup to finalize it: its (empty) record is written here
        const double zeros[32] = {0.0};
        expand_rigid_host(zeros, poses_host + 16 * i, reinterpret_cast<double*>(static_cast<gp_linearized6*>(b->h_out.ptr) + i));
        static_cast<volatile unsigned long long*>(b->h_done.ptr)[i] = done.seq;
      }
    GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, done.seq, b->stream, spin_budget_us(b)));
    if (!words_arrived(b, (int)F, done.seq)) {
      // (the stream is idle and a record is missing: a counter was left dirty by a launch that did not run to its end.  The rows are complete: clean the
      // counters and let the finalize kernel do this call's records)
      GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, sizeof(unsigned long long) * gp::kFactorArriveStride * F));
      const gp::DoneFlags again{done.flags, ++b->seq, nullptr};
      ps.inl.arrive = nullptr;
      GP_TRY(launch_finalize<false>(b, ps, partials, reinterpret_cast<gp_linearized6*>(b->h_out_dev), again, 1));
      GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, again.seq, b->stream, spin_budget_us(b)));
    }
    b->factor_arrive_dirty = false;  // every factor's last arriver reset its counter (or the fallback cleaned them)
  } else {
    GP_TRY(launch_linearize(b, ps, reinterpret_cast<gp_linearized6*>(b->h_out_dev), rigid, done, sums_only ? -parts : parts, b->timing));
    GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F * (size_t)parts, done.seq, b->stream, spin_budget_us(b)));
  }
  if (b->timing) {  // measurement: the two kernels of THIS synchronous pass, i.e. behind the idle queue the host left between two passes
    GP_HIP(hipEventSynchronize(b->ev[2]));
    GP_HIP(hipEventElapsedTime(&b->last_tile_ms, b->ev[0], b->ev[1]));
    GP_HIP(hipEventElapsedTime(&b->last_finalize_ms, b->ev[1], b->ev[2]));
  }
  if (parts == 1) {
    if (view) *view = static_cast<const gp_linearized6*>(b->h_out.ptr);
    if (out_host && !(view && F == 1)) memcpy(out_host, b->h_out.ptr, sizeof(gp_linearized6) * F);
  } else {
    if (view) *view = out_host;  // F == 1: caller's array or view_store
    const double* p = static_cast<const double*>(b->h_out.ptr);
    double* o = reinterpret_cast<double*>(out_host);
    constexpr int N = (int)(sizeof(gp_linearized6) / sizeof(double));
    if (sums_only) {
      // the parts' 32 sums add up in slot order; one expansion on the host
      double total[32];
      for (int k = 0; k < 32; k++) {
        double a = p[k];
        for (int q = 1; q < parts; q++) a += p[(size_t)q * N + k];
        total[k] = a;
      }
      expand_rigid_host(total, poses_host, o);
    } else {
      // the partial records add up entry by entry (all 122 scalars are linear in the tile sums), in slot order
      for (int k = 0; k < N; k++) {
        double a = p[k];
        for (int q = 1; q < parts; q++) a += p[(size_t)q * N + k];
        o[k] = a;
      }
    }
  }
  return GP_OK;
}

// host-side check hook (no device needed): the tiles a planned single-factor launch of n points deals to its workgroups, in tile-list order
// (XCD-major).  begin / count: arrays of `capacity` ints; *num_tiles = workgroups of the launch (<= 1024).
int gp_debug_stream_plan(int n, int skew_permille, const int* xcd_weights_permille, int capacity, int* begin, int* count, int* num_tiles) {
  if (n < 0 || skew_permille < -1 || !num_tiles) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_stream_plan: bad arguments");
  gp::StreamPlan plan;
  const int G = make_stream_plan(n, skew_permille, xcd_weights_permille, &plan);
  *num_tiles = G;
  if (begin && count) {
    if (capacity < G) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_stream_plan: capacity too small");
    int t = 0;
    for (int x = 0; x < gp::kNumXCD; x++)
      for (int q = 0; q < plan.wgs_per_xcd; q++, t++) gp::plan_tile(gp::plan_fields(plan, x, q), x, q, &begin[t], &count[t]);
  }
  return GP_OK;
}

// host-side check hook (no device needed): the expansion the synchronous single-factor call runs on the added sums of its finalize parts
int gp_debug_expand_rigid(const double sums[32], const double pose[16], gp_linearized6* out) {
  if (!sums || !pose || !out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_expand_rigid: null");
  expand_rigid_host(sums, pose, reinterpret_cast<double*>(out));
  return GP_OK;
}

int gp_vgicp_batch_linearize(gp_vgicp_batch_t* b, const double* poses_host, gp_linearized6* out_host) {
  if (!b || !poses_host || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_linearize: null");
  return batch_linearize_sync(b, poses_host, out_host, nullptr);
}

int gp_vgicp_batch_linearize_view(gp_vgicp_batch_t* b, const double* poses_host, const gp_linearized6** out_view) {
  if (!b || !poses_host || !out_view) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_linearize_view: null");
  return batch_linearize_sync(b, poses_host, nullptr, out_view);
}

int gp_vgicp_batch_compute_error(gp_vgicp_batch_t* b, const double* poses_lin_host, const double* poses_eval_host, double* out_host) {
  if (!b || !poses_lin_host || !poses_eval_host || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_compute_error: null");
  const size_t F = b->factors.size();
  if (F == 0) return GP_OK;
  if (table_is_stale(b)) GP_TRY(build_table(b));
  PoseSource ps;
  GP_TRY(stage_poses(b, poses_lin_host, poses_eval_host, &ps, true));
  const gp::DoneFlags done{static_cast<unsigned long long*>(b->h_done_dev), ++b->seq};
  const int parts = (F == 1 && ps.inl.use && b->family == GP_KERNEL_STREAM && b->num_tiles >= kFinalizeSplitTiles) ? finalize_parts() : 1;
  if (parts > 1 && parts <= 16) {
    // one large factor: eight part sums, added here in slot order -- by the part's last tile workgroup (fused, one launch) or by a second kernel
    double* partials = nullptr;
    GP_TRY(partials_ptr(b, &partials));
    const int per = (b->num_tiles + parts - 1) / parts;
    constexpr int kSlot = (int)(sizeof(gp_linearized6) / sizeof(double));
    bool fused = b->tuning.fused_finalize != 0;
    if (fused) {
      GP_TRY(arm_arrival(b, &ps, parts, per, done));
      GP_TRY(launch_tiles<gp::MODE_ERR>(b, ps, partials));
      fused_launched(b, ps, parts);
      GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), (size_t)parts, done.seq, b->stream, spin_budget_us(b)));
      if (!words_arrived(b, parts, done.seq)) {  // (see batch_linearize_sync)
        GP_TRY(reset_arrival(b));
        fused = false;
      }
    }
    if (!fused) {
      const gp::DoneFlags again{done.flags, b->tuning.fused_finalize ? ++b->seq : done.seq};
      if (!b->tuning.fused_finalize) GP_TRY(launch_tiles<gp::MODE_ERR>(b, ps, partials));
      hipLaunchKernelGGL(gp::vgicp_finalize_error_parts_kernel, dim3(parts), dim3(256), 0, b->stream, (const double*)partials, b->num_tiles, per,
                         static_cast<double*>(b->h_out_dev), kSlot, again);
      GP_HIP(hipGetLastError());
      GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), (size_t)parts, again.seq, b->stream, spin_budget_us(b)));
    }
    const double* p = static_cast<const double*>(b->h_out.ptr);
    double a = p[0];
    for (int q = 1; q < parts; q++) a += p[(size_t)q * kSlot];
    out_host[0] = a;
    return GP_OK;
  }
  if (b->family == GP_KERNEL_STREAM && b->tuning.fused_finalize && !b->trace) {
    // ONE launch (round 4): the workgroup that stores a factor's last row adds the factor's rows up -- in the order of vgicp_finalize_error_kernel -- and hands the sum and
    // the completion word to the host (the by-factor form of the linearise, gp_vgicp_stream.hpp); small single factors and batches alike
    if (b->factor_arrive_count < F) {
      const size_t bytes = sizeof(unsigned long long) * gp::kFactorArriveStride * F;
      GP_TRY(b->d_factor_arrive.alloc(bytes));
      GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, bytes));
      b->factor_arrive_count = F;
    }
    double* partials = nullptr;
    GP_TRY(partials_ptr(b, &partials));
    GP_TRY(clean_factor_arrivals(b));
    ps.inl.arrive = b->d_factor_arrive.as<unsigned long long>();
    ps.inl.rows_per_part = 0;
    ps.inl.num_rows = b->num_tiles;
    ps.inl.fin_out = static_cast<double*>(b->h_out_dev);
    ps.inl.fin_stride = 1;  // one double per factor
    ps.inl.fin_flags = done.flags;
    ps.inl.fin_seq = done.seq;
    b->factor_arrive_dirty = true;
    GP_TRY(launch_tiles<gp::MODE_ERR>(b, ps, partials));
    for (size_t i = 0; i < F; i++)
      if (b->h_descs[i].tile_count == 0) {  // no points, no workgroup: the empty sum is written here
        static_cast<volatile double*>(b->h_out.ptr)[i] = 0.0;
        static_cast<volatile unsigned long long*>(b->h_done.ptr)[i] = done.seq;
      }
    GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, done.seq, b->stream, spin_budget_us(b)));
    if (!words_arrived(b, (int)F, done.seq)) {  // a counter left dirty by a launch that did not run to its end: clean them, the finalize kernel does this call's sums
      GP_HIP(hipMemset(b->d_factor_arrive.ptr, 0, sizeof(unsigned long long) * gp::kFactorArriveStride * F));
      const gp::DoneFlags again{done.flags, ++b->seq};
      hipLaunchKernelGGL(gp::vgicp_finalize_error_kernel, dim3((int)F), dim3(gp::kBlockThreads), 0, b->stream, b->d_factors.as<gp::FactorDesc>(), (const double*)partials,
                         static_cast<double*>(b->h_out_dev), -1, again);
      GP_HIP(hipGetLastError());
      GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, again.seq, b->stream, spin_budget_us(b)));
    }
    b->factor_arrive_dirty = false;
    memcpy(out_host, b->h_out.ptr, sizeof(double) * F);
    return GP_OK;
  }
  GP_TRY(launch_error(b, ps, reinterpret_cast<double*>(b->h_out_dev), done));
  GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(b->h_done.ptr), F, done.seq, b->stream, spin_budget_us(b)));
  memcpy(out_host, b->h_out.ptr, sizeof(double) * F);
  return GP_OK;
}

int gp_vgicp_batch_time_linearize(gp_vgicp_batch_t* b, const double* poses_host, int iters, float* ms_total, float* ms_main_kernel, float* ms_finalize_kernel) {
  if (!b || !poses_host || iters <= 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_batch_time_linearize: bad arguments");
  if (table_is_stale(b)) GP_TRY(build_table(b));
  const size_t F = b->factors.size();
  if (F == 0) return GP_OK;
  // the device work of the SYNCHRONOUS call (gp_vgicp_batch_linearize), incl. its split finalize for a single large factor
  const int parts = (F == 1 && poses_are_rigid(poses_host, F) && b->num_tiles >= kFinalizeSplitTiles) ? finalize_parts() : 1;
  gp::DeviceArray d_out;
  GP_TRY(d_out.alloc(sizeof(gp_linearized6) * F * (size_t)parts));
  PoseSource ps;
  GP_TRY(stage_poses(b, poses_host, nullptr, &ps));
  double* partials = nullptr;
  GP_TRY(partials_ptr(b, &partials));
  const bool rigid = poses_are_rigid(poses_host, F);
  hipEvent_t e0, e1, e2;
  GP_HIP(hipEventCreate(&e0));
  GP_HIP(hipEventCreate(&e1));
  GP_HIP(hipEventCreate(&e2));
  GP_TRY(launch_linearize(b, ps, d_out.as<gp_linearized6>(), rigid, {}, parts));  // warm-up
  GP_HIP(hipStreamSynchronize(b->stream));
  // whole pass, back to back
  GP_HIP(hipEventRecord(e0, b->stream));
  for (int i = 0; i < iters; i++) GP_TRY(launch_linearize(b, ps, d_out.as<gp_linearized6>(), rigid, {}, parts));
  GP_HIP(hipEventRecord(e1, b->stream));
  GP_HIP(hipEventSynchronize(e1));
  float t_total = 0.f;
  GP_HIP(hipEventElapsedTime(&t_total, e0, e1));
  // main kernel alone, then finalize alone (same stream the product path launches on)
  GP_HIP(hipEventRecord(e0, b->stream));
  for (int i = 0; i < iters; i++) GP_TRY(rigid ? launch_tiles<gp::MODE_LIN>(b, ps, partials) : launch_tiles<gp::MODE_LIN_GENERAL>(b, ps, partials));
  GP_HIP(hipEventRecord(e1, b->stream));
  for (int i = 0; i < iters; i++)
    GP_TRY(rigid ? launch_finalize<false>(b, ps, partials, d_out.as<gp_linearized6>(), {}, parts) : launch_finalize<true>(b, ps, partials, d_out.as<gp_linearized6>()));
  GP_HIP(hipEventRecord(e2, b->stream));
  GP_HIP(hipEventSynchronize(e2));
  float t_main = 0.f, t_fin = 0.f;
  GP_HIP(hipEventElapsedTime(&t_main, e0, e1));
  GP_HIP(hipEventElapsedTime(&t_fin, e1, e2));
  if (ms_total) *ms_total = t_total / (float)iters;
  if (ms_main_kernel) *ms_main_kernel = t_main / (float)iters;
  if (ms_finalize_kernel) *ms_finalize_kernel = t_fin / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(e2);
  return GP_OK;
}

}  // extern "C"
