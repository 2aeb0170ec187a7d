// gp_vgicp_finalize.hpp -- the pieces of the rigid (29-sum) finalize that more than one kernel runs: the ordered sum of a factor's partial rows and the 6x6
// expansion of the 32 sums into a LinearizedSystem6 record.  vgicp_finalize_rigid_kernel (gp_vgicp.hip: one workgroup per factor behind the tile kernel) and
// the tile kernel's own fused tail (gp_vgicp_stream.hpp: the workgroup that stores a factor's last row finalizes the factor) go through the SAME functions in
// the SAME order, so a record does not depend on which of the two produced it (tests: plain batch == sharded batch, bit for bit).
#pragma once
#include "gp_vgicp_shared.hpp"

namespace gp {

// visibility of LDS writes between the lanes of ONE wave (its LDS operations execute in program order; this only keeps the compiler
// from moving them and makes it wait for the writes): what __syncthreads() is for a workgroup, without the s_barrier
#define GP_WAVE_SYNC()                                       \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
  } while (0)

// sum of r^T M r over a factor's partial rows in the order of vgicp_finalize_error_kernel: thread t of 256 adds rows t, t + 256, ...; the wave's 64 values meet in a
// shuffle-down tree; the four waves' sums are added in wave order.  All 256 threads of the workgroup call it; the total is returned on thread 0.  SC1: the rows were
// stored write-through by other workgroups of this launch (the fused finalize of the tile kernel): read past this XCD's L2 view.  wsum: 4 doubles of LDS.
template <bool SC1>
__device__ __forceinline__ double error_factor_total(const double* __restrict__ partials, int tile_begin, int tile_count, double* wsum) {
  double s = 0.0;
  for (int t = threadIdx.x; t < tile_count; t += 256) {
    const double* p = partials + (size_t)(tile_begin + t) * ACC_STRIDE + ACC_ERR;
    s += SC1 ? __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : *p;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  double a = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < 4; w++) a += wsum[w];
  return a;
}

__device__ __forceinline__ double pick9(int i, double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7, double a8) {
  double v = a0;
  v = i == 1 ? a1 : v;
  v = i == 2 ? a2 : v;
  v = i == 3 ? a3 : v;
  v = i == 4 ? a4 : v;
  v = i == 5 ? a5 : v;
  v = i == 6 ? a6 : v;
  v = i == 7 ? a7 : v;
  v = i == 8 ? a8 : v;
  return v;
}

// LDS of the expansion (one wave works on it)
struct RigidScratch {
  double sum[32];
  double Rl[9], Xl[9];  // R and [t]x, row-major
  double Ht[6][6], Ad[6][6], HtA[6][6], bt[6];
  double dst[122];
};

// one lane's share of the row sum: rows slice, slice + KSLICES, ... of `base` (= the factor's first row + the lane's component), 32 at a time with a
// fixed pairwise tree, batches added in order.  SC1: the rows were stored write-through by other workgroups of THIS launch (fused tail): read past L1
template <int KSLICES, bool SC1>
__device__ __forceinline__ double rigid_slice_total(const double* __restrict__ base, const int tile_count, const int slice) {
  double total = 0.0;
  for (int t0 = slice; t0 < tile_count; t0 += 32 * KSLICES) {
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
      const int t = t0 + k * KSLICES;
      if constexpr (SC1)
        v[k] = t < tile_count ? __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(base) + (size_t)t * ACC_STRIDE, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_AGENT))
                              : 0.0;
      else
        v[k] = t < tile_count ? base[(size_t)t * ACC_STRIDE] : 0.0;
    }
#pragma unroll
    for (int w = 16; w > 0; w >>= 1) {
#pragma unroll
      for (int k = 0; k < w; k++) v[k] += v[k + w];
    }
    total += v[0];
  }
  return total;
}

// the waves' sums of one component meet: pairwise tree over wsum[KWAVES][32]
template <int KWAVES>
__device__ __forceinline__ double rigid_wave_tree(const double* __restrict__ wsum, const int comp) {
  double v[KWAVES];
#pragma unroll
  for (int k = 0; k < KWAVES; k++) v[k] = wsum[k * 32 + comp];
#pragma unroll
  for (int w = KWAVES / 2; w > 0; w >>= 1) {
#pragma unroll
    for (int k = 0; k < w; k++) v[k] += v[k + w];
  }
  return v[0];
}

// ONE wave (all 64 lanes call): S.sum[0..31] -> S.dst[0..121], the record in the layout of gp_linearized6
__device__ __forceinline__ void rigid_expand_wave(RigidScratch& S, const Pose& T, const int lane) {
  if (lane >= 32 && lane < 41) {
    const int i = lane - 32;
    S.Rl[i] = pick9(i, T.r00, T.r01, T.r02, T.r10, T.r11, T.r12, T.r20, T.r21, T.r22);
    S.Xl[i] = pick9(i, 0.0, -T.tz, T.ty, T.tz, 0.0, -T.tx, -T.ty, T.tx, 0.0);
  }
  GP_WAVE_SYNC();
  const int t = lane;
  const int r = t / 6, c = t % 6;  // t < 36: one 6x6 entry per lane
  constexpr int OFF_HT = 2, OFF_HS = 38, OFF_HTS = 74, OFF_BT = 110, OFF_BS = 116;
  auto sym3 = [](int a, int b) {  // packed index of a symmetric 3x3 (00 01 02 11 12 22)
    const int i = a < b ? a : b, j = a < b ? b : a;
    return (i * (5 - i)) / 2 + j;
  };
  if (t < 36) {
    // H_t = [[TL, -K^T], [-K, M]]
    double h;
    if (r < 3 && c < 3) {
      h = S.sum[ACC_TL + sym3(r, c)];
    } else if (r >= 3 && c < 3) {
      h = -S.sum[ACC_K + (r - 3) * 3 + c];
    } else if (r < 3) {
      h = -S.sum[ACC_K + (c - 3) * 3 + r];
    } else {
      h = S.sum[ACC_M + sym3(r - 3, c - 3)];
    }
    S.Ht[r][c] = h;
    S.dst[OFF_HT + c * 6 + r] = h;
    // Ad(delta) = [[R, 0], [[t]x R, R]]   ([omega, v] ordering, GTSAM Pose3::AdjointMap)
    double a;
    if (r < 3 && c < 3) {
      a = S.Rl[r * 3 + c];
    } else if (r < 3) {
      a = 0.0;
    } else if (c >= 3) {
      a = S.Rl[(r - 3) * 3 + (c - 3)];
    } else {
      a = S.Xl[(r - 3) * 3] * S.Rl[c] + S.Xl[(r - 3) * 3 + 1] * S.Rl[3 + c] + S.Xl[(r - 3) * 3 + 2] * S.Rl[6 + c];
    }
    S.Ad[r][c] = a;
  } else if (t < 42) {
    const int k = t - 36;
    const double b = k < 3 ? S.sum[ACC_QXMR + k] : S.sum[ACC_MR + k - 3];
    S.bt[k] = b;
    S.dst[OFF_BT + k] = b;
  } else if (t == 42) {
    S.dst[0] = S.sum[ACC_COUNT];
    S.dst[1] = S.sum[ACC_ERR];
  }
  GP_WAVE_SYNC();
  if (t < 36) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) a += S.Ht[r][k] * S.Ad[k][c];
    S.HtA[r][c] = a;
    S.dst[OFF_HTS + c * 6 + r] = -a;  // H_ts = -H_t Ad
  } else if (t < 42) {
    const int k6 = t - 36;
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) a += S.Ad[k][k6] * S.bt[k];
    S.dst[OFF_BS + k6] = -a;  // b_s = -Ad^T b_t
  }
  GP_WAVE_SYNC();
  if (t < 36) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) a += S.Ad[k][r] * S.HtA[k][c];
    S.dst[OFF_HS + c * 6 + r] = a;  // H_s = Ad^T H_t Ad
  }
  GP_WAVE_SYNC();
}

}  // namespace gp
