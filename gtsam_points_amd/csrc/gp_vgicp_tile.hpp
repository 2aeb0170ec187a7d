// gp_vgicp_tile.hpp -- the tuned tile kernel of the VGICP path (rigid poses; MODE_LIN / MODE_ERR).
//
// Same arithmetic as accumulate_point<> in gp_vgicp.hip (the reference-shaped kernel), restructured around what the
// micro-benchmarks say about gfx950 (scripts/stream_bench.py, scripts/alu_rate.py):
//   * the 48 B/point source stream, the gather of the voxel data and the f64 algebra each cost 7-9 us per million points and
//     ADD UP when a wave does them one after the other;
//   * so the source travels by LDS-DMA (no VGPRs while in flight) in a rolling 3-stage ring, requested two steps ahead;
//   * the voxel lookup is two dependent round trips with no data-dependent loop: one 16-B occupancy-block entry (4x4x4 voxels:
//     64 occupancy bits + the index of the block's first voxel, voxels numbered in block order -> index = base + popcount),
//     then the 64-B voxel record.  80 B of gather per point instead of the 128 B of a hashed key line + record, no hashing,
//     no key compares, and the block grid (16 B per 64 cells) stays in L2 / L1.  Maps whose bounding box is too large for the
//     grid keep the hashed line table (template parameter GRID = false);
//   * all waits are hand-placed (vmcnt retires in order; the compiler cannot count across the DMA requests).
//   All pointers are cast to the global address space (descriptors loaded from memory would otherwise make hipcc emit
//   flat_load, which also ties up lgkmcnt).
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

#define GP_LDS __attribute__((address_space(3)))

// per-correspondence algebra shared by the VGICP pipeline kernel and the GICP kernel.
//   a[6]        source covariance C_A, symmetric part, (xx xy xz yy yz zz) in f64
//   c01 c23 c45 target covariance C_B (xx xy | xz yy | yz zz) in f64
//   r           mu_B - q   (residual, f64)          q   transformed source point (f64; only the linearise uses it)
// M = (C_B + R C_A R^T)^-1 is always formed in f64: the inverse amplifies an input error by the condition number of the
// fused covariance (~10^3 for the regularised (1e-3, 1, 1) covariances), so 6e-8 from an f32 product would become 3e-5 on M,
// identical for every point of a planar patch, i.e. not averaged away by the sum.  What follows the inverse has no such
// amplification; acc_t = float computes it at twice the issue rate (each lane adds only PPT points before the f64 reduction).
template <int MODE, typename acc_t, typename rq_t>
__device__ __forceinline__ void accumulate_core(const Pose& Tl, const double* a, const v2d& c01, const v2d& c23, const v2d& c45, rq_t rxd, rq_t ryd, rq_t rzd, rq_t qx,
                                                rq_t qy, rq_t qz, acc_t* acc) {
  double m[6];
  {
    const double a00 = a[0], a01 = a[1], a02 = a[2], a11 = a[3], a12 = a[4], a22 = a[5];
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
  const acc_t M0 = (acc_t)m[0], M1 = (acc_t)m[1], M2 = (acc_t)m[2], M3 = (acc_t)m[3], M4 = (acc_t)m[4], M5 = (acc_t)m[5];
  const acc_t RX = (acc_t)rxd, RY = (acc_t)ryd, RZ = (acc_t)rzd;
  const acc_t mrx = M0 * RX + M1 * RY + M2 * RZ, mry = M1 * RX + M3 * RY + M4 * RZ, mrz = M2 * RX + M4 * RY + M5 * RZ;
  acc[ACC_COUNT] += (acc_t)1;
  acc[ACC_ERR] += RX * mrx + RY * mry + RZ * mrz;
  if constexpr (MODE == MODE_LIN) {
    const acc_t QX = (acc_t)qx, QY = (acc_t)qy, QZ = (acc_t)qz;
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

// GICP entry: the target mean mu_B is given in f64 (the matched target point); the source covariance arrives as the nine
// floats of the column-major 3x3 and is symmetrised in f64 (exactly the input when it is symmetric)
template <int MODE, typename acc_t>
__device__ __forceinline__ void accumulate_terms_mu(const Pose& Tl, const Pose& Te, float pxf, float pyf, float pzf, const float* cA9, double mux, double muy, double muz,
                                                    const v2d& c01, const v2d& c23, const v2d& c45, acc_t* acc) {
  const double px = (double)pxf, py = (double)pyf, pz = (double)pzf;
  const double a[6] = {(double)cA9[0], 0.5 * ((double)cA9[3] + (double)cA9[1]), 0.5 * ((double)cA9[6] + (double)cA9[2]),
                       (double)cA9[4], 0.5 * ((double)cA9[7] + (double)cA9[5]), (double)cA9[8]};
  const double qx = Te.r00 * px + Te.r01 * py + Te.r02 * pz + Te.tx;
  const double qy = Te.r10 * px + Te.r11 * py + Te.r12 * pz + Te.ty;
  const double qz = Te.r20 * px + Te.r21 * py + Te.r22 * pz + Te.tz;
  accumulate_core<MODE, acc_t, double>(Tl, a, c01, c23, c45, mux - qx, muy - qy, muz - qz, qx, qy, qz, acc);
}

// the source covariance of this lane from the nine floats of its column-major 3x3 (LDS or global): upper triangle when the
// matrix is symmetric -- the wave-uniform common case, estimate_covariances output rounds to a symmetric float matrix -- and the
// f64 mean of the two triangles otherwise (the symmetric part, which is what the reference's full 3x3 algebra sees to first order)
template <typename P>
__device__ __forceinline__ void load_cov6(P c9, double* a) {
  const float u01 = c9[3], u02 = c9[6], u12 = c9[7], l10 = c9[1], l20 = c9[2], l21 = c9[5];
  a[0] = (double)c9[0];
  a[1] = (double)u01;
  a[2] = (double)u02;
  a[3] = (double)c9[4];
  a[4] = (double)u12;
  a[5] = (double)c9[8];
  if (__builtin_amdgcn_ballot_w64(u01 != l10 || u02 != l20 || u12 != l21) != 0) {
    a[1] = 0.5 * (a[1] + (double)l10);
    a[2] = 0.5 * (a[2] + (double)l20);
    a[4] = 0.5 * (a[4] + (double)l21);
  }
}

// optional per-workgroup phase timestamps (s_memtime) for timeline analysis: [num_tiles][16] uint64 (slots 0-7 phases, 8 HW_ID,
// 9 XCC_ID), enabled by the host
#define GP_TRACE(slot)                                                                                 \
  do {                                                                                                 \
    if constexpr (TRACE) {                                                                             \
      if (trace && threadIdx.x == 0) trace[(size_t)tile_idx * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    }                                                                                                  \
  } while (0)

// =====================================================================================================================
// vgicp_pipeline_kernel -- rolling LDS-DMA source pipeline, one 64-point chunk per wave step.
//
// A workgroup owns a tile of 4 waves x PPT chunks x 64 points.  Per wave:
//   prologue   request chunks 0 and 1 (3 full-wave 16-B DMA instructions each) into stages 0 and 1 of its 3-stage LDS ring
//   step j     read point + covariance of this lane from stage j%3 (strides of 3 and 9 dwords: bank-conflict free)
//              f64 transform, floor, hash
//              hop 1: the 4 keys of the home line of the line table (one 64-B line, one round trip)   -> voxel index
//              hop 2: the 64-B voxel record; right behind it the DMA request for chunk j+2 (its stage held chunk j-1)
//              wait "all but the 3 youngest" (vmcnt retires in order) -> the record is here, chunk j+2 keeps travelling
//              while the f64 algebra of chunk j runs
//   epilogue   transposing butterfly across the 64 lanes, 4-wave sum through LDS, one 32-double partial per tile
// LDS: 3 stages x 3 KB per wave = 36 KB per workgroup -> 4 workgroups (16 waves) per CU at <= 128 VGPRs.
// A wave whose rows are not all there (last tile of a factor) or whose base pointers are not 16-B aligned reads its
// points with plain per-lane loads instead (same arithmetic, same order).
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

// voxel lookup with hand-placed waits: the loads are issued from inline asm so that the compiler does not track them (its own
// vmcnt bookkeeping would fold the younger DMA requests into the wait); the *_wait() helpers tie the destination registers to
// the s_waitcnt so that no use can be scheduled ahead of it.
// line-table lookup, hop 1: the four keys of the home line; hop 2: the 64-B record.  WAIT_YOUNGER = how many younger VMEM
// instructions (the next chunk's DMA requests) may stay in flight when the record is needed
__device__ __forceinline__ void line_issue(const GP_GLOBAL char* line, v4i& k0, v4i& k1, v4i& k2, v4i& k3) {
  asm volatile(
    "global_load_dwordx4 %0, %4, off\n\t"
    "global_load_dwordx4 %1, %4, off offset:16\n\t"
    "global_load_dwordx4 %2, %4, off offset:32\n\t"
    "global_load_dwordx4 %3, %4, off offset:48"
    : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
    : "v"(line)
    : "memory");
}
__device__ __forceinline__ void line_wait(v4i& k0, v4i& k1, v4i& k2, v4i& k3) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3) : : "memory");
}
__device__ __forceinline__ void record_issue(const GP_GLOBAL char* rec, v4f& head, v2d& c01, v2d& c23, v2d& c45) {
  asm volatile(
    "global_load_dwordx4 %0, %4, off\n\t"
    "global_load_dwordx4 %1, %4, off offset:16\n\t"
    "global_load_dwordx4 %2, %4, off offset:32\n\t"
    "global_load_dwordx4 %3, %4, off offset:48"
    : "=&v"(head), "=&v"(c01), "=&v"(c23), "=&v"(c45)
    : "v"(rec)
    : "memory");
}
template <int WAIT_YOUNGER>
__device__ __forceinline__ void record_wait(v4f& head, v2d& c01, v2d& c23, v2d& c45) {
  if constexpr (WAIT_YOUNGER == 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  } else {
    static_assert(WAIT_YOUNGER == 3, "one chunk request = 3 DMA instructions");
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  }
}
__device__ __forceinline__ int line_match(const v4i& k0, const v4i& k1, const v4i& k2, const v4i& k3, int cx, int cy, int cz) {
  int idx = -1;
  if (k0.w >= 0 && k0.x == cx && k0.y == cy && k0.z == cz) idx = k0.w;
  if (k1.w >= 0 && k1.x == cx && k1.y == cy && k1.z == cz) idx = k1.w;
  if (k2.w >= 0 && k2.x == cx && k2.y == cy && k2.z == cz) idx = k2.w;
  if (k3.w >= 0 && k3.x == cx && k3.y == cy && k3.z == cz) idx = k3.w;
  return idx;
}

// hop 1 of the grid lookup: the 16-B occupancy-block entry
__device__ __forceinline__ void grid_issue(const GP_GLOBAL char* entry, v4i& blk) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(blk) : "v"(entry) : "memory");
}
template <int WAIT_YOUNGER = 0>
__device__ __forceinline__ void grid_wait(v4i& blk) {
  if constexpr (WAIT_YOUNGER == 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(blk) : : "memory");
  } else if constexpr (WAIT_YOUNGER == 1) {
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(blk) : : "memory");
  } else {
    static_assert(WAIT_YOUNGER == 3, "one chunk request = 3 DMA instructions");
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(blk) : : "memory");
  }
}

// HW_ID register fields (s_getreg_b32): wave slot within the SIMD, and the whole word / the XCC id for the timeline traces
#define GP_GETREG_WAVE_SLOT ((3 << 11) | (0 << 6) | 4)  // HW_REG_HW_ID[3:0]
#define GP_GETREG_HW_ID ((31 << 11) | (0 << 6) | 4)
#define GP_GETREG_XCC_ID ((31 << 11) | (0 << 6) | 20)

// Waves per SIMD: 4 with f32 accumulators (<= 128 VGPRs), 3 with f64 accumulators (<= 168 VGPRs).  The kernel must not spill:
// scratch traffic counts in vmcnt and would break the hand-placed waits (tests/test_build_cpu.py checks the resource usage).
// LEAN: the prologue requests only chunk 0 (everybody's first burst is half as large, so it lands sooner); chunk 1 is requested
// right behind the first lookup's hop 1.
// AHEAD (block grid, f32 outer products, linearise): hop 1 of chunk j+1 is issued BEFORE hop 1 of chunk j is consumed, so it travels
// together with hop 2 of chunk j -- one exposed round trip per step instead of two.  What chunk j+1 needs after its lookup (cell bit,
// centre - l and q as floats, the block entry: 12 registers) stays live through the algebra of chunk j; the kernel has the room
// (105 VGPRs without it, 128 allowed at four waves per SIMD).
template <int MODE, bool OUTER_F32, int PPT, bool GRID, bool TRACE = false, bool LEAN = false, bool AHEAD = false>
__global__ void __launch_bounds__(256, (OUTER_F32 || MODE == MODE_ERR) ? 4 : 3) vgicp_pipeline_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                             const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                             double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  constexpr int STAGES = 3;
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGES * kChunkBytes];  // 36 KB
  // XCD-aware workgroup -> tile map (workgroup b runs on XCD b % 8)
  int tile_idx;
  if (inl.xcd_chunk > 0) {
    const int c = inl.xcd_chunk, x = blockIdx.x % kNumXCD, q = blockIdx.x / kNumXCD;
    tile_idx = ((q / c) * kNumXCD + x) * c + (q % c);
  } else {
    const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
    tile_idx = (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;
  }
  if (tile_idx >= num_tiles) return;
  unsigned long long* trace = TRACE ? inl.trace : nullptr;
  GP_TRACE(0);
  if constexpr (TRACE) {
    // s_memtime (the stamps above) runs at the shader clock but is not synchronised across compute units: only differences inside
    // one workgroup mean anything.  s_memrealtime is the 100 MHz constant clock shared by the whole device: start / end of every
    // workgroup on one time axis (10 ns resolution) for the ramp and the tail of the launch.
    if (trace && threadIdx.x == 0) trace[(size_t)tile_idx * 16 + 10] = __builtin_amdgcn_s_memrealtime();
    if (trace && threadIdx.x == 0) {
      trace[(size_t)tile_idx * 16 + 8] = __builtin_amdgcn_s_getreg(GP_GETREG_HW_ID);
      trace[(size_t)tile_idx * 16 + 9] = __builtin_amdgcn_s_getreg(GP_GETREG_XCC_ID);
    }
  }
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
    tile.row = tile_idx;
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
    if (PPT > 1 && (!LEAN || AHEAD)) chunk_dma(points, covs, first + kChunkPoints, wbase + kChunkBytes, lane);
  }

  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;

  // phase stagger (tuning knob, 0 = off): the waves that share a SIMD start together and would gather / compute in lock-step;
  // the odd wave slots of every SIMD wait `stagger` x 512 clocks here, with their source requests already in flight, so that
  // one half of a SIMD's waves computes while the other half waits for its lookups
  if (inl.stagger > 0 && (__builtin_amdgcn_s_getreg(GP_GETREG_WAVE_SLOT) & 1)) {
    for (int i = 0; i < inl.stagger; i++) __builtin_amdgcn_s_sleep(8);
  }

  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  const GP_GLOBAL char* lines = (const GP_GLOBAL char*)f.map.plines;
  const GP_GLOBAL char* gblocks = (const GP_GLOBAL char*)f.map.gblocks;
  const GP_GLOBAL char* records = (const GP_GLOBAL char*)f.map.records;

  // one chunk: transform, two-hop voxel lookup, [request for chunk j+2], algebra
  // (the covariance of the point is fetched by `load_a` only after the lookup: 12 fewer live registers while the gathers fly)
  auto step = [&](auto ring_tag, int j, bool active, float px, float py, float pz, auto load_a) {
    constexpr bool RING = decltype(ring_tag)::value;
    const double dx = (double)px, dy = (double)py, dz = (double)pz;
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    // voxel coordinate = floor(l * (1 / leaf)): the CPU map's fast_floor (util/fast_floor.hpp:12-15) for every in-range value.
    // (centre - l) = leaf (floor(u) + 0.5 - u) is formed here, so the large coordinates never meet (|error| ~ 1e-14 m)
    // With f32 outer products the linearise keeps (centre - l) and q = l as floats from here on: they are only ever used in
    // f32 after the lookup, and the six registers matter while the gathers are in flight.
    using rq_t = typename std::conditional<OUTER_F32 && MODE == MODE_LIN, float, double>::type;
    int cx, cy, cz;
    rq_t ex, ey, ez;  // voxel centre - l
    {
      const double ux = lx * f.map.inv_leaf, uy = ly * f.map.inv_leaf, uz = lz * f.map.inv_leaf;
      const double fx = __builtin_floor(ux), fy = __builtin_floor(uy), fz = __builtin_floor(uz);
      cx = (int)fx;
      cy = (int)fy;
      cz = (int)fz;
      ex = (rq_t)(f.map.leaf * ((fx + 0.5) - ux));
      ey = (rq_t)(f.map.leaf * ((fy + 0.5) - uy));
      ez = (rq_t)(f.map.leaf * ((fz + 0.5) - uz));
    }
    const rq_t qx = (rq_t)lx, qy = (rq_t)ly, qz = (rq_t)lz;
    bool live = active && finite3(px, py, pz);
    if (f.surface_validation && live && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (first + (size_t)j * kChunkPoints + lane))) live = false;
    int idx = -1;
    if constexpr (GRID) {
      // hop 1: the occupancy entry of the 4x4x4 block (lanes outside the box read entry 0 and are masked)
      const int bx = (cx >> 2) - f.map.glo[0], by = (cy >> 2) - f.map.glo[1], bz = (cz >> 2) - f.map.glo[2];
      const bool inbox = (unsigned)bx < (unsigned)f.map.gdim[0] && (unsigned)by < (unsigned)f.map.gdim[1] && (unsigned)bz < (unsigned)f.map.gdim[2];
      const unsigned lin = inbox ? ((unsigned)bz * (unsigned)f.map.gdim[1] + (unsigned)by) * (unsigned)f.map.gdim[0] + (unsigned)bx : 0u;  // < 2^24 blocks
      v4i blk;
      grid_issue(gblocks + 16 * (size_t)lin, blk);
      if (RING && LEAN && PPT > 1 && j == 0) {
        chunk_dma(points, covs, first + kChunkPoints, wbase + kChunkBytes, lane);
        grid_wait<3>(blk);
      } else {
        grid_wait<0>(blk);
      }
      if (j == 0) GP_TRACE(2);
      if (j == 1) GP_TRACE(4);
      const unsigned long long bits = ((unsigned long long)(unsigned)blk.y << 32) | (unsigned long long)(unsigned)blk.x;
      const int pos = ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
      if (inbox && ((bits >> pos) & 1ull)) idx = blk.z + __popcll(bits & ((1ull << pos) - 1ull));
    } else {
      // hop 1: the home line's four keys
      uint32_t l = coord_hash32(cx, cy, cz) & f.map.plmask;
      v4i k0, k1, k2, k3;
      line_issue(lines + 64 * (size_t)l, k0, k1, k2, k3);
      line_wait(k0, k1, k2, k3);
      if (j == 0) GP_TRACE(2);
      if (j == 1) GP_TRACE(4);
      idx = line_match(k0, k1, k2, k3, cx, cy, cz);
      if (live && idx < 0 && k3.w >= 0) {  // full line, no match (rare): walk on
        for (;;) {
          l = (l + 1) & f.map.plmask;
          const GP_GLOBAL v4i* q = (const GP_GLOBAL v4i*)(lines + 64 * (size_t)l);
          const v4i ka = q[0], kb = q[1], kc = q[2], kd = q[3];
          idx = line_match(ka, kb, kc, kd, cx, cy, cz);
          if (idx >= 0 || kd.w < 0) break;
        }
      }
    }
    const bool hit = live && idx >= 0;
    // hop 2: the record (lanes without a voxel read record 0 / the line table: any valid address)
    v4f head;
    v2d c01, c23, c45;
    record_issue(hit ? records + 64 * (size_t)idx : (GRID ? records : lines), head, c01, c23, c45);
    if constexpr (RING) {
      if (j + 2 < PPT) {
        chunk_dma(points, covs, first + (size_t)(j + 2) * kChunkPoints, wbase + ((j + 2) % STAGES) * kChunkBytes, lane);
        record_wait<3>(head, c01, c23, c45);
      } else {
        record_wait<0>(head, c01, c23, c45);
      }
    } else {
      record_wait<0>(head, c01, c23, c45);
    }
    double a[6];
    load_a(a);
    if (hit) {
      // r = mu_B - q with mu_B = voxel centre + mean_local
      rq_t rx = ex + (rq_t)head.x;
      rq_t ry = ey + (rq_t)head.y;
      rq_t rz = ez + (rq_t)head.z;
      if constexpr (MODE == MODE_ERR) {
        // residual at the evaluation pose, correspondence and M at the linearisation pose (vgicp_derivatives.cuh:85-139)
        rx += qx - (Te.r00 * dx + Te.r01 * dy + Te.r02 * dz + Te.tx);
        ry += qy - (Te.r10 * dx + Te.r11 * dy + Te.r12 * dz + Te.ty);
        rz += qz - (Te.r20 * dx + Te.r21 * dy + Te.r22 * dz + Te.tz);
      }
      accumulate_core<MODE, acc_t, rq_t>(Tl, a, c01, c23, c45, rx, ry, rz, qx, qy, qz, acc);
    }
  };

  if constexpr (AHEAD) {
    static_assert(GRID && OUTER_F32 && MODE == MODE_LIN, "the look-ahead pipeline exists for the default linearise kernel");
  }
  struct Ahead {  // what a chunk carries from its front half (transform, hop 1 issued) to its back half (hop 2, algebra)
    v4i blk;
    float ex, ey, ez, qx, qy, qz;
    int pos;  // bit of the voxel inside its block; < 0: outside the grid's box, or rejected by the surface validation
  };
  auto front = [&](int j, Ahead& P) {
    const float* lp = reinterpret_cast<const float*>(wbase + (j % STAGES) * kChunkBytes);
    const double dx = (double)lp[3 * lane], dy = (double)lp[3 * lane + 1], dz = (double)lp[3 * lane + 2];
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    const double ux = lx * f.map.inv_leaf, uy = ly * f.map.inv_leaf, uz = lz * f.map.inv_leaf;
    const double fx = __builtin_floor(ux), fy = __builtin_floor(uy), fz = __builtin_floor(uz);
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    P.ex = (float)(f.map.leaf * ((fx + 0.5) - ux));
    P.ey = (float)(f.map.leaf * ((fy + 0.5) - uy));
    P.ez = (float)(f.map.leaf * ((fz + 0.5) - uz));
    P.qx = (float)lx;
    P.qy = (float)ly;
    P.qz = (float)lz;
    bool live = finite3(lp[3 * lane], lp[3 * lane + 1], lp[3 * lane + 2]);
    if (f.surface_validation && live && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (first + (size_t)j * kChunkPoints + lane))) live = false;
    const int bx = (cx >> 2) - f.map.glo[0], by = (cy >> 2) - f.map.glo[1], bz = (cz >> 2) - f.map.glo[2];
    const bool inbox = (unsigned)bx < (unsigned)f.map.gdim[0] && (unsigned)by < (unsigned)f.map.gdim[1] && (unsigned)bz < (unsigned)f.map.gdim[2];
    const unsigned lin = inbox ? ((unsigned)bz * (unsigned)f.map.gdim[1] + (unsigned)by) * (unsigned)f.map.gdim[0] + (unsigned)bx : 0u;
    P.pos = (inbox && live) ? (((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3)) : -1;
    grid_issue(gblocks + 16 * (size_t)lin, P.blk);
  };
  auto back = [&](int j, Ahead& P) {  // P.blk has landed
    const unsigned long long bits = ((unsigned long long)(unsigned)P.blk.y << 32) | (unsigned long long)(unsigned)P.blk.x;
    const int pos = P.pos < 0 ? 0 : P.pos;
    const bool hit = P.pos >= 0 && ((bits >> pos) & 1ull);
    const int idx = P.blk.z + __popcll(bits & ((1ull << pos) - 1ull));
    v4f head;
    v2d c01, c23, c45;
    record_issue(hit ? records + 64 * (size_t)idx : records, head, c01, c23, c45);
    if (j + 2 < PPT) {
      chunk_dma(points, covs, first + (size_t)(j + 2) * kChunkPoints, wbase + ((j + 2) % STAGES) * kChunkBytes, lane);
      record_wait<3>(head, c01, c23, c45);  // the record -- and hop 1 of chunk j+1, which is older -- are here; chunk j+2 keeps travelling
    } else {
      record_wait<0>(head, c01, c23, c45);
    }
    double a[6];
    load_cov6(reinterpret_cast<const float*>(wbase + (j % STAGES) * kChunkBytes) + kChunkPoints * 3 + 9 * lane, a);
    if (hit) accumulate_core<MODE, acc_t, float>(Tl, a, c01, c23, c45, P.ex + head.x, P.ey + head.y, P.ez + head.z, P.qx, P.qy, P.qz, acc);
  };

  if (ring && AHEAD) {
    if constexpr (AHEAD) {
      Ahead P[2];
      // in flight: chunk 0, chunk 1 (3 requests each)
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // chunk 0 is in LDS
      GP_TRACE(1);
      front(0, P[0]);                                    // in flight: chunk 1, hop 1 of chunk 0
#pragma unroll
      for (int j = 0; j < PPT; j++) {
        if (j + 1 < PPT) {
          // chunk j+1 must be in LDS.  j == 0: it is older than hop 1 of chunk 0, which may keep travelling; j >= 1: the only requests
          // in flight are chunk j+1's (step j-1 waited for its record, and hop 1 of chunk j is older than that)
          if (j == 0) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          front(j + 1, P[(j + 1) & 1]);
          grid_wait<1>(P[j & 1].blk);  // hop 1 of chunk j (older than the one just issued)
        } else {
          grid_wait<0>(P[j & 1].blk);
        }
        if (j == 0) GP_TRACE(2);
        if (j == 1) GP_TRACE(4);
        back(j, P[j & 1]);
        if (j == 0) GP_TRACE(3);
        if (j == 1) GP_TRACE(5);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if (ring) {
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      // only chunk j+1's request may be younger than chunk j (normally already satisfied: step j-1 waited for its gather,
      // which is younger than chunk j -- but a step whose lanes were all rejected waits for nothing)
      if (j + 1 < PPT && !(LEAN && j == 0)) {
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (j == 0) GP_TRACE(1);
      if (j == 1) GP_TRACE(3);
      if (j == 2) GP_TRACE(5);
      const float* lp = reinterpret_cast<const float*>(wbase + (j % STAGES) * kChunkBytes);
      const float* lc = lp + kChunkPoints * 3 + 9 * lane;
      const float px = lp[3 * lane], py = lp[3 * lane + 1], pz = lp[3 * lane + 2];
      step(std::true_type{}, j, true, px, py, pz, [&](double* a) { load_cov6(lc, a); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int j = 0; j < PPT; j++) {
      const int nj = wcount - j * kChunkPoints;  // wave-uniform
      if (nj <= 0) break;
      const bool active = lane < nj;
      const size_t i = first + (size_t)j * kChunkPoints + (active ? lane : 0);
      const GP_GLOBAL float* pp = points + 3 * i;
      step(std::false_type{}, j, active, pp[0], pp[1], pp[2], [&](double* a) { load_cov6(covs + 9 * i, a); });
    }
  }

  GP_TRACE(6);
  // ---- reduction.  The wave's ring (3 x 3 KB, drained) becomes a transposition buffer: 16 components x 64 lanes of f64
  // at a row stride of 68 doubles (conflicts <= 2-way) are written lane-major and read back so that every lane sums 16
  // values of one component, the 4 lanes of a quad are combined with two DPP swaps, and lane 4c holds component c.
  // Two passes (components 0-15, 16-31) ~ 300 issue cycles per wave instead of ~1000 for a 64-lane f64 butterfly.
  // The 4-wave sum goes through the last 256 B of each wave's own region; one 32-double partial per tile. ----
  constexpr int kRowStride = 68;
  static_assert(16 * kRowStride * 8 + 32 * 8 <= STAGES * kChunkBytes, "transposition buffer + wave sums must fit the wave's ring");
  double* wtrans = reinterpret_cast<double*>(wbase);
  double* wsums = reinterpret_cast<double*>(wbase + STAGES * kChunkBytes - 32 * 8);
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = (double)acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) wsums[k] = v;
    }
  } else if constexpr (OUTER_F32) {
    // f32 accumulators (each holds <= PPT points): ONE pass of 32 components x 64 lanes of f32 at a row stride of 66 floats
    // (conflict-free both ways), every lane sums 32 values of one component in f64, lane pairs meet with one DPP swap: half the
    // LDS bytes of the f64 transposition
    constexpr int kRowStrideF = 66;
    static_assert(32 * kRowStrideF * 4 + 32 * 8 <= STAGES * kChunkBytes, "f32 transposition buffer + wave sums must fit the wave's ring");
    float* wtf = reinterpret_cast<float*>(wbase);
#pragma unroll
    for (int k = 0; k < 32; k++) wtf[k * kRowStrideF + lane] = acc[k];
    const int comp = lane >> 1, part = lane & 1;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      s0 += (double)wtf[comp * kRowStrideF + 2 * i + part];
      s1 += (double)wtf[comp * kRowStrideF + 2 * (i + 1) + part];
      s2 += (double)wtf[comp * kRowStrideF + 2 * (i + 2) + part];
      s3 += (double)wtf[comp * kRowStrideF + 2 * (i + 3) + part];
    }
    double v = (s0 + s1) + (s2 + s3);
    v += __shfl_xor(v, 1, 64);
    if (part == 0) wsums[comp] = v;
  } else {
    const int comp = lane >> 2, part = lane & 3;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
      for (int k = 0; k < 16; k++) wtrans[k * kRowStride + lane] = (double)acc[pass * 16 + k];
      // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait between the writes and the reads
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        s0 += wtrans[comp * kRowStride + 4 * i + part];
        s1 += wtrans[comp * kRowStride + 4 * (i + 1) + part];
        s2 += wtrans[comp * kRowStride + 4 * (i + 2) + part];
        s3 += wtrans[comp * kRowStride + 4 * (i + 3) + part];
      }
      double v = (s0 + s1) + (s2 + s3);
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      if (part == 0) wsums[pass * 16 + comp] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double sum = 0.0;
    if (threadIdx.x < NACC) {
      const double* w0 = reinterpret_cast<const double*>(smem + 1 * STAGES * kChunkBytes - 32 * 8);
      const double* w1 = reinterpret_cast<const double*>(smem + 2 * STAGES * kChunkBytes - 32 * 8);
      const double* w2 = reinterpret_cast<const double*>(smem + 3 * STAGES * kChunkBytes - 32 * 8);
      const double* w3 = reinterpret_cast<const double*>(smem + 4 * STAGES * kChunkBytes - 32 * 8);
      sum = (w0[threadIdx.x] + w1[threadIdx.x]) + (w2[threadIdx.x] + w3[threadIdx.x]);
    }
    ((GP_GLOBAL double*)partials)[(size_t)tile.row * ACC_STRIDE + threadIdx.x] = sum;
  }
  GP_TRACE(7);
  if constexpr (TRACE) {
    if (trace && threadIdx.x == 0) trace[(size_t)tile_idx * 16 + 11] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace gp
