// gp_vgicp_tile2.hpp -- second generation of the rigid-pose tile kernel: linearise and error evaluation over block-grid maps, f32 outer
// products, 1024- / 512- / 256-point tiles (the default kernel since late round 2; replaces vgicp_derivatives_kernel / vgicp_error_kernel of
// include/gtsam_points/cuda/kernels/vgicp_derivatives.cuh:15-139 together with lookup_voxels.cuh:19-97 and the CUB reduction of
// src/gtsam_points/factors/integrated_vgicp_derivatives_{linearize,compute}.cu).
//
// Same pipeline idea as vgicp_pipeline_kernel (gp_vgicp_tile.hpp): LDS-DMA source ring, two-hop lookup with hand-placed waits,
// 29 sums, f32 transposition + f64 reduction.  What changed, and why (ISA of the round-2 default kernel, DESIGN.md section 8:
// a 64-point wave step is ~1000 VALU cycles, 62 % of them f64-rate instructions, ~13 % integer / address arithmetic):
//   * every address of the step is "wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane offset": the source ring is requested
//     with FOUR 12-B-per-lane DMA instructions per chunk (one for the 64 points, three for the 64 covariances -- each instruction
//     reads one array, so no per-lane base select), the block entry and the record with the saddr form of global_load.  The
//     round-2 kernel spent 7 v_mad_u64_u32 + 7 v_lshl_add_u64 + 3 v_lshlrev_b64 per step on 64-bit per-lane addresses; the
//     per-chunk advance of the bases is now scalar arithmetic;
//   * 12-B DMA rows need no 16-B alignment of the caller's arrays: every full wave takes the ring; LDS per workgroup 34 KB (was 36);
//   * f64 diet: transform as three 3-deep fma chains (9 instead of 12 instructions), centre - l = fma(-leaf, fract(u), leaf/2)
//     (6 instead of 9), one Newton step behind v_rcp_f64 (2^-46 relative: eight orders inside what the f32 outer products keep);
//   * block index with 24-bit multiply-adds (the grid has < 2^24 blocks);
//   * schedule: lean, points-first start -- the first burst of a wave is the 768 B of chunk 0's points, so the first transform and hop 1
//     do not queue behind everybody's covariances; those follow hop 1, the points of chunk 1 go out before hop 2 and its covariances
//     behind it -- and the front half of chunk j+1 (transform, hop 1 issued) sits behind the record wait of chunk j, so that its hop 1
//     travels under the algebra of chunk j.  The alternatives (the round-2 look-ahead order with both chunks requested up front; the
//     lean start without points first) were measured with this kernel and removed: profiles/r02_gen2_ab.txt;
//   * reduction: four f32 in-lane partial sums per component instead of sixteen cvt + f64 adds (a third of its issue cycles);
//   * NT: the non-temporal policy on the source stream, chosen per batch by the host: +3.5 % on C2 / +4.5 % on the C4 shard, where
//     every source cloud is read by one factor of the launch, -29 % on C3, where four factors share a cloud and the re-reads would
//     have hit L2.
// Arithmetic differs from variants 4 / 8 at the 1e-16 level (fma contraction, fract), not bit for bit; parity tests are the same.
#pragma once

#include "gp_vgicp_tile.hpp"

namespace gp {

// a wave-uniform pointer the compiler may have left in vector registers (a descriptor field selected between the kernel arguments
// and a table in memory): pin it to an SGPR pair so that the saddr addressing forms can take it
template <typename T>
__device__ __forceinline__ const GP_GLOBAL T* uniform_ptr(const GP_GLOBAL T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const GP_GLOBAL T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uniform_f64(double x) {
  const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// LDS layout of global_load_lds_dwordx3, measured on gfx950 (scripts/probe/lds_dma_layout.hip): lane L's 12 bytes land at
// M0 + instruction offset + 16 * L -- a SIXTEEN-byte lane stride with a 4-byte hole, not lane x 12 -- and the instruction offset
// moves the global and the LDS address alike.  One instruction therefore fills 64 slots of 16 B (1 KB of LDS for 768 B of data):
//   points       64 slots: slot p = point p                       -> one ds_read_b96 per lane, conflict-free
//   covariances  3 x 64 slots: slot q = 12-B piece q of the 2304 contiguous bytes, point p = slots 3p, 3p+1, 3p+2 (its three
//                columns) -> three ds_read_b96 at a 48-B lane stride (12 dwords: conflict-free within the 8-lane groups of a b96 read)
// A wave's ring: 2 point arrays (1 KB each) + 2 covariance arrays (3 KB each) = 8 KB; the reduction needs 8.5 KB.
constexpr int kPtsSlotBytes = 1024, kCovSlotBytes = 3072, kWaveLdsBytes = 8704;
static_assert(2 * kPtsSlotBytes + 2 * kCovSlotBytes <= kWaveLdsBytes, "ring must fit the wave's LDS region");

// one 64-point chunk: `upts` / `ucov` are wave-uniform (the chunk's first row, in SGPR pairs), `voff` = lane * 12.  Issued from inline
// asm: (i) the saddr form is guaranteed (the builtin fell back to 64-bit per-lane addresses for every chunk but the first), (ii)
// hipcc does not track these requests, so it cannot put a vmcnt(0) of its own in front of the first LDS read (it did, behind the
// builtin).  Covariance instruction k reads global bytes [768 k, 768 (k+1)) into slots [64 k, 64 (k+1)): its M0 is the array base +
// 256 k, because the instruction offset 768 k is added to the LDS address as well.  M0 is saved and restored.
// NT: the non-temporal policy on the source stream (it is read once per launch; the block grid and the records are what should stay in L2)
template <bool NT>
__device__ __forceinline__ void chunk_dma12_pts(const GP_GLOBAL char* upts, unsigned voff, char* pslot) {
  const unsigned lds_pts = (unsigned)(size_t)(GP_LDS char*)pslot;
  unsigned saved;
  if constexpr (NT) {
    asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "global_load_lds_dwordx3 %1, %2 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(voff), "s"(upts), "s"(lds_pts)
      : "memory");
  } else {
    asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "global_load_lds_dwordx3 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(voff), "s"(upts), "s"(lds_pts)
      : "memory");
  }
}
template <bool NT>
__device__ __forceinline__ void chunk_dma12_cov(const GP_GLOBAL char* ucov, unsigned voff, char* cslot) {
  const unsigned lds_c0 = (unsigned)(size_t)(GP_LDS char*)cslot, lds_c1 = lds_c0 + 256u, lds_c2 = lds_c0 + 512u;
  unsigned saved;
  if constexpr (NT) {
    asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "global_load_lds_dwordx3 %1, %2 nt\n\t"
      "s_mov_b32 m0, %4\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:768 nt\n\t"
      "s_mov_b32 m0, %5\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:1536 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(voff), "s"(ucov), "s"(lds_c0), "s"(lds_c1), "s"(lds_c2)
      : "memory");
  } else {
    asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "global_load_lds_dwordx3 %1, %2\n\t"
      "s_mov_b32 m0, %4\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:768\n\t"
      "s_mov_b32 m0, %5\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:1536\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(voff), "s"(ucov), "s"(lds_c0), "s"(lds_c1), "s"(lds_c2)
      : "memory");
  }
}

// 24-bit multiply-add (the block grid has < 2^24 blocks): hipcc turned __umul24(a, b) + c into a v_mad_u64_u32
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// saddr-form lookups: wave-uniform base in an SGPR pair, 32-bit byte offset per lane
__device__ __forceinline__ void grid_issue_s(const GP_GLOBAL char* base, unsigned off, v4i& blk) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(blk) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void record_issue_s(const GP_GLOBAL char* base, unsigned off, v4f& head, v2d& c01, v2d& c23, v2d& c45) {
  asm volatile(
    "global_load_dwordx4 %0, %4, %5\n\t"
    "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
    "global_load_dwordx4 %2, %4, %5 offset:32\n\t"
    "global_load_dwordx4 %3, %4, %5 offset:48"
    : "=&v"(head), "=&v"(c01), "=&v"(c23), "=&v"(c45)
    : "v"(off), "s"(base)
    : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
  static_assert(N >= 0 && N <= 8, "counts used by the schedules below");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  static_assert(N == 0 || N == 1 || N == 3 || N == 4 || N == 7, "add the count");
}
template <int N>
__device__ __forceinline__ void vm_wait_blk(v4i& blk) {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(blk) : : "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(blk) : : "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(blk) : : "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(blk) : : "memory");
  static_assert(N == 0 || N == 1 || N == 3 || N == 4, "add the count");
}
template <int N>
__device__ __forceinline__ void vm_wait_rec(v4f& head, v2d& c01, v2d& c23, v2d& c45) {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  static_assert(N == 0 || N == 3 || N == 4, "add the count");
}

// M = (C_B + R C_A R^T)^-1 in f64 and the 29 sums in f32: accumulate_core of gp_vgicp_tile.hpp with one Newton step behind the
// hardware reciprocal (v_rcp_f64 is good to ~2^-23, one step gives 2^-46)
template <int MODE>
__device__ __forceinline__ void accumulate_core2(const Pose& Tl, const double* a, const v2d& c01, const v2d& c23, const v2d& c45, float RX, float RY, float RZ, float QX,
                                                 float QY, float QZ, float* acc) {
  float M0, M1, M2, M3, M4, M5;
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
    const double det = s00 * i00 + s01 * i01 + s02 * i02;
    double x = __builtin_amdgcn_rcp(det);
    x = x * (2.0 - det * x);
    M0 = (float)(i00 * x);
    M1 = (float)(i01 * x);
    M2 = (float)(i02 * x);
    M3 = (float)((s00 * s22 - s02 * s02) * x);
    M4 = (float)((s01 * s02 - s00 * s12) * x);
    M5 = (float)((s00 * s11 - s01 * s01) * x);
  }
  const float mrx = M0 * RX + M1 * RY + M2 * RZ, mry = M1 * RX + M3 * RY + M4 * RZ, mrz = M2 * RX + M4 * RY + M5 * RZ;
  acc[ACC_COUNT] += 1.0f;
  acc[ACC_ERR] += RX * mrx + RY * mry + RZ * mrz;
  if constexpr (MODE == MODE_ERR) return;
  acc[ACC_M + 0] += M0;
  acc[ACC_M + 1] += M1;
  acc[ACC_M + 2] += M2;
  acc[ACC_M + 3] += M3;
  acc[ACC_M + 4] += M4;
  acc[ACC_M + 5] += M5;
  const float k00 = M1 * QZ - M2 * QY, k01 = M2 * QX - M0 * QZ, k02 = M0 * QY - M1 * QX;
  const float k10 = M3 * QZ - M4 * QY, k11 = M4 * QX - M1 * QZ, k12 = M1 * QY - M3 * QX;
  const float k20 = M4 * QZ - M5 * QY, k21 = M5 * QX - M2 * QZ, k22 = M2 * QY - M4 * QX;
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

// INL: a single-factor launch; the factor descriptor, the pose and the tile geometry come out of the kernel arguments through scalar
// loads.  (With a run-time `inl.use ? inl.factor : factors[...]` hipcc selects between the two ADDRESSES and reads the descriptor
// with flat loads: a vector-memory round trip in front of the first source request, also for the in-argument copy.)
// MODE_ERR (vgicp_error_kernel, vgicp_derivatives.cuh:85-139): correspondence and M at the linearisation pose, residual at the evaluation
// pose: r = mu_B - T_e p = (centre - l) + mean_local + (l - l_e), the last term formed in f64 in the front half; 2 sums.
template <int MODE, int PPT, bool NT, bool INL, bool TRACE = false>
__global__ void __launch_bounds__(256, 4) vgicp_pipeline2_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                                  const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                                  double* __restrict__ partials) {
  static_assert(PPT == 1 || PPT == 2 || PPT == 4, "256-, 512- and 1024-point tiles");
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "rigid linearise and error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : 32;
  __shared__ __attribute__((aligned(16))) char smem[4 * kWaveLdsBytes];  // 34 KB
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
    if (trace && threadIdx.x == 0) {
      trace[(size_t)tile_idx * 16 + 10] = __builtin_amdgcn_s_memrealtime();
      trace[(size_t)tile_idx * 16 + 8] = __builtin_amdgcn_s_getreg(GP_GETREG_HW_ID);
      trace[(size_t)tile_idx * 16 + 9] = __builtin_amdgcn_s_getreg(GP_GETREG_XCC_ID);
    }
  }
  TileDesc tile;
  FactorDesc f;
  if constexpr (INL) {
    tile.factor = 0;
    tile.begin = tile_idx * inl.tile_points;
    tile.count = min(inl.tile_points, inl.factor.n - tile.begin);
    tile.row = tile_idx;
    f = inl.factor;
  } else {
    tile = tiles[tile_idx];
    f = factors[tile.factor];
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t first = (size_t)tile.begin + (size_t)wave * (PPT * kChunkPoints);  // the wave's first point (wave-uniform)
  int wcount = __builtin_amdgcn_readfirstlane(tile.count) - wave * (PPT * kChunkPoints);
  wcount = wcount < 0 ? 0 : (wcount > PPT * kChunkPoints ? PPT * kChunkPoints : wcount);
  const bool ring = wcount == PPT * kChunkPoints;
  char* wbase = smem + wave * kWaveLdsBytes;
  auto pslot = [&](int j) { return wbase + (j & 1) * kPtsSlotBytes; };
  auto cslot = [&](int j) { return wbase + 2 * kPtsSlotBytes + (j & 1) * kCovSlotBytes; };
  const GP_GLOBAL char* upts = uniform_ptr((const GP_GLOBAL char*)as_global(f.points) + 12 * first);
  const GP_GLOBAL char* ucov = uniform_ptr((const GP_GLOBAL char*)as_global(f.covs) + 36 * first);
  const unsigned voff = (unsigned)lane * 12u;
  auto dma_pts = [&](int j) { chunk_dma12_pts<NT>(upts + (size_t)j * (kChunkPoints * 12), voff, pslot(j)); };
  auto dma_cov = [&](int j) { chunk_dma12_cov<NT>(ucov + (size_t)j * (kChunkPoints * 36), voff, cslot(j)); };
  auto dma = [&](int j) {
    dma_pts(j);
    dma_cov(j);
  };

  if (ring) dma_pts(0);

  const Pose Tl = INL ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (INL ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;
  const double leaf = uniform_f64(f.map.leaf), inv_leaf = uniform_f64(f.map.inv_leaf), half_leaf = uniform_f64(0.5 * f.map.leaf);
  const int glo0 = f.map.glo[0], glo1 = f.map.glo[1], glo2 = f.map.glo[2];
  const unsigned gd0 = (unsigned)f.map.gdim[0], gd1 = (unsigned)f.map.gdim[1], gd2 = (unsigned)f.map.gdim[2];
  const GP_GLOBAL char* gblocks = uniform_ptr((const GP_GLOBAL char*)f.map.gblocks);
  const GP_GLOBAL char* records = uniform_ptr((const GP_GLOBAL char*)f.map.records);

  // the translation lives in vector registers: a VOP3 instruction reads ONE scalar operand, so fma(r02, dz, tx) with both in SGPRs
  // costs a v_mov_b64 per row per chunk; the kernel has the six registers to spare (108 of 128 without them)
  double tvx, tvy, tvz;
  asm volatile("v_mov_b64 %0, %1" : "=v"(tvx) : "s"(Tl.tx));
  asm volatile("v_mov_b64 %0, %1" : "=v"(tvy) : "s"(Tl.ty));
  asm volatile("v_mov_b64 %0, %1" : "=v"(tvz) : "s"(Tl.tz));

  float acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.f;

  struct Ahead {  // what a chunk carries from its front half (transform, hop 1 issued) to its back half (hop 2, algebra)
    v4i blk;
    float ex, ey, ez, qx, qy, qz;
    int pos;  // bit of the voxel inside its block; < 0: outside the grid's box or an inactive lane
  };
  // front half: transform, voxel coordinate, hop 1 issued
  auto front = [&](float pxf, float pyf, float pzf, bool active, Ahead& P) {
    const double dx = (double)pxf, dy = (double)pyf, dz = (double)pzf;
    const double lx = __builtin_fma(Tl.r00, dx, __builtin_fma(Tl.r01, dy, __builtin_fma(Tl.r02, dz, tvx)));
    const double ly = __builtin_fma(Tl.r10, dx, __builtin_fma(Tl.r11, dy, __builtin_fma(Tl.r12, dz, tvy)));
    const double lz = __builtin_fma(Tl.r20, dx, __builtin_fma(Tl.r21, dy, __builtin_fma(Tl.r22, dz, tvz)));
    // voxel coordinate = floor(l * (1 / leaf)): the CPU map's rule (util/fast_floor.hpp:12-15, gaussian_voxelmap_cpu.cpp:59-61);
    // centre - l = leaf (floor(u) + 0.5 - u) = leaf/2 - leaf fract(u): the large coordinates never meet
    const double ux = lx * inv_leaf, uy = ly * inv_leaf, uz = lz * inv_leaf;
    const int cx = (int)__builtin_floor(ux), cy = (int)__builtin_floor(uy), cz = (int)__builtin_floor(uz);
    if constexpr (MODE == MODE_ERR) {
      const double ex_ = Te.r00 * dx + Te.r01 * dy + Te.r02 * dz + Te.tx, ey_ = Te.r10 * dx + Te.r11 * dy + Te.r12 * dz + Te.ty, ez_ = Te.r20 * dx + Te.r21 * dy + Te.r22 * dz + Te.tz;
      P.ex = (float)(__builtin_fma(-leaf, __builtin_amdgcn_fract(ux), half_leaf) + (lx - ex_));
      P.ey = (float)(__builtin_fma(-leaf, __builtin_amdgcn_fract(uy), half_leaf) + (ly - ey_));
      P.ez = (float)(__builtin_fma(-leaf, __builtin_amdgcn_fract(uz), half_leaf) + (lz - ez_));
      P.qx = P.qy = P.qz = 0.f;
    } else {
      P.ex = (float)__builtin_fma(-leaf, __builtin_amdgcn_fract(ux), half_leaf);
      P.ey = (float)__builtin_fma(-leaf, __builtin_amdgcn_fract(uy), half_leaf);
      P.ez = (float)__builtin_fma(-leaf, __builtin_amdgcn_fract(uz), half_leaf);
      P.qx = (float)lx;
      P.qy = (float)ly;
      P.qz = (float)lz;
    }
    const bool live = active && finite3(pxf, pyf, pzf);  // (factors with surface validation stay on the round-2 kernel: its normals read is a compiler-tracked load)
    const unsigned bx = (unsigned)((cx >> 2) - glo0), by = (unsigned)((cy >> 2) - glo1), bz = (unsigned)((cz >> 2) - glo2);
    const bool inbox = (bx < gd0) & (by < gd1) & (bz < gd2);
    const unsigned lin = inbox ? mad24(mad24(bz, gd1, by), gd0, bx) : 0u;  // < 2^24 blocks
    P.pos = (inbox && live) ? (((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3)) : -1;
    grid_issue_s(gblocks, lin * 16u, P.blk);
  };
  auto front_ring = [&](int j, Ahead& P) {
    const v3f pt = *reinterpret_cast<const v3f*>(pslot(j) + 16 * lane);
    front(pt.x, pt.y, pt.z, true, P);
  };
  // back half, part 1: P.blk has landed -> record requested
  auto back_issue = [&](const Ahead& P, v4f& head, v2d& c01, v2d& c23, v2d& c45) -> bool {
    const unsigned long long bits = ((unsigned long long)(unsigned)P.blk.y << 32) | (unsigned long long)(unsigned)P.blk.x;
    const int pos = P.pos < 0 ? 0 : P.pos;
    const bool hit = P.pos >= 0 && ((bits >> pos) & 1ull);
    const int idx = P.blk.z + __popcll(bits & ((1ull << pos) - 1ull));
    record_issue_s(records, hit ? (unsigned)idx << 6 : 0u, head, c01, c23, c45);
    return hit;
  };
  // the covariance of this lane's point out of the ring (three 12-B columns), symmetrised like load_cov6 does
  auto cov_ring = [&](int j, double* a) {
    const char* c = cslot(j) + 48 * lane;
    const v3f c0 = *reinterpret_cast<const v3f*>(c), c1 = *reinterpret_cast<const v3f*>(c + 16), c2 = *reinterpret_cast<const v3f*>(c + 32);
    const float c9[9] = {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z};
    load_cov6(c9, a);
  };
  auto algebra = [&](const double* a, const Ahead& P, bool hit, const v4f& head, const v2d& c01, const v2d& c23, const v2d& c45) {
    if (hit) accumulate_core2<MODE>(Tl, a, c01, c23, c45, P.ex + head.x, P.ey + head.y, P.ez + head.z, P.qx, P.qy, P.qz, acc);
  };

  if (ring) {
    Ahead P[2];
    v4f head;
    v2d c01, c23, c45;
    double a[6];
    {
      // points first: only the 768 B of chunk 0's points are in flight, so the first transform and hop 1 do not queue behind everybody's
      // covariances; those follow hop 1 (they are needed behind hop 2), the points of chunk 1 go out before hop 2 and its covariances
      // behind it, so that the wait for the first record does not drag a source request that was issued a moment ago
      vm_wait<0>();
      GP_TRACE(1);
      front_ring(0, P[0]);  // in flight: H0
      dma_cov(0);
      if (PPT > 1) dma_pts(1);  // in flight: H0, C0 x3, P1
#pragma unroll
      for (int j = 0; j < PPT; j++) {
        if (j == 0 && PPT > 1) vm_wait_blk<4>(P[0].blk);    // [H0, C0 x3, P1]
        else if (j == 0) vm_wait_blk<3>(P[0].blk);            // [H0, C0 x3] (one chunk per wave)
        else if (j + 1 < PPT) vm_wait_blk<4>(P[j & 1].blk);  // [H(j), chunk j+1 x4]
        else vm_wait_blk<0>(P[j & 1].blk);
        if (j == 0) GP_TRACE(2);
        if (j == 1) GP_TRACE(4);
        const bool hit = back_issue(P[j & 1], head, c01, c23, c45);
        if (j == 0 && PPT > 1) {
          dma_cov(1);                           // [C0 x3, P1, R0 x4, C1 x3]
          vm_wait_rec<3>(head, c01, c23, c45);  // the record, the covariances of chunk 0 and the points of chunk 1
        } else {
          vm_wait_rec<0>(head, c01, c23, c45);  // the record, and chunk j+1 (requested a step ago), which the front half below reads
        }
        if (j + 1 < PPT) front_ring(j + 1, P[(j + 1) & 1]);  // its hop 1 travels under the algebra of chunk j
        cov_ring(j, a);
        if (j + 2 < PPT) {  // chunk j+2 takes the places of chunk j, whose points and covariance have just been read
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma(j + 2);
        }
        algebra(a, P[j & 1], hit, head, c01, c23, c45);
        if (j == 0) GP_TRACE(3);
        if (j == 1) GP_TRACE(5);
      }
    }
    vm_wait<0>();
  } else {
    // a partial wave (last tile of a factor): per-lane loads, same arithmetic in the same order.  The loads are issued from inline asm
    // like everything else here: a load hipcc tracks itself makes it guard registers of the ring path with vmcnt(0) waits of its own
    // (its dataflow sees a path from this loop into the ring code), which would drain the source requests in flight there.
    const GP_GLOBAL float* points = as_global(f.points);
    const GP_GLOBAL float* covs = as_global(f.covs);
    for (int j = 0; j < PPT; j++) {
      const int nj = wcount - j * kChunkPoints;  // wave-uniform
      if (nj <= 0) break;
      const bool active = lane < nj;
      const size_t i = first + (size_t)j * kChunkPoints + (active ? lane : 0);
      v3f pt;
      v4f ca, cb;
      float cc;
      asm volatile(
        "global_load_dwordx3 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %5, off\n\t"
        "global_load_dwordx4 %2, %5, off offset:16\n\t"
        "global_load_dword %3, %5, off offset:32\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(pt), "=&v"(ca), "=&v"(cb), "=&v"(cc)
        : "v"(points + 3 * i), "v"(covs + 9 * i)
        : "memory");
      const float c9[9] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w, cc};
      Ahead P;
      v4f head;
      v2d c01, c23, c45;
      front(pt.x, pt.y, pt.z, active, P);
      vm_wait_blk<0>(P.blk);
      const bool hit = back_issue(P, head, c01, c23, c45);
      vm_wait_rec<0>(head, c01, c23, c45);
      double a[6];
      load_cov6(c9, a);
      if (hit) accumulate_core2<MODE>(Tl, a, c01, c23, c45, P.ex + head.x, P.ey + head.y, P.ez + head.z, P.qx, P.qy, P.qz, acc);
    }
  }

  GP_TRACE(6);
  // ---- reduction: the wave's drained ring becomes a 32 x 64 f32 transposition buffer (row stride 66 floats: conflict-free both
  // ways), every lane sums 32 values of one component in f64, lane pairs meet with one swap; the 4-wave sum goes through the
  // last 256 B of each wave's region; one 32-double partial per tile (fixed order: bit-reproducible) ----
  constexpr int kRowStrideF = 66;
  static_assert(32 * kRowStrideF * 4 + 32 * 8 <= kWaveLdsBytes, "f32 transposition buffer + wave sums must fit the wave's LDS region");
  float* wtf = reinterpret_cast<float*>(wbase);
  double* wsums = reinterpret_cast<double*>(wbase + kWaveLdsBytes - 32 * 8);
  if constexpr (MODE == MODE_ERR) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double v = (double)acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) wsums[k] = v;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 32; k++) wtf[k * kRowStrideF + lane] = acc[k];
    const int comp = lane >> 1, part = lane & 1;
    // four f32 partial sums of 4 values each (every value is itself the sum of <= PPT points), met in f64: a third of the issue
    // cycles of sixteen cvt + f64 adds; the rounding it adds (2^-24 relative per wave partial, random sign) averages out over the tiles
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      s0 += wtf[comp * kRowStrideF + 2 * i + part];
      s1 += wtf[comp * kRowStrideF + 2 * (i + 1) + part];
      s2 += wtf[comp * kRowStrideF + 2 * (i + 2) + part];
      s3 += wtf[comp * kRowStrideF + 2 * (i + 3) + part];
    }
    double v = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
    v += __shfl_xor(v, 1, 64);
    if (part == 0) wsums[comp] = v;
  }
  __syncthreads();
  if (threadIdx.x < ACC_STRIDE) {
    double sum = 0.0;
    if (threadIdx.x < (MODE == MODE_ERR ? 2 : ACC_SIZE)) {
      const double* w0 = reinterpret_cast<const double*>(smem + 1 * kWaveLdsBytes - 32 * 8);
      const double* w1 = reinterpret_cast<const double*>(smem + 2 * kWaveLdsBytes - 32 * 8);
      const double* w2 = reinterpret_cast<const double*>(smem + 3 * kWaveLdsBytes - 32 * 8);
      const double* w3 = reinterpret_cast<const double*>(smem + 4 * kWaveLdsBytes - 32 * 8);
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
