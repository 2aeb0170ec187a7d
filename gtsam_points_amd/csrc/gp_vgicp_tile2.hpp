// gp_vgicp_tile2.hpp -- the building blocks of the second / third generation of the rigid-pose tile kernel (12-B LDS-DMA rows, saddr lookups,
// hand-counted waits, the f64-diet algebra).  The second-generation kernel itself (vgicp_pipeline2_kernel: fixed tiles of 1024 / 512 / 256 points,
// the default of late round 2) was superseded by vgicp_stream_kernel (gp_vgicp_stream.hpp) in round 3 and removed after the A/B
// (profiles/r03_sweep_*.jsonl; `git show 07b3b2b:gtsam_points_amd/csrc/gp_vgicp_tile2.hpp` has it).  What the notes below say about the
// schedule holds for the stream kernel, which keeps it.  (Replaces vgicp_derivatives_kernel / vgicp_error_kernel of
// include/gtsam_points/cuda/kernels/vgicp_derivatives.cuh:15-139 together with lookup_voxels.cuh:19-97 and the CUB reduction of
// src/gtsam_points/factors/integrated_vgicp_derivatives_{linearize,compute}.cu.)
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

// the covariance rows of a PACKED chunk (SourceMirror: [64 points | 64 x (c00, c01, c02) | 64 x (c11, c12, c22)], 768 B each): rows 1 and 2 of the
// chunk at `uchunk` into the second and third KB of the chunk's 3 KB slot (M0 = slot + 256 k for the row at instruction offset 768 k, as above)
constexpr int kPackedChunkBytes = 2304;
template <bool NT>
__device__ __forceinline__ void chunk_dma12_cov_packed(const GP_GLOBAL char* uchunk, unsigned voff, char* slot) {
  const unsigned lds_c1 = (unsigned)(size_t)(GP_LDS char*)slot + 256u, lds_c2 = lds_c1 + 256u;
  unsigned saved;
  if constexpr (NT) {
    asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:768 nt\n\t"
      "s_mov_b32 m0, %4\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:1536 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(voff), "s"(uchunk), "s"(lds_c1), "s"(lds_c2)
      : "memory");
  } else {
    asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:768\n\t"
      "s_mov_b32 m0, %4\n\t"
      "global_load_lds_dwordx3 %1, %2 offset:1536\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(voff), "s"(uchunk), "s"(lds_c1), "s"(lds_c2)
      : "memory");
  }
}

// 24-bit multiply-add (the block grid has < 2^24 blocks): hipcc turned __umul24(a, b) + c into a v_mad_u64_u32
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ... with a wave-uniform multiplier straight out of an SGPR (a VOP3 instruction reads one scalar operand; the "v" form cost a v_mov per use)
__device__ __forceinline__ unsigned mad24s(unsigned a, unsigned b_uniform, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
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
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  static_assert(N == 0 || N == 1 || N == 2 || N == 3 || N == 4 || N == 7, "add the count");
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
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(head), "+v"(c01), "+v"(c23), "+v"(c45) : : "memory");
  static_assert(N == 0 || N == 2 || N == 3 || N == 4, "add the count");
}

// M = (C_B + R C_A R^T)^-1 in f64 and the 29 sums in f32: accumulate_core of gp_vgicp_tile.hpp with one Newton step behind the
// hardware reciprocal (v_rcp_f64 is good to ~2^-23, one step gives 2^-46).
// Round 4: the steady state of the stream kernel is VALU-bound (240 vector instructions per 64-point chunk; f64 instructions issue in 4 cycles, f32 in 2 --
// profiles/r01_microbench.txt -- ~750 cycles per chunk and SIMD, 7 of the launch's 10.5 us), and every instantiation must compute the same bits (the packed
// and the unpacked stream, with and without surface validation: tests/test_mirror_gpu.py), so the arithmetic is spelled out instruction by instruction:
// `#pragma clang fp contract(off)` + explicit fused multiply-adds in exactly the places hipcc's contraction had put them.  Two cheaper forms were measured and
// dropped: cofactors x (1 / det) in f32 (-1 % cycles; the record moved 6e-8 against the round-2 kernel), and the sums' products added straight into the
// accumulators (see below).
// ROT32 (measurement only, round 5: the upper bound of what ANY cheaper form of R C_A R^T could buy -- VERDICT r04 #1c): the rotation of the source covariance in
// plain f32.  It breaks the parity contract (3e-5 on H) and is never selected by the product; profiles/r05_kernel_experiments.txt has what it measured.
template <int MODE, bool ROT32 = false>
__device__ __forceinline__ void accumulate_core2(const Pose& Tl, const double* a, const v2d& c01, const v2d& c23, const v2d& c45, float RX, float RY, float RZ, float QX,
                                                 float QY, float QZ, float* acc) {
#pragma clang fp contract(off)
  float M0, M1, M2, M3, M4, M5;
  if constexpr (ROT32) {
    const float r00 = (float)Tl.r00, r01 = (float)Tl.r01, r02 = (float)Tl.r02, r10 = (float)Tl.r10, r11 = (float)Tl.r11, r12 = (float)Tl.r12, r20 = (float)Tl.r20,
                r21 = (float)Tl.r21, r22 = (float)Tl.r22;
    const float a00 = (float)a[0], a01 = (float)a[1], a02 = (float)a[2], a11 = (float)a[3], a12 = (float)a[4], a22 = (float)a[5];
    const auto d3 = [](float x0, float y0, float x1, float y1, float x2, float y2) { return __builtin_fmaf(x2, y2, __builtin_fmaf(x1, y1, x0 * y0)); };
    const float rc00 = d3(r00, a00, r01, a01, r02, a02), rc01 = d3(r00, a01, r01, a11, r02, a12), rc02 = d3(r00, a02, r01, a12, r02, a22);
    const float rc10 = d3(r10, a00, r11, a01, r12, a02), rc11 = d3(r10, a01, r11, a11, r12, a12), rc12 = d3(r10, a02, r11, a12, r12, a22);
    const float rc20 = d3(r20, a00, r21, a01, r22, a02), rc21 = d3(r20, a01, r21, a11, r22, a12), rc22 = d3(r20, a02, r21, a12, r22, a22);
    const double s00 = c01.x + (double)d3(rc00, r00, rc01, r01, rc02, r02), s01 = c01.y + (double)d3(rc00, r10, rc01, r11, rc02, r12);
    const double s02 = c23.x + (double)d3(rc00, r20, rc01, r21, rc02, r22), s11 = c23.y + (double)d3(rc10, r10, rc11, r11, rc12, r12);
    const double s12 = c45.x + (double)d3(rc10, r20, rc11, r21, rc12, r22), s22 = c45.y + (double)d3(rc20, r20, rc21, r21, rc22, r22);
    const double i00 = __builtin_fma(s11, s22, -(s12 * s12)), i01 = __builtin_fma(s02, s12, -(s01 * s22)), i02 = __builtin_fma(s01, s12, -(s02 * s11));
    const double det = __builtin_fma(s02, i02, __builtin_fma(s01, i01, s00 * i00));
    double x = __builtin_amdgcn_rcp(det);
    x = x * __builtin_fma(-det, x, 2.0);
    M0 = (float)(i00 * x);
    M1 = (float)(i01 * x);
    M2 = (float)(i02 * x);
    M3 = (float)(__builtin_fma(s00, s22, -(s02 * s02)) * x);
    M4 = (float)(__builtin_fma(s01, s02, -(s00 * s12)) * x);
    M5 = (float)(__builtin_fma(s00, s11, -(s01 * s01)) * x);
  } else {
    const double a00 = a[0], a01 = a[1], a02 = a[2], a11 = a[3], a12 = a[4], a22 = a[5];
    const auto dot3 = [](double x0, double y0, double x1, double y1, double x2, double y2) { return __builtin_fma(x2, y2, __builtin_fma(x1, y1, x0 * y0)); };
    const auto dot3p = [](double c, double x0, double y0, double x1, double y1, double x2, double y2) { return __builtin_fma(x2, y2, __builtin_fma(x1, y1, __builtin_fma(x0, y0, c))); };
    const double rc00 = dot3(Tl.r00, a00, Tl.r01, a01, Tl.r02, a02), rc01 = dot3(Tl.r00, a01, Tl.r01, a11, Tl.r02, a12), rc02 = dot3(Tl.r00, a02, Tl.r01, a12, Tl.r02, a22);
    const double rc10 = dot3(Tl.r10, a00, Tl.r11, a01, Tl.r12, a02), rc11 = dot3(Tl.r10, a01, Tl.r11, a11, Tl.r12, a12), rc12 = dot3(Tl.r10, a02, Tl.r11, a12, Tl.r12, a22);
    const double rc20 = dot3(Tl.r20, a00, Tl.r21, a01, Tl.r22, a02), rc21 = dot3(Tl.r20, a01, Tl.r21, a11, Tl.r22, a12), rc22 = dot3(Tl.r20, a02, Tl.r21, a12, Tl.r22, a22);
    const double s00 = dot3p(c01.x, rc00, Tl.r00, rc01, Tl.r01, rc02, Tl.r02);
    const double s01 = dot3p(c01.y, rc00, Tl.r10, rc01, Tl.r11, rc02, Tl.r12);
    const double s02 = dot3p(c23.x, rc00, Tl.r20, rc01, Tl.r21, rc02, Tl.r22);
    const double s11 = dot3p(c23.y, rc10, Tl.r10, rc11, Tl.r11, rc12, Tl.r12);
    const double s12 = dot3p(c45.x, rc10, Tl.r20, rc11, Tl.r21, rc12, Tl.r22);
    const double s22 = dot3p(c45.y, rc20, Tl.r20, rc21, Tl.r21, rc22, Tl.r22);
    const double i00 = __builtin_fma(s11, s22, -(s12 * s12)), i01 = __builtin_fma(s02, s12, -(s01 * s22)), i02 = __builtin_fma(s01, s12, -(s02 * s11));
    const double det = __builtin_fma(s02, i02, __builtin_fma(s01, i01, s00 * i00));
    double x = __builtin_amdgcn_rcp(det);
    x = x * __builtin_fma(-det, x, 2.0);
    M0 = (float)(i00 * x);
    M1 = (float)(i01 * x);
    M2 = (float)(i02 * x);
    M3 = (float)(__builtin_fma(s00, s22, -(s02 * s02)) * x);
    M4 = (float)(__builtin_fma(s01, s02, -(s00 * s12)) * x);
    M5 = (float)(__builtin_fma(s00, s11, -(s01 * s01)) * x);
  }
  const auto fm = [](float x, float y, float c) { return __builtin_fmaf(x, y, c); };
  const float mrx = fm(M2, RZ, fm(M1, RY, M0 * RX)), mry = fm(M4, RZ, fm(M3, RY, M1 * RX)), mrz = fm(M5, RZ, fm(M4, RY, M2 * RX));
  acc[ACC_COUNT] += 1.0f;
  acc[ACC_ERR] += fm(RZ, mrz, fm(RY, mry, RX * mrx));
  if constexpr (MODE == MODE_ERR) return;
  acc[ACC_M + 0] += M0;
  acc[ACC_M + 1] += M1;
  acc[ACC_M + 2] += M2;
  acc[ACC_M + 3] += M3;
  acc[ACC_M + 4] += M4;
  acc[ACC_M + 5] += M5;
  // K = M S, S = [q]x
  const float k00 = fm(M1, QZ, -(M2 * QY)), k01 = fm(M2, QX, -(M0 * QZ)), k02 = fm(M0, QY, -(M1 * QX));
  const float k10 = fm(M3, QZ, -(M4 * QY)), k11 = fm(M4, QX, -(M1 * QZ)), k12 = fm(M1, QY, -(M3 * QX));
  const float k20 = fm(M4, QZ, -(M5 * QY)), k21 = fm(M5, QX, -(M2 * QZ)), k22 = fm(M2, QY, -(M4 * QX));
  acc[ACC_K + 0] += k00;
  acc[ACC_K + 1] += k01;
  acc[ACC_K + 2] += k02;
  acc[ACC_K + 3] += k10;
  acc[ACC_K + 4] += k11;
  acc[ACC_K + 5] += k12;
  acc[ACC_K + 6] += k20;
  acc[ACC_K + 7] += k21;
  acc[ACC_K + 8] += k22;
  // TL = -S K (upper triangle), b_t = [q x (M r); M r]: every term is formed first and added once (adding the products straight into the accumulators -- two fused
  // multiply-adds per sum, 13 instructions fewer -- rounds twice at the magnitude of the PRODUCTS, which cancel: measured 2e-8 instead of 6e-9 against the oracle
  // and as much between two partitions of the same cloud.  Not kept.)
  acc[ACC_TL + 0] += fm(QZ, k10, -(QY * k20));
  acc[ACC_TL + 1] += fm(QZ, k11, -(QY * k21));
  acc[ACC_TL + 2] += fm(QZ, k12, -(QY * k22));
  acc[ACC_TL + 3] += fm(QX, k21, -(QZ * k01));
  acc[ACC_TL + 4] += fm(QX, k22, -(QZ * k02));
  acc[ACC_TL + 5] += fm(QY, k02, -(QX * k12));
  acc[ACC_QXMR + 0] += fm(QY, mrz, -(QZ * mry));
  acc[ACC_QXMR + 1] += fm(QZ, mrx, -(QX * mrz));
  acc[ACC_QXMR + 2] += fm(QX, mry, -(QY * mrx));
  acc[ACC_MR + 0] += mrx;
  acc[ACC_MR + 1] += mry;
  acc[ACC_MR + 2] += mrz;
}

}  // namespace gp
