// gp_vgicp_tile.hpp -- the tuned tile kernel of the VGICP path (rigid poses; MODE_LIN / MODE_ERR).
//
// Same arithmetic as accumulate_point<> in gp_vgicp.hip (the reference-shaped kernel), restructured around what the
// micro-benchmarks say about gfx950 (scripts/stream_bench.py, scripts/alu_rate.py):
//   * the 48 B/point source stream, the gather of L2-resident voxel records and the f64 algebra each cost 8-10 us per
//     million points and ADD UP when a wave does them one after the other: a wave that is hashing or multiplying has no
//     source bytes in flight, and 16 waves per CU keep HBM busy only while all of them wait on it;
//   * so the source travels by LDS-DMA (no VGPRs while in flight) in a rolling 3-stage ring, requested two steps ahead;
//   * the voxel lookup is two dependent round trips with no data-dependent loop in the common case (line table);
//   * all waits are hand-placed (vmcnt retires in order; the compiler cannot count across the DMA requests).
//   All pointers are cast to the global address space (descriptors loaded from memory would otherwise make hipcc emit
//   flat_load, which also ties up lgkmcnt).  The 64-lane reduction is a transposing butterfly: 32 xor-shuffles of doubles
//   instead of 29 x 6.
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

// ---- source-frame formulation ------------------------------------------------------------------------------------------
// The fused covariance C_B + R C_A R^T costs 45 f64 FMAs per POINT in the target frame.  Rotated into the source frame it is
// R^T C_B R + C_A: the rotated term depends on the voxel and the pose only, so a pre-pass computes it once per VOXEL
// (pose_records_kernel: mu' = R^T (mu_B - t), C' = R^T C_B R, 10 doubles per voxel), and a point pays 6 additions.  With
// r' = mu' - p and q' = p + R^T t every sum of the target-side system is the rotation of the same sum formed from the primed
// quantities (M = R M' R^T, K = R K' R^T, -S K = R (-S' K') R^T, q x Mr = R (q' x M'r'), Mr = R M'r'; r^T M r and the count are
// invariant), which the finalize kernel applies once per factor.  ~140 instead of ~200 f64-rate instructions per point.
constexpr int kPosedDoubles = 10;  // mu'(3), C'(xx xy xz yy yz zz), pad: 80 B = five 16-B loads

struct SrcFrame {
  double rtx, rty, rtz;                                  // R_l^T t_l                       (MODE_LIN: q' = p + R^T t)
  double d00, d01, d02, d10, d11, d12, d20, d21, d22;    // D = R_l^T R_e                   (MODE_ERR: r' = mu' - (D p + d))
  double dx, dy, dz;                                     // d = R_l^T (t_e - t_l)
};

__device__ __forceinline__ SrcFrame make_src_frame(const Pose& Tl, const Pose& Te) {
  SrcFrame s;
  s.rtx = Tl.r00 * Tl.tx + Tl.r10 * Tl.ty + Tl.r20 * Tl.tz;
  s.rty = Tl.r01 * Tl.tx + Tl.r11 * Tl.ty + Tl.r21 * Tl.tz;
  s.rtz = Tl.r02 * Tl.tx + Tl.r12 * Tl.ty + Tl.r22 * Tl.tz;
  s.d00 = Tl.r00 * Te.r00 + Tl.r10 * Te.r10 + Tl.r20 * Te.r20;
  s.d01 = Tl.r00 * Te.r01 + Tl.r10 * Te.r11 + Tl.r20 * Te.r21;
  s.d02 = Tl.r00 * Te.r02 + Tl.r10 * Te.r12 + Tl.r20 * Te.r22;
  s.d10 = Tl.r01 * Te.r00 + Tl.r11 * Te.r10 + Tl.r21 * Te.r20;
  s.d11 = Tl.r01 * Te.r01 + Tl.r11 * Te.r11 + Tl.r21 * Te.r21;
  s.d12 = Tl.r01 * Te.r02 + Tl.r11 * Te.r12 + Tl.r21 * Te.r22;
  s.d20 = Tl.r02 * Te.r00 + Tl.r12 * Te.r10 + Tl.r22 * Te.r20;
  s.d21 = Tl.r02 * Te.r01 + Tl.r12 * Te.r11 + Tl.r22 * Te.r21;
  s.d22 = Tl.r02 * Te.r02 + Tl.r12 * Te.r12 + Tl.r22 * Te.r22;
  const double ex = Te.tx - Tl.tx, ey = Te.ty - Tl.ty, ez = Te.tz - Tl.tz;
  s.dx = Tl.r00 * ex + Tl.r10 * ey + Tl.r20 * ez;
  s.dy = Tl.r01 * ex + Tl.r11 * ey + Tl.r21 * ez;
  s.dz = Tl.r02 * ex + Tl.r12 * ey + Tl.r22 * ez;
  return s;
}

// per-correspondence algebra in the source frame; rec = {mu'x mu'y | mu'z C'xx | C'xy C'xz | C'yy C'yz | C'zz pad}
template <int MODE, typename acc_t>
__device__ __forceinline__ void accumulate_terms_src(const SrcFrame& sf, float pxf, float pyf, float pzf, const float* cA, const v2d& r0, const v2d& r1, const v2d& r2,
                                                     const v2d& r3, const v2d& r4, acc_t* acc) {
  const double px = (double)pxf, py = (double)pyf, pz = (double)pzf;
  double m[6];
  {
    const double s00 = r1.y + (double)cA[0], s01 = r2.x + (double)cA[1], s02 = r2.y + (double)cA[2];
    const double s11 = r3.x + (double)cA[3], s12 = r3.y + (double)cA[4], s22 = r4.x + (double)cA[5];
    const double i00 = s11 * s22 - s12 * s12, i01 = s02 * s12 - s01 * s22, i02 = s01 * s12 - s02 * s11;
    const double invdet = fast_rcp(s00 * i00 + s01 * i01 + s02 * i02);
    m[0] = i00 * invdet;
    m[1] = i01 * invdet;
    m[2] = i02 * invdet;
    m[3] = (s00 * s22 - s02 * s02) * invdet;
    m[4] = (s01 * s02 - s00 * s12) * invdet;
    m[5] = (s00 * s11 - s01 * s01) * invdet;
  }
  double rxd, ryd, rzd, qx, qy, qz;
  if constexpr (MODE == MODE_ERR) {
    rxd = r0.x - (sf.d00 * px + sf.d01 * py + sf.d02 * pz + sf.dx);
    ryd = r0.y - (sf.d10 * px + sf.d11 * py + sf.d12 * pz + sf.dy);
    rzd = r1.x - (sf.d20 * px + sf.d21 * py + sf.d22 * pz + sf.dz);
    qx = qy = qz = 0.0;
  } else {
    rxd = r0.x - px;
    ryd = r0.y - py;
    rzd = r1.x - pz;
    qx = px + sf.rtx;
    qy = py + sf.rty;
    qz = pz + sf.rtz;
  }
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

// the pre-pass: one thread per (factor, voxel); tiles[] = {factor, first voxel, count} in chunks of 256 voxels
template <int UNUSED = 0>  // a template only so that the header can be included by several translation units
__global__ void __launch_bounds__(256) pose_records_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ vtiles, const double* __restrict__ poses_lin,
                                                           const InlinePoses inl) {
  TileDesc tile;
  if (inl.use) {
    tile.factor = 0;
    tile.begin = blockIdx.x * 256;
    tile.count = min(256, inl.factor.map.num_voxels - tile.begin);
  } else {
    tile = vtiles[blockIdx.x];
  }
  if ((int)threadIdx.x >= tile.count) return;
  const FactorDesc f = inl.use ? inl.factor : factors[tile.factor];
  const Pose T = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const int v = tile.begin + threadIdx.x;
  const VoxelRecord rec = f.map.records[v];
  const int* c = f.map.voxel_coords + 3 * (size_t)v;
  const double mx = ((double)c[0] + 0.5) * f.map.leaf + (double)rec.mean_local[0] - T.tx;
  const double my = ((double)c[1] + 0.5) * f.map.leaf + (double)rec.mean_local[1] - T.ty;
  const double mz = ((double)c[2] + 0.5) * f.map.leaf + (double)rec.mean_local[2] - T.tz;
  double* out = f.posed + kPosedDoubles * (size_t)v;
  out[0] = T.r00 * mx + T.r10 * my + T.r20 * mz;
  out[1] = T.r01 * mx + T.r11 * my + T.r21 * mz;
  out[2] = T.r02 * mx + T.r12 * my + T.r22 * mz;
  // C' = R^T C R with C symmetric (xx xy xz yy yz zz)
  const double c00 = rec.cov[0], c01 = rec.cov[1], c02 = rec.cov[2], c11 = rec.cov[3], c12 = rec.cov[4], c22 = rec.cov[5];
  const double R[3][3] = {{T.r00, T.r01, T.r02}, {T.r10, T.r11, T.r12}, {T.r20, T.r21, T.r22}};
  double CR[3][3];  // C R
#pragma unroll
  for (int j = 0; j < 3; j++) {
    CR[0][j] = c00 * R[0][j] + c01 * R[1][j] + c02 * R[2][j];
    CR[1][j] = c01 * R[0][j] + c11 * R[1][j] + c12 * R[2][j];
    CR[2][j] = c02 * R[0][j] + c12 * R[1][j] + c22 * R[2][j];
  }
  auto rtcr = [&](int i, int j) { return R[0][i] * CR[0][j] + R[1][i] * CR[1][j] + R[2][i] * CR[2][j]; };
  out[3] = rtcr(0, 0);
  out[4] = rtcr(0, 1);
  out[5] = rtcr(0, 2);
  out[6] = rtcr(1, 1);
  out[7] = rtcr(1, 2);
  out[8] = rtcr(2, 2);
  out[9] = 0.0;
}

// hop 2 of the source-frame kernel: the 80-B posed record
__device__ __forceinline__ void posed_issue(const GP_GLOBAL char* rec, v2d& r0, v2d& r1, v2d& r2, v2d& r3, v2d& r4) {
  asm volatile(
    "global_load_dwordx4 %0, %5, off\n\t"
    "global_load_dwordx4 %1, %5, off offset:16\n\t"
    "global_load_dwordx4 %2, %5, off offset:32\n\t"
    "global_load_dwordx4 %3, %5, off offset:48\n\t"
    "global_load_dwordx4 %4, %5, off offset:64"
    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4)
    : "v"(rec)
    : "memory");
}
template <int WAIT_YOUNGER>
__device__ __forceinline__ void posed_wait(v2d& r0, v2d& r1, v2d& r2, v2d& r3, v2d& r4) {
  if constexpr (WAIT_YOUNGER == 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : : "memory");
  } else {
    static_assert(WAIT_YOUNGER == 3, "one chunk request = 3 DMA instructions");
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : : "memory");
  }
}

template <int MODE, bool OUTER_F32, int PPT, bool SRC = false>
__global__ void __launch_bounds__(256, 4) vgicp_pipeline_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
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

  const SrcFrame sf = make_src_frame(Tl, Te);  // only the SRC instantiation uses it
  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  // one chunk: transform + hash, two-hop voxel lookup, [request for chunk j+2], algebra
  auto step = [&](auto ring_tag, int j, bool active, float px, float py, float pz, const float* cA) {
    constexpr bool RING = decltype(ring_tag)::value;
    const double dx = (double)px, dy = (double)py, dz = (double)pz;
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    const int cx = fast_floor(lx * f.map.inv_leaf), cy = fast_floor(ly * f.map.inv_leaf), cz = fast_floor(lz * f.map.inv_leaf);
    bool live = active;
    if (f.surface_validation && live && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (first + (size_t)j * kChunkPoints + lane))) live = false;
    v4f head;
    v2d c01, c23, c45;
    bool hit = false;
    // hop 1: the home line's four keys
    uint32_t l = coord_hash32(cx, cy, cz) & f.map.plmask;
    const GP_GLOBAL char* lines = (const GP_GLOBAL char*)f.map.plines;
    v4i k0, k1, k2, k3;
    line_issue(lines + 64 * (size_t)l, k0, k1, k2, k3);
    line_wait(k0, k1, k2, k3);
    if (j == 0) GP_TRACE(2);
    if (j == 1) GP_TRACE(4);
    int idx = line_match(k0, k1, k2, k3, cx, cy, cz);
    if (live && idx < 0 && k3.w >= 0) {  // full line, no match (rare): walk on
      for (;;) {
        l = (l + 1) & f.map.plmask;
        const GP_GLOBAL v4i* q = (const GP_GLOBAL v4i*)(lines + 64 * (size_t)l);
        const v4i a = q[0], b = q[1], c = q[2], d = q[3];
        idx = line_match(a, b, c, d, cx, cy, cz);
        if (idx >= 0 || d.w < 0) break;
      }
    }
    hit = live && idx >= 0;
    if constexpr (SRC) {
      // hop 2: the voxel's statistics in the source frame of this factor's linearisation pose (80 B, pose_records_kernel)
      v2d p0, p1, p2, p3, p4;
      posed_issue(hit ? (const GP_GLOBAL char*)f.posed + 8 * kPosedDoubles * (size_t)idx : lines, p0, p1, p2, p3, p4);
      if constexpr (RING) {
        if (j + 2 < PPT) {
          chunk_dma(points, covs, first + (size_t)(j + 2) * kChunkPoints, wbase + ((j + 2) % STAGES) * kChunkBytes, lane);
          posed_wait<3>(p0, p1, p2, p3, p4);
        } else {
          posed_wait<0>(p0, p1, p2, p3, p4);
        }
      } else {
        posed_wait<0>(p0, p1, p2, p3, p4);
      }
      if (hit) accumulate_terms_src<MODE, acc_t>(sf, px, py, pz, cA, p0, p1, p2, p3, p4, acc);
    } else {
      // hop 2: the record (lanes without a voxel read the line table again: any valid address)
      const GP_GLOBAL char* rec = hit ? (const GP_GLOBAL char*)f.map.records + 64 * (size_t)idx : lines;
      record_issue(rec, head, c01, c23, c45);
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
      if (hit) accumulate_terms<MODE, acc_t>(Tl, Te, f.map.leaf, px, py, pz, cA, cx, cy, cz, head, c01, c23, c45, acc);
    }
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
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = sum;
  }
  GP_TRACE(7);
}


// =====================================================================================================================
// vgicp_deep_pipeline_kernel -- the pipeline kernel with the voxel lookup taken out of the critical path as well.
//
// In vgicp_pipeline_kernel a wave still waits twice per step (keys, then record) and, because the four waves of a SIMD start
// together and do the same work, they wait together: ~1.2 us of every ~3 us step has the SIMD idle.  Here every wave keeps
// three chunks in different stages at once, so its own algebra covers its own latencies:
//     iteration j:  wait (keys of chunk j+1, record of chunk j; the youngest DMA keeps flying)
//                   match keys(j+1) -> request record(j+1)
//                   point(j+2) from LDS -> transform, floor, hash -> request keys(j+2)
//                   point + covariance (j) from LDS -> [request source chunk j+4 into the stage just read]
//                   algebra(j) with record(j)
// Cost: two records' worth of VGPRs in flight (3 waves per SIMD instead of 4) and a 4-stage ring (12 KB per wave, 48 KB per
// workgroup, 3 workgroups per CU).  All VMEM traffic of the ring path, the DMA included, is issued from inline asm: the
// compiler then tracks none of it and inserts no vmcnt waits of its own; every wait below is explicit and static.
// =====================================================================================================================
__device__ __forceinline__ void chunk_dma_asm(const GP_GLOBAL float* points, const GP_GLOBAL float* covs, size_t first_point, char* stage, int lane) {
  const GP_GLOBAL char* gp = (const GP_GLOBAL char*)(points + 3 * first_point);
  const GP_GLOBAL char* gc = (const GP_GLOBAL char*)(covs + 9 * first_point);
  const GP_GLOBAL char* a0 = lane < 48 ? gp + lane * 16 : gc + (lane - 48) * 16;
  const GP_GLOBAL char* a1 = gc + (lane + 16) * 16;
  const GP_GLOBAL char* a2 = gc + (lane + 80) * 16;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(GP_LDS char*)stage);
  uint32_t saved_m0;
  // M0 carries the LDS base of an LDS-DMA instruction; it is a reserved register for the compiler, so it is saved and restored
  // around the three requests instead of being declared clobbered
  asm volatile(
    "s_mov_b32 %0, m0\n\t"
    "s_mov_b32 m0, %4\n\t"
    "s_nop 0\n\t"
    "global_load_lds_dwordx4 %1, off\n\t"
    "s_add_i32 m0, %4, 0x400\n\t"
    "s_nop 0\n\t"
    "global_load_lds_dwordx4 %2, off\n\t"
    "s_add_i32 m0, %4, 0x800\n\t"
    "s_nop 0\n\t"
    "global_load_lds_dwordx4 %3, off\n\t"
    "s_mov_b32 m0, %0"
    : "=&s"(saved_m0)
    : "v"(a0), "v"(a1), "v"(a2), "s"(lds0)
    : "memory");
}

template <int MODE, bool OUTER_F32, int PPT>
__global__ void __launch_bounds__(256, 3) vgicp_deep_pipeline_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                                  const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                                  double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "tuned kernel covers the rigid linearise and the error evaluation");
  static_assert(PPT >= 4, "the prologue fills four ring stages");
  constexpr int NACC = MODE == MODE_ERR ? 2 : ACC_SIZE;
  constexpr int STAGES = 4;
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGES * kChunkBytes];  // 48 KB
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
  const size_t first = (size_t)tile.begin + (size_t)wave * (PPT * kChunkPoints);
  int wcount = __builtin_amdgcn_readfirstlane(tile.count) - wave * (PPT * kChunkPoints);
  wcount = wcount < 0 ? 0 : (wcount > PPT * kChunkPoints ? PPT * kChunkPoints : wcount);
  const bool ring = wcount == PPT * kChunkPoints && (((uintptr_t)f.points | (uintptr_t)f.covs) & 15) == 0;
  char* wbase = smem + wave * (STAGES * kChunkBytes);
  const GP_GLOBAL char* lines = (const GP_GLOBAL char*)f.map.plines;
  const GP_GLOBAL char* records = (const GP_GLOBAL char*)f.map.records;

  if (ring) {
    chunk_dma_asm(points, covs, first, wbase, lane);
    chunk_dma_asm(points, covs, first + kChunkPoints, wbase + kChunkBytes, lane);
  }
  const Pose Tl = inl.use ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)tile.factor);
  const Pose Te = MODE == MODE_ERR ? (inl.use ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)tile.factor)) : Tl;

  using acc_t = typename std::conditional<OUTER_F32, float, double>::type;
  acc_t acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = (acc_t)0;

  // stage S1: this lane's point of chunk c -> voxel coordinate -> request the 4 keys of its home line
  auto hash_and_request_keys = [&](int c, int& cx, int& cy, int& cz, bool& live, v4i& k0, v4i& k1, v4i& k2, v4i& k3) {
    const float* lp = reinterpret_cast<const float*>(wbase + (c % STAGES) * kChunkBytes);
    const double dx = (double)lp[3 * lane], dy = (double)lp[3 * lane + 1], dz = (double)lp[3 * lane + 2];
    const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
    const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
    const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
    cx = fast_floor(lx * f.map.inv_leaf);
    cy = fast_floor(ly * f.map.inv_leaf);
    cz = fast_floor(lz * f.map.inv_leaf);
    live = true;
    if (f.surface_validation && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * (first + (size_t)c * kChunkPoints + lane))) live = false;
    const uint32_t l = coord_hash32(cx, cy, cz) & f.map.plmask;
    line_issue(lines + 64 * (size_t)l, k0, k1, k2, k3);
  };
  // stage S2: keys -> voxel index (the rare full line without a match walks on, synchronously) -> request the record
  auto match_and_request_record = [&](int cx, int cy, int cz, bool live, const v4i& k0, const v4i& k1, const v4i& k2, const v4i& k3, bool& hit, v4f& head, v2d& c01,
                                      v2d& c23, v2d& c45) {
    int idx = line_match(k0, k1, k2, k3, cx, cy, cz);
    if (live && idx < 0 && k3.w >= 0) {
      uint32_t l = coord_hash32(cx, cy, cz) & f.map.plmask;
      for (;;) {
        l = (l + 1) & f.map.plmask;
        const GP_GLOBAL v4i* q = (const GP_GLOBAL v4i*)(lines + 64 * (size_t)l);
        const v4i a = q[0], b = q[1], c = q[2], d = q[3];
        idx = line_match(a, b, c, d, cx, cy, cz);
        if (idx >= 0 || d.w < 0) break;
      }
    }
    hit = live && idx >= 0;
    record_issue(hit ? records + 64 * (size_t)idx : lines, head, c01, c23, c45);
  };

  if (ring) {
    // rotating state (indices are compile-time after unrolling): coordinates / liveness of chunks j, j+1, j+2; two records
    int cxs[3], cys[3], czs[3];
    bool lives[3], hits[2];
    v4i k0, k1, k2, k3;
    v4f head[2];
    v2d c01[2], c23[2], c45[2];
    // ---- prologue: leaves {record(0), keys(1), DMA(3)} in flight, in that issue order.  (Threading the source requests
    // one by one between the hops -- chunk 0 alone first -- lands chunk 0 sooner, 2.2 instead of 3.4 us, but the hops then
    // queue behind everybody's chunk-1 burst: measured no better.) ----
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // chunk 0
    GP_TRACE(1);
    hash_and_request_keys(0, cxs[0], cys[0], czs[0], lives[0], k0, k1, k2, k3);
    chunk_dma_asm(points, covs, first + 2 * kChunkPoints, wbase + 2 * kChunkBytes, lane);
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3) : : "memory");  // chunk 1 and keys(0)
    match_and_request_record(cxs[0], cys[0], czs[0], lives[0], k0, k1, k2, k3, hits[0], head[0], c01[0], c23[0], c45[0]);
    hash_and_request_keys(1, cxs[1], cys[1], czs[1], lives[1], k0, k1, k2, k3);
    chunk_dma_asm(points, covs, first + 3 * kChunkPoints, wbase + 3 * kChunkBytes, lane);
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      const int cur = j & 1, nxt = cur ^ 1;
      // record(j) and keys(j+1) have landed when only the youngest DMA request (3 instructions, if one was issued) is left
      if (j == 0 || j + 3 < PPT) {
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(head[cur]), "+v"(c01[cur]), "+v"(c23[cur]), "+v"(c45[cur]) : : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(head[cur]), "+v"(c01[cur]), "+v"(c23[cur]), "+v"(c45[cur]) : : "memory");
      }
      if (j == 0) GP_TRACE(2);
      if (j == 1) GP_TRACE(3);
      if (j == 2) GP_TRACE(4);
      if (j == 3) GP_TRACE(5);
      if (j + 1 < PPT)
        match_and_request_record(cxs[(j + 1) % 3], cys[(j + 1) % 3], czs[(j + 1) % 3], lives[(j + 1) % 3], k0, k1, k2, k3, hits[nxt], head[nxt], c01[nxt], c23[nxt],
                                 c45[nxt]);
      if (j + 2 < PPT) hash_and_request_keys(j + 2, cxs[(j + 2) % 3], cys[(j + 2) % 3], czs[(j + 2) % 3], lives[(j + 2) % 3], k0, k1, k2, k3);
      const float* lp = reinterpret_cast<const float*>(wbase + (j % STAGES) * kChunkBytes);
      const float* lc = lp + kChunkPoints * 3;
      const float px = lp[3 * lane], py = lp[3 * lane + 1], pz = lp[3 * lane + 2];
      const float cA[6] = {lc[9 * lane], lc[9 * lane + 3], lc[9 * lane + 6], lc[9 * lane + 4], lc[9 * lane + 7], lc[9 * lane + 8]};
      if (j + 4 < PPT) {
        // the stage just read is free: the LDS reads above must have returned before the DMA may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        chunk_dma_asm(points, covs, first + (size_t)(j + 4) * kChunkPoints, wbase + (j % STAGES) * kChunkBytes, lane);
      }
      if (hits[cur])
        accumulate_terms<MODE, acc_t>(Tl, Te, f.map.leaf, px, py, pz, cA, cxs[j % 3], cys[j % 3], czs[j % 3], head[cur], c01[cur], c23[cur], c45[cur], acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_TRACE(6);
  } else {
    for (int j = 0; j < PPT; j++) {
      const int nj = wcount - j * kChunkPoints;  // wave-uniform
      if (nj <= 0) break;
      const bool active = lane < nj;
      const size_t i = first + (size_t)j * kChunkPoints + (active ? lane : 0);
      const GP_GLOBAL float* pp = points + 3 * i;
      const GP_GLOBAL float* cp = covs + 9 * i;
      const float px = pp[0], py = pp[1], pz = pp[2];
      const float cA[6] = {cp[0], cp[3], cp[6], cp[4], cp[7], cp[8]};
      const double dx = (double)px, dy = (double)py, dz = (double)pz;
      const double lx = Tl.r00 * dx + Tl.r01 * dy + Tl.r02 * dz + Tl.tx;
      const double ly = Tl.r10 * dx + Tl.r11 * dy + Tl.r12 * dz + Tl.ty;
      const double lz = Tl.r20 * dx + Tl.r21 * dy + Tl.r22 * dz + Tl.tz;
      const int cx = fast_floor(lx * f.map.inv_leaf), cy = fast_floor(ly * f.map.inv_leaf), cz = fast_floor(lz * f.map.inv_leaf);
      bool live = active;
      if (f.surface_validation && live && surface_rejected(Tl, lx, ly, lz, f.normals + 3 * i)) live = false;
      const uint32_t l = coord_hash32(cx, cy, cz) & f.map.plmask;
      v4i k0, k1, k2, k3;
      line_issue(lines + 64 * (size_t)l, k0, k1, k2, k3);
      line_wait(k0, k1, k2, k3);
      bool hit;
      v4f head;
      v2d c01, c23, c45;
      match_and_request_record(cx, cy, cz, live, k0, k1, k2, k3, hit, head, c01, c23, c45);
      record_wait<0>(head, c01, c23, c45);
      if (hit) accumulate_terms<MODE, acc_t>(Tl, Te, f.map.leaf, px, py, pz, cA, cx, cy, cz, head, c01, c23, c45, acc);
    }
  }

  // ---- reduction: as in vgicp_pipeline_kernel (transposition through the wave's own drained ring) ----
  constexpr int kRowStride = 68;
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
  } else {
    const int comp = lane >> 2, part = lane & 3;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
      for (int k = 0; k < 16; k++) wtrans[k * kRowStride + lane] = (double)acc[pass * 16 + k];
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
    ((GP_GLOBAL double*)partials)[(size_t)tile_idx * ACC_STRIDE + threadIdx.x] = sum;
  }
  GP_TRACE(7);
}

}  // namespace gp
