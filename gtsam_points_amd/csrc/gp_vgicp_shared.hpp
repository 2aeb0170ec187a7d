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
  const char* packed;    // the source's packed mirror (gp::SourceMirror: 2304 B per 64-point chunk) or null: what vgicp_stream_kernel<PK> streams instead of points / covs
  VoxelMapView map;
  int n;
  int surface_validation;
  int tile_begin;
  int tile_count;
};


// How the stream kernel (gp_vgicp_stream.hpp) deals the 64-point chunks of ONE large factor to the workgroups of a launch.  The chunk list is
// cut into eight contiguous shares, one per XCD (workgroup b runs on XCD b % 8, position q = b / 8 in the share).  The dispatcher places a
// share's workgroups in rounds of 32 (one per compute unit of the XCD), and a compute unit issues from its OLDEST waves first: with equal
// shares the workgroups of a CU end in the order they were placed, ~1 us apart, and the tail of the launch runs on a quarter of the waves
// (per-workgroup timeline of round 3, profiles/r03_sweep.jsonl).  So the rounds get DIFFERENT shares: workgroups of round r < last take n[r]
// chunks each (more for the early rounds), the workgroups from `last_begin` on split what is left evenly (`lo`, the first `extra` one more).
constexpr int kArriveStride = 512;  // 64-bit words between the arrival counters of two parts (fused finalize): 4 KB, another memory channel
constexpr int kFactorArriveStride = 16;  // 64-bit words between the arrival counters of two FACTORS (fused finalize of a batch): one 128-byte line each
constexpr int kStreamRound = 32;  // workgroups per dispatch round of an XCD = its compute units
struct StreamPlan {
  int wgs_per_xcd, last_begin;  // workgroups per XCD share; first workgroup of the share's last round (a multiple of 32)
  int tail;                     // points behind the last full chunk (the very last workgroup reads them with per-lane loads)
  int pad_;
  // per XCD: the XCDs do not run the same kernel equally fast (XCDs 4-7 end 0.5-0.9 us behind 0-3 on equal shares, whatever data they are
  // given: profiles/r03_sweep.jsonl), so their shares differ (host: kXcdWeightPermille in gp_vgicp.hip)
  int xbegin[kNumXCD];               // first chunk of the XCD's share
  int n[kNumXCD][3], pre[kNumXCD][3];  // rounds in front of the last one: chunks per workgroup, and chunks of the share in front of the round
  int before_last[kNumXCD];          // chunks of the share in front of its last round
  int lo[kNumXCD], extra[kNumXCD];   // the last round: `lo` chunks per workgroup, the first `extra` workgroups one more
};

// a single-factor launch carries its poses AND its factor descriptor in the kernel arguments: no H2D copy and no
// dependent descriptor loads on the latency path (2.8 us per workgroup in the timeline traces)
struct InlinePoses {
  double lin[16];
  double eval[16];
  FactorDesc factor;
  int use;
  int tile_points;
  int stagger;    // tuning knob of the pipeline kernel: odd wave slots start `stagger` x 512 clocks late (0 = off)
  int xcd_chunk;  // workgroup -> tile map of the pipeline kernel: 0 = every XCD walks a contiguous eighth of the tile list; c > 0 = the
                  // tile list is dealt to the XCDs in runs of c tiles (round robin), which evens out what the XCDs have to do
  StreamPlan plan;            // stream kernel, single-factor launches
  unsigned long long* trace;  // timeline build of the tile kernels (per batch: gp_vgicp_batch_set_trace_buffer); null = off
  // fused finalize (synchronous single-factor calls; GP_TUNE_FUSED_FINALIZE, on by default): a workgroup publishes its partial row write-through and
  // adds 1 to arrive[row / rows_per_part]; the workgroup whose add completes a part (arrive_target) sums the part's rows in the fixed order of
  // the split finalize kernel and hands the 32 sums to the host (fin_out slot + completion word): no second kernel, no kernel boundary
  unsigned long long* arrive;  // null = off
  int rows_per_part;  // > 0: fused finalize by parts of ONE factor's row list (counter g for rows [g * rows_per_part, ...)); 0 with `arrive` set: by FACTOR
                      // (counter factor_idx * kFactorArriveStride, reset by the last arriver; fin_out / fin_flags indexed by factor)
  int num_rows;
  unsigned long long arrive_target[16];  // the counters are monotonic: what arrive[g] reads when this launch's part g is complete
  double* fin_out;                       // host-mapped records, part g -> fin_out + g * fin_stride
  unsigned long long* fin_flags;         // host-mapped completion words
  unsigned long long fin_seq;
  int fin_stride;                        // doubles between the slots of two parts
  int pad2_;
};

struct TileDesc {
  int factor;
  int begin;  // first point
  int count;  // <= kTilePoints
  int row;    // row of the partials array this tile writes (rows of a factor are contiguous, the finalize kernels rely on it; the
              // EXECUTION order of the tile list may differ: tiles of factors that share a source cloud are interleaved)
};

__device__ __forceinline__ int xcd_swizzle(int b, int num_tiles) {
  // workgroup b -> tile index; tiles [x*per, (x+1)*per) go to XCD x (dispatcher places workgroup b on XCD b % 8)
  const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
  return (b % kNumXCD) * per + b / kNumXCD;
}

enum : int { MODE_LIN = 0, MODE_ERR = 1, MODE_LIN_GENERAL = 2 };


// the sums of one correspondence given M (6, symmetric), the source point p, q = T_eval p and r = mu_B - q, in f64:
// MODE_ERR: count and r^T M r; MODE_LIN: + the 27 target-side sums; MODE_LIN_GENERAL: + the explicit source-side and cross
// blocks (92 sums, vgicp_derivatives.cuh:57-70) for poses whose 3x3 block is not orthonormal.  Shared by the reference-shaped
// VGICP kernel (gp_vgicp.hip) and the GICP kernel's general path (gp_knn.hip).
template <int MODE>
__device__ __forceinline__ void accumulate_sums(const Pose& Tl, const double* m, double px, double py, double pz, double qx, double qy, double qz, double rx, double ry,
                                                double rz, double* acc) {
  const double mrx = m[0] * rx + m[1] * ry + m[2] * rz;
  const double mry = m[1] * rx + m[3] * ry + m[4] * rz;
  const double mrz = m[2] * rx + m[4] * ry + m[5] * rz;
  acc[ACC_COUNT] += 1.0;
  acc[ACC_ERR] += rx * mrx + ry * mry + rz * mrz;
  if constexpr (MODE != MODE_ERR) {
    for (int k = 0; k < 6; k++) acc[ACC_M + k] += m[k];
    // K = M S, S = [q]x ; K[:,0] = M[:,1] qz - M[:,2] qy ; K[:,1] = M[:,2] qx - M[:,0] qz ; K[:,2] = M[:,0] qy - M[:,1] qx
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
    // TL = -S K (= S^T M S), rows of -S: [0, qz, -qy], [-qz, 0, qx], [qy, -qx, 0]; upper triangle
    acc[ACC_TL + 0] += qz * k10 - qy * k20;
    acc[ACC_TL + 1] += qz * k11 - qy * k21;
    acc[ACC_TL + 2] += qz * k12 - qy * k22;
    acc[ACC_TL + 3] += qx * k21 - qz * k01;
    acc[ACC_TL + 4] += qx * k22 - qz * k02;
    acc[ACC_TL + 5] += qy * k02 - qx * k12;
    // b_t = [q x (M r); M r]
    acc[ACC_QXMR + 0] += qy * mrz - qz * mry;
    acc[ACC_QXMR + 1] += qz * mrx - qx * mrz;
    acc[ACC_QXMR + 2] += qx * mry - qy * mrx;
    acc[ACC_MR + 0] += mrx;
    acc[ACC_MR + 1] += mry;
    acc[ACC_MR + 2] += mrz;

    if constexpr (MODE == MODE_LIN_GENERAL) {
      // explicit source side, vgicp_derivatives.cuh:57-70: J_s = [R [p]x, -R] = [G, -R]
      const double Mf[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
      const double Rf[3][3] = {{Tl.r00, Tl.r01, Tl.r02}, {Tl.r10, Tl.r11, Tl.r12}, {Tl.r20, Tl.r21, Tl.r22}};
      const double Kf[3][3] = {{k00, k01, k02}, {k10, k11, k12}, {k20, k21, k22}};
      double G[3][3], Js[3][6], JtM[6][3], JsM[6][3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        G[r][0] = Rf[r][1] * pz - Rf[r][2] * py;
        G[r][1] = Rf[r][2] * px - Rf[r][0] * pz;
        G[r][2] = Rf[r][0] * py - Rf[r][1] * px;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          Js[r][c] = G[r][c];
          Js[r][3 + c] = -Rf[r][c];
        }
      }
      // JtM = J_t^T M = [S M; M] = [-K^T; M] ;  JsM = J_s^T M
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          JtM[r][c] = -Kf[c][r];
          JtM[3 + r][c] = Mf[r][c];
        }
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) JsM[r][c] = Js[0][r] * Mf[0][c] + Js[1][r] * Mf[1][c] + Js[2][r] * Mf[2][c];
      // H_s = JsM J_s : TL (0..2 x 0..2, upper), BL (3..5 x 0..2), BR (3..5 x 3..5, upper)
      int idx = ACCG_HS_TL;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) acc[idx++] += JsM[r][0] * Js[0][c] + JsM[r][1] * Js[1][c] + JsM[r][2] * Js[2][c];
      idx = ACCG_HS_BL;
#pragma unroll
      for (int r = 3; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc[idx++] += JsM[r][0] * Js[0][c] + JsM[r][1] * Js[1][c] + JsM[r][2] * Js[2][c];
      idx = ACCG_HS_BR;
#pragma unroll
      for (int r = 3; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) acc[idx++] += JsM[r][0] * Js[0][c] + JsM[r][1] * Js[1][c] + JsM[r][2] * Js[2][c];
      // H_ts = JtM J_s (6x6, row-major)
      idx = ACCG_HTS;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) acc[idx++] += JtM[r][0] * Js[0][c] + JtM[r][1] * Js[1][c] + JtM[r][2] * Js[2][c];
      // b_s = JsM r
#pragma unroll
      for (int r = 0; r < 6; r++) acc[ACCG_BS + r] += JsM[r][0] * rx + JsM[r][1] * ry + JsM[r][2] * rz;
    }
  }
}

// host-side launchers of the finalize kernels (defined in gp_vgicp.hip; used by gp_knn.hip for the GICP factor, whose
// partial rows have the same layout): one factor, rigid pose given both on the host (kernel arguments) and on the device
// general == true: the partial rows hold the 92 explicit sums (ACCG layout) and are expanded without the adjoint identity
// Completion words in host-mapped memory: a finalize kernel writes `seq` into done.flags[factor] behind its record, and the
// synchronous calls poll those words instead of going through hipStreamSynchronize -- the record is two PCIe writes away from the
// host the moment it exists, while the stream's completion signal has to wait for the end-of-kernel cache flush, the command
// processor and the runtime's signal handling (measured: ~6 us of a 35 us call).  flags == nullptr: no signalling.
struct DoneFlags {
  unsigned long long* flags = nullptr;
  unsigned long long seq = 0;
  unsigned long long* trace = nullptr;  // timeline build (gp_debug_set_trace_buffer): 8 shader-clock stamps of the finalize kernel
};
#define GP_FIN_TRACE(k)                                                            \
  do {                                                                             \
    if (done.trace && threadIdx.x == 0 && blockIdx.x == 0) done.trace[k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// every thread that has written part of the record calls this with wrote = true; all threads of the workgroup must call it
__device__ __forceinline__ void signal_done(const DoneFlags& done, int slot, bool wrote) {
  if (!done.flags) return;
  GP_FIN_TRACE(4);
  if (wrote) __threadfence_system();  // the record is visible to the host before ...
  __syncthreads();
  GP_FIN_TRACE(5);
  if (threadIdx.x == 0) __hip_atomic_store(done.flags + slot, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // ... the word that announces it
  GP_FIN_TRACE(6);
}
// host side: spin on the words (bounded), then fall back to the stream -- which also surfaces a failed kernel as an error
int wait_done(const unsigned long long* flags_host, size_t count, unsigned long long seq, hipStream_t stream, long spin_us = 100);

int launch_finalize_single(hipStream_t stream, const double* pose_dev, const double* pose_host, const double* partials, int num_tiles, gp_linearized6* out_dev,
                           bool general = false, DoneFlags done = {});

// is the 3x3 block of a column-major 4x4 pose orthonormal to 1e-9 with det > 0?  (GTSAM Pose3 values are; poses parsed from
// 6-digit text are not)
inline bool pose_is_rigid(const double* m) {
  for (int a = 0; a < 3; a++)
    for (int c = a; c < 3; c++) {
      const double d = m[4 * a] * m[4 * c] + m[4 * a + 1] * m[4 * c + 1] + m[4 * a + 2] * m[4 * c + 2] - (a == c ? 1.0 : 0.0);
      if (!(d < 1e-9 && d > -1e-9)) return false;
    }
  const double det = m[0] * (m[5] * m[10] - m[9] * m[6]) - m[4] * (m[1] * m[10] - m[9] * m[2]) + m[8] * (m[1] * m[6] - m[5] * m[2]);
  return det > 0.0;
}
int launch_finalize_error_single(hipStream_t stream, const double* partials, int num_tiles, double* out_dev, DoneFlags done = {});

}  // namespace gp
