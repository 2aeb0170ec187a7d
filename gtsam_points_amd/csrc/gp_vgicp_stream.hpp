// gp_vgicp_stream.hpp -- third generation of the rigid-pose tile kernel (round 3): the pipeline of gp_vgicp_tile2.hpp as a ROLLED loop over a
// per-wave stream of 64-point chunks whose length is a run-time, wave-uniform number.  Replaces vgicp_derivatives_kernel / vgicp_error_kernel
// (include/gtsam_points/cuda/kernels/vgicp_derivatives.cuh:15-139) + lookup_voxels_kernel incl. its surface validation
// (lookup_voxels.cuh:19-97) + the CUB reductions of src/gtsam_points/factors/integrated_vgicp_derivatives_{linearize,compute}.cu.
//
// Why: the second-generation kernel hands out whole tiles of PPT x 256 points, so the 1 M-point headline is
// 977 workgroups on 256 compute units -- 209 CUs carry four tiles and end at 12.2 us, 47 carry three and end at 10.5 us -- and every launch
// larger than one round of workgroups pays the cold start of the pipeline (first points 0.8 us, first hop 1.0 us, a 2.3 us first step) once per
// tile.  Here
//   * a large single-factor launch deals CHUNKS, not fixed tiles: at most 1024 workgroups = one resident round whatever the size of the cloud;
//     every XCD owns a contiguous eighth of the chunk list and its four dispatch rounds of 32 workgroups take DIFFERENT shares (StreamPlan,
//     gp_vgicp_shared.hpp): a compute unit issues from its oldest waves first, so equal shares end ~1 us apart in dispatch order and the
//     launch's tail runs on a quarter of the waves; a workgroup's chunks are dealt to its four waves as evenly as they go;
//   * clouds beyond one round of 4-chunk waves stay in ONE round: the waves simply stream more chunks through the same ring (an 8 M-point
//     source is 30 chunks per wave), so the cold start is paid once per wave, not once per 1024 points;
//   * surface validation (SV) rides in the same ring: the normals are a fourth 12-B LDS-DMA row per chunk, issued from inline asm like the rest,
//     so that the hand-counted vmcnt scheme survives (the compiler-tracked normals load of the round-2 kernel is what kept factors with
//     set_enable_surface_validation(true) -- and the whole batch they are in -- off the second-generation kernel);
//   * the trace buffer and the tuning knobs arrive in the kernel arguments (per batch), not through process-global device symbols.
// The arithmetic per point, the order of the waits and the reduction are those of gp_vgicp_tile2.hpp; the partition changes which points a
// wave adds up, so records differ from the second generation at the 1e-16 level (fixed order per launch geometry: still bit-reproducible).
#pragma once

#include "gp_vgicp_finalize.hpp"
#include "gp_vgicp_tile2.hpp"

namespace gp {

// normals: one more 12-B row per chunk (64 slots of 16 B, like the points)
template <bool NT>
__device__ __forceinline__ void chunk_dma12_row(const GP_GLOBAL char* urow, unsigned voff, char* slot) {
  chunk_dma12_pts<NT>(urow, voff, slot);
}

template <int N>
__device__ __forceinline__ void vm_wait_blk_n(v4i& blk) {
  static_assert(N == 0 || N == 3 || N == 4 || N == 5, "counts of the stream schedule");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(blk) : : "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(blk) : : "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(blk) : : "memory");
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(blk) : : "memory");
}

// the stream of one wave: `first` point, `n` full 64-point chunks, then `tail` (< 64) points read with per-lane loads
struct WaveWork {
  size_t first;
  int n;
  int tail;
};

// a workgroup's points [begin, begin + count) -> its four waves: the full chunks are dealt as evenly as they go (the first `extra` waves take
// one more), the points behind the last full chunk go to the last wave
template <int W = 4>
__device__ __forceinline__ WaveWork split_tile(int begin, int count, int wave) {
  static_assert(W == 4 || W == 8 || W == 16, "waves per workgroup");
  const int chunks = count >> 6, base = chunks / W, extra = chunks % W;  // (W is a power of two: shift and mask)
  WaveWork ww;
  ww.first = (size_t)begin + (size_t)(wave * base + (wave < extra ? wave : extra)) * kChunkPoints;
  ww.n = base + (wave < extra ? 1 : 0);
  ww.tail = wave == W - 1 ? (count & 63) : 0;
  return ww;
}

// the points workgroup (x = XCD, q = position in the XCD's share) of a planned single-factor launch owns -- also what the host writes into the
// tile table for the launches that cannot carry the descriptor in their arguments (make_stream_plan in gp_vgicp.hip)
struct PlanFields {  // the fields of the plan workgroup (x, q) needs
  int nr, pre, lo, extra, xbegin, L, before_last, gx, tail;
};
__host__ __device__ __forceinline__ PlanFields plan_fields(const StreamPlan& p, int x, int q) {
  const int r = q / kStreamRound < 2 ? q / kStreamRound : 2;  // (workgroups of the last round do not use nr / pre)
  return PlanFields{p.n[x][r], p.pre[x][r], p.lo[x], p.extra[x], p.xbegin[x], p.last_begin, p.before_last[x], p.wgs_per_xcd, p.tail};
}
__host__ __device__ __forceinline__ void plan_tile(const PlanFields& f, int x, int q, int* begin, int* count) {
  const int i = q % kStreamRound, k = q - f.L;
  const bool late = q >= f.L;
  const int n = late ? f.lo + (k < f.extra ? 1 : 0) : f.nr;
  const int c0 = late ? f.before_last + k * f.lo + (k < f.extra ? k : f.extra) : f.pre + i * f.nr;
  *begin = (f.xbegin + c0) * kChunkPoints;
  *count = n * kChunkPoints + ((x == kNumXCD - 1 && q == f.gx - 1) ? f.tail : 0);  // the very last workgroup also takes the points behind the last full chunk
}

// The last workgroup of a part (fused finalize, see InlinePoses): all 256 threads sum the part's rows -- every one stored write-through by its
// workgroup before that workgroup's add to the arrival counter -- in exactly the order of the split form of vgicp_finalize_rigid_kernel<256>
// (8 slices of rows, pairwise tree over a slice's 32 rows per batch, the wave's two slices, then (w0 + w2) + (w1 + w3)): the records are bit-identical to the two-kernel form
// (tests/test_vgicp_gpu.py::test_fused_finalize_equals_the_two_kernel_form).  wsum: 4 x 32 doubles of LDS nobody else uses any more.
__device__ __forceinline__ void finalize_part_rows(const double* __restrict__ partials, const int row_begin, const int row_count, double* wsum, double* out,
                                                   unsigned long long* flag, const unsigned long long seq, unsigned long long* tr, const unsigned long long t_rows) {
  constexpr int kSlices = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int comp = threadIdx.x & 31, slice = threadIdx.x >> 5;
  // read past this CU's L1 and the XCD's L2 view of other XCDs' lines (sc1), all of a lane's rows requested in one batch
  const unsigned long long* base = reinterpret_cast<const unsigned long long*>(partials + (size_t)row_begin * ACC_STRIDE + comp);
  double total = 0.0;
  // (wide workgroups, W > 4: the first four waves do what the 256 threads of the narrow form do -- the same order, the same bits as the split finalize kernel --
  // the others only meet them at the barrier)
  for (int t0 = slice; t0 < row_count && wave < 4; t0 += 32 * kSlices) {
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
      const int t = t0 + k * kSlices;
      v[k] = t < row_count ? __builtin_bit_cast(double, __hip_atomic_load(base + (size_t)t * ACC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
    }
#pragma unroll
    for (int w = 16; w > 0; w >>= 1) {
#pragma unroll
      for (int k = 0; k < w; k++) v[k] += v[k + w];
    }
    total += v[0];
  }
  total += __shfl_xor(total, 32, 64);  // the wave's two slices
  if (tr && threadIdx.x == 0) tr[14] = __builtin_amdgcn_s_memrealtime();
  if (lane < 32 && wave < 4) wsum[wave * 32 + lane] = total;
  __syncthreads();
  if (wave != 0) return;
  // the record slot and the completion word are host-mapped (uncached on this side): the sums go out as system-scope stores, the word follows
  // their acknowledgement -- no cache write-back (`__threadfence_system()` = buffer_wbl2 + buffer_inv costs 3.5 us here and has nothing to write back)
  double s = lane < 32 ? (wsum[lane] + wsum[64 + lane]) + (wsum[32 + lane] + wsum[96 + lane]) : 0.0;
  if (lane == 32) s = __builtin_bit_cast(double, t_rows);                           // time stamps ride in the same store: when the part's last row was in,
  if (lane == 33) s = __builtin_bit_cast(double, __builtin_amdgcn_s_memrealtime());  // and when its sums leave
  if (lane < 34) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(out + lane), "v"(s) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(flag), "v"(seq) : "memory");
  if (tr && lane == 0) tr[15] = __builtin_amdgcn_s_memrealtime();
}

// The same for an error evaluation: the part's sum of r^T M r (row entry ACC_ERR) in the fixed order thread -> wave tree -> (w0 + w1) + (w2 + w3).
// Shared by the fused form (the part's last tile workgroup) and by vgicp_finalize_error_parts_kernel (two-kernel form): identical bits.
__device__ __forceinline__ void finalize_part_error(const double* __restrict__ partials, const int row_begin, const int row_count, double* wsum, double* out,
                                                    unsigned long long* flag, const unsigned long long seq) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long* base = reinterpret_cast<const unsigned long long*>(partials + (size_t)row_begin * ACC_STRIDE + ACC_ERR);
  double s = 0.0;
  for (int t = threadIdx.x; t < row_count && wave < 4; t += 256)  // (wide workgroups: the first four waves, as above)
    s += __builtin_bit_cast(double, __hip_atomic_load(base + (size_t)t * ACC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0 && wave < 4) wsum[wave] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double total = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(out), "v"(total) : "memory");
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(flag), "v"(seq) : "memory");
}

// EXP (measurement instantiations of round 5, never the default; GP_TUNE_EXPERIMENT selects them for a synchronous planned single-factor linearise):
//   1 = every workgroup touches its share of the map's block grid right behind its first request, so that an XCD's L2 holds the whole grid (1 MB for the headline map)
//       by the time the first hop 1 asks for it (VERDICT r04 #1b: the cold first chunk);  2 = R C_A R^T in f32 (accumulate_core2<ROT32>: the upper bound of #1c)
// W (round 6): waves per workgroup.  4 = the product geometry (1024 workgroups of 256 threads for a planned launch, four per compute unit);
//   16 = ONE 1024-thread workgroup per compute unit (256 workgroups: a quarter of the dispatches, partial rows and arrivals; the sixteen waves' sums meet in LDS in
//   a fixed pairwise tree), 8 = two per compute unit.  Same waves, same rings, same per-wave schedule: only who shares a row changes.
//   MEASURED AND NOT ADOPTED (profiles/r06_wg_geometry.jsonl, r06_wg_geometry_kernel_stats.csv; commit 1af148a carries the launch code and the GP_TUNE_WG_WAVES knob):
//   in step, alternating, whole fused kernel 12.40-12.46 us (W = 4) / 12.41 (16) / 12.40-12.51 (8); rocprofv3 averages 14.09 / 14.23 / 13.95 us -- the fused tail is a
//   latency chain (row store -> arrival -> loads -> sums over PCIe), not a count of rows or dispatches.  Only W = 4 is instantiated.
template <int MODE, bool NT, bool INL, bool SV, bool PK, bool TRACE = false, int EXP = 0, int W = 4>
__global__ void __launch_bounds__(64 * W, 4) vgicp_stream_kernel(const FactorDesc* __restrict__ factors, const TileDesc* __restrict__ tiles, int num_tiles,
                                                               const double* __restrict__ poses_lin, const double* __restrict__ poses_eval, const InlinePoses inl,
                                                               double* __restrict__ partials) {
  static_assert(MODE == MODE_LIN || MODE == MODE_ERR, "rigid linearise and error evaluation");
  constexpr int NACC = MODE == MODE_ERR ? 2 : 32;
  // PK: the source stream is the factor's PACKED MIRROR (SourceMirror, gp_host.hpp): per 64-point chunk 2304 contiguous bytes = 64 points (12 B) | 64 x
  // (c00, c01, c02) | 64 x (c11, c12, c22) -- the symmetric covariance the algebra uses, 36 B per point instead of the API layout's 48: three 12-B DMA rows
  // per chunk instead of four, and no symmetry test.  Built only from covariances that are symmetric to the last bit, so the six floats ARE the
  // caller's: records are bit-identical to the unpacked stream (tests/test_vgicp_gpu.py::test_packed_mirror_is_bit_identical).
  constexpr int KC = PK ? 2 : 3;                // covariance rows per chunk
  constexpr int K = 1 + KC + (SV ? 1 : 0);      // vector-memory requests per chunk: points (+ normals) + covariance rows
  constexpr int kNrmSlotBytes = SV ? kPtsSlotBytes : 0;
  constexpr int kChunkSlotBytes = PK ? kCovSlotBytes : kPtsSlotBytes + kCovSlotBytes;  // one chunk of the ring: 3 KB packed (points in its first KB), else 1 + 3 KB
  constexpr int kRingBytes = 2 * kChunkSlotBytes + 2 * kNrmSlotBytes;
  constexpr int kWaveBytes = kRingBytes > kWaveLdsBytes ? kRingBytes : kWaveLdsBytes;  // 10 KB for the unpacked stream with normals, else 8.5 KB (the reduction's)
  static_assert(kWaveBytes >= kWaveLdsBytes, "the reduction needs 8.5 KB of the wave's region");
  static_assert(W * kWaveBytes <= 160 * 1024, "one workgroup's rings must fit the compute unit's LDS");
  __shared__ __attribute__((aligned(16))) char smem[W * kWaveBytes];
  const unsigned long long t_begin = INL ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz constant clock; used by the fused form's own time stamps (below)
  // ---- what the first source request needs, in as few dependent scalar-load round trips as possible.  The in-argument form reads every field
  // it may need -- the plan's entries for this workgroup included -- up front and pins them with an empty asm: left alone, hipcc sinks those loads
  // into the branches of the tile arithmetic (five dependent s_load / s_waitcnt rounds in front of the first DMA, 1.26 us from workgroup start
  // to the first points instead of 0.83: per-workgroup timeline of round 3) ----
  const int bx = blockIdx.x % kNumXCD, bq = blockIdx.x / kNumXCD;
  int tile_idx;  // index into the tile list: XCD x walks a contiguous eighth of it
  FactorDesc f;
  WaveWork ww;
  int factor_idx = 0, row;
  int wgs_per_xcd = 0;  // (EXP 1)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (INL) {
    PlanFields pf = plan_fields(inl.plan, bx, bq);
    wgs_per_xcd = pf.gx;
    int tile_points = inl.tile_points, fn = inl.factor.n;
    const float* fpts = inl.factor.points;
    const float* fcov = inl.factor.covs;
    const float* fnrm = inl.factor.normals;
    const char* fpk = inl.factor.packed;
    asm volatile("" : "+s"(pf.nr), "+s"(pf.pre), "+s"(pf.lo), "+s"(pf.extra), "+s"(pf.xbegin), "+s"(pf.L), "+s"(pf.before_last), "+s"(pf.gx), "+s"(pf.tail), "+s"(tile_points),
                 "+s"(fn), "+s"(fpts), "+s"(fcov), "+s"(fnrm), "+s"(fpk));
    int begin, count;
    if (tile_points == 0) {  // one large factor: the tile list is the StreamPlan
      tile_idx = bx * pf.gx + bq;
      plan_tile(pf, bx, bq, &begin, &count);
    } else {  // fixed tiles of tile_points
      // (the in-argument launch always uses the contiguous map, whatever xcd_chunk says; a grid rounded up for another map must not run a tile twice:
      // the fused finalize counts arrivals -- ADVICE r03)
      const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
      tile_idx = bx * per + bq;
      if (bq >= per || tile_idx >= num_tiles) return;
      begin = tile_idx * tile_points;
      count = min(tile_points, fn - begin);
    }
    f = inl.factor;
    f.points = fpts;
    f.covs = fcov;
    f.normals = fnrm;
    f.packed = fpk;
    row = tile_idx;
    ww = split_tile<W>(begin, count, wave);
  } else {
    if (inl.xcd_chunk > 0) {
      const int c = inl.xcd_chunk;
      tile_idx = ((bq / c) * kNumXCD + bx) * c + (bq % c);
    } else {
      const int per = (num_tiles + kNumXCD - 1) / kNumXCD;
      tile_idx = bx * per + bq;
    }
    if (tile_idx >= num_tiles) return;
    const TileDesc tile = tiles[tile_idx];
    f = factors[tile.factor];
    factor_idx = tile.factor;
    row = tile.row;
    ww = split_tile<W>(__builtin_amdgcn_readfirstlane(tile.begin), __builtin_amdgcn_readfirstlane(tile.count), wave);
  }
  unsigned long long* trace = TRACE ? inl.trace : nullptr;
  GP_TRACE(0);
  if constexpr (INL && MODE == MODE_LIN) {
    // fused form: the first workgroup of every XCD leaves its start time in the part's host slot (word 34), fire and forget; the parts' finalizers add when
    // their last row was in (32) and when their sums left (33).  The host turns the three into the duration of the streaming part and of the
    // whole kernel as THIS step ran it (gp_vgicp_batch_device_times): the step's own clock, no events, no profiler
    // (parts form only: in the by-factor form fin_out slot i is factor i's record, not a part's scratch slot)
    if (inl.arrive && inl.rows_per_part > 0 && blockIdx.x < kNumXCD && threadIdx.x == 0)
      asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(inl.fin_out + (size_t)blockIdx.x * inl.fin_stride + 34), "v"(t_begin) : "memory");
  }
  if constexpr (TRACE) {
    if (trace && threadIdx.x == 0) {
      trace[(size_t)tile_idx * 16 + 10] = __builtin_amdgcn_s_memrealtime();
      trace[(size_t)tile_idx * 16 + 8] = __builtin_amdgcn_s_getreg(GP_GETREG_HW_ID);
      trace[(size_t)tile_idx * 16 + 9] = __builtin_amdgcn_s_getreg(GP_GETREG_XCC_ID);
    }
  }
  const int lane = threadIdx.x & 63;
  const int n = __builtin_amdgcn_readfirstlane(ww.n);
  const int tail = __builtin_amdgcn_readfirstlane(ww.tail);
  const size_t first = ww.first;
  char* wbase = smem + wave * kWaveBytes;
  // ring layout of a wave.  Unpacked: [points 0 | points 1 | normals 0 | normals 1 | covariances 0 | covariances 1]; packed: [chunk 0 | chunk 1 | normals 0 |
  // normals 1] with a chunk = [points | (c00, c01, c02) | (c11, c12, c22)], 1 KB each
  auto pslot = [&](int par) { return PK ? wbase + par * kCovSlotBytes : wbase + par * kPtsSlotBytes; };
  auto nslot = [&](int par) { return PK ? wbase + 2 * kCovSlotBytes + par * kNrmSlotBytes : wbase + 2 * kPtsSlotBytes + par * kNrmSlotBytes; };
  auto cslot = [&](int par) { return PK ? wbase + par * kCovSlotBytes : wbase + 2 * kPtsSlotBytes + 2 * kNrmSlotBytes + par * kCovSlotBytes; };
  // (a wave's first point is a multiple of 64 in every tile list: plan shares, fixed tiles and split_tile all deal whole chunks)
  const GP_GLOBAL char* upts = PK ? uniform_ptr((const GP_GLOBAL char*)f.packed + (first >> 6) * (size_t)kPackedChunkBytes) : uniform_ptr((const GP_GLOBAL char*)as_global(f.points) + 12 * first);
  const GP_GLOBAL char* ucov = PK ? upts : uniform_ptr((const GP_GLOBAL char*)as_global(f.covs) + 36 * first);
  const GP_GLOBAL char* unrm = SV ? uniform_ptr((const GP_GLOBAL char*)as_global(f.normals) + 12 * first) : nullptr;
  const unsigned voff = (unsigned)lane * 12u;
  constexpr size_t kSrcChunkStride = PK ? (size_t)kPackedChunkBytes : (size_t)kChunkPoints * 12, kCovChunkStride = PK ? (size_t)kPackedChunkBytes : (size_t)kChunkPoints * 36;
  // requests of chunk j (its rows start j * 64 points behind the wave's first point); par = j & 1 = which half of the ring
  auto dma_head = [&](int j, int par) {  // what the front half of a chunk reads: its points (and normals)
    chunk_dma12_pts<NT>(upts + (size_t)j * kSrcChunkStride, voff, pslot(par));
    if constexpr (SV) chunk_dma12_row<NT>(unrm + (size_t)j * (kChunkPoints * 12), voff, nslot(par));
  };
  auto dma_cov = [&](int j, int par) {
    if constexpr (PK) chunk_dma12_cov_packed<NT>(ucov + (size_t)j * kCovChunkStride, voff, cslot(par));
    else chunk_dma12_cov<NT>(ucov + (size_t)j * kCovChunkStride, voff, cslot(par));
  };

  if (n > 0) dma_head(0, 0);

  const Pose Tl = INL ? load_pose(inl.lin) : load_pose(poses_lin + 16 * (size_t)factor_idx);
  const Pose Te = MODE == MODE_ERR ? (INL ? load_pose(inl.eval) : load_pose(poses_eval + 16 * (size_t)factor_idx)) : Tl;
  const double leaf = uniform_f64(f.map.leaf), inv_leaf = uniform_f64(f.map.inv_leaf), half_leaf = uniform_f64(0.5 * f.map.leaf);
  const int glo0 = f.map.glo[0], glo1 = f.map.glo[1], glo2 = f.map.glo[2];
  const unsigned gd0 = __builtin_amdgcn_readfirstlane((unsigned)f.map.gdim[0]), gd1 = __builtin_amdgcn_readfirstlane((unsigned)f.map.gdim[1]), gd2 = (unsigned)f.map.gdim[2];
  const GP_GLOBAL char* gblocks = uniform_ptr((const GP_GLOBAL char*)f.map.gblocks);
  const GP_GLOBAL char* records = uniform_ptr((const GP_GLOBAL char*)f.map.records);

  v4i warm = {0, 0, 0, 0};
  if constexpr (EXP == 1) {
    // workgroup q of an XCD asks for bytes [(k * wgs + q) * 4096 + 16 tid, + 16) of the grid, k = 0 .. 3: 2 MB per XCD at 128 workgroups.  Behind the points'
    // request in the queue; both are retired by the vmcnt(0) in front of the first transform
    const unsigned gbytes = gd0 * gd1 * gd2 * 16u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned off = ((unsigned)(k * wgs_per_xcd + bq) * 256u + threadIdx.x) * 16u;
      if (off < gbytes) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(warm) : "v"(off), "s"(gblocks) : "memory");
    }
  }
  // the translation lives in vector registers: a VOP3 instruction reads ONE scalar operand (gp_vgicp_tile2.hpp)
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
    int pos;  // bit of the voxel inside its block; < 0: outside the grid's box, an inactive lane, or rejected by the surface validation
  };
  // front half: transform, (surface validation,) voxel coordinate, hop 1 issued
  auto front = [&](float pxf, float pyf, float pzf, float nxf, float nyf, float nzf, bool active, Ahead& P) {
#pragma clang fp contract(off)  // (every instantiation computes the same bits: the fused multiply-adds are the ones written out)
    const double dx = (double)pxf, dy = (double)pyf, dz = (double)pzf;
    const double lx = __builtin_fma(Tl.r00, dx, __builtin_fma(Tl.r01, dy, __builtin_fma(Tl.r02, dz, tvx)));
    const double ly = __builtin_fma(Tl.r10, dx, __builtin_fma(Tl.r11, dy, __builtin_fma(Tl.r12, dz, tvy)));
    const double lz = __builtin_fma(Tl.r20, dx, __builtin_fma(Tl.r21, dy, __builtin_fma(Tl.r22, dz, tvz)));
    // voxel coordinate = floor(l * (1 / leaf)): the CPU map's rule (util/fast_floor.hpp:12-15, gaussian_voxelmap_cpu.cpp:59-61);
    // centre - l = leaf (floor(u) + 0.5 - u) = leaf/2 - leaf fract(u): the large coordinates never meet
    const double ux = lx * inv_leaf, uy = ly * inv_leaf, uz = lz * inv_leaf;
    const int cx = (int)__builtin_floor(ux), cy = (int)__builtin_floor(uy), cz = (int)__builtin_floor(uz);
    if constexpr (MODE == MODE_ERR) {
      const double ex_ = __builtin_fma(Te.r00, dx, __builtin_fma(Te.r01, dy, __builtin_fma(Te.r02, dz, Te.tx)));
      const double ey_ = __builtin_fma(Te.r10, dx, __builtin_fma(Te.r11, dy, __builtin_fma(Te.r12, dz, Te.ty)));
      const double ez_ = __builtin_fma(Te.r20, dx, __builtin_fma(Te.r21, dy, __builtin_fma(Te.r22, dz, Te.tz)));
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
    bool live = active && finite3(pxf, pyf, pzf);
    if constexpr (SV) {
      // lookup_voxels.cuh:41-50: reject when normalized(q) . (R n) > cos(80 deg), at the LINEARISATION pose (the error evaluation keeps the
      // correspondences of the linearise, vgicp_derivatives.cuh:85-139).  s / |q| > c  <=>  s > 0 and s^2 > c^2 |q|^2: no square root, no division
      const double nx = (double)nxf, ny = (double)nyf, nz = (double)nzf;
      const double tnx = __builtin_fma(Tl.r02, nz, __builtin_fma(Tl.r01, ny, Tl.r00 * nx)), tny = __builtin_fma(Tl.r12, nz, __builtin_fma(Tl.r11, ny, Tl.r10 * nx));
      const double tnz = __builtin_fma(Tl.r22, nz, __builtin_fma(Tl.r21, ny, Tl.r20 * nx));
      const double s = __builtin_fma(lz, tnz, __builtin_fma(ly, tny, lx * tnx));
      const double qq = __builtin_fma(lz, lz, __builtin_fma(ly, ly, lx * lx));
      if (s > 0.0 && s * s > (0.174 * 0.174) * qq) live = false;
    }
    const unsigned bx = (unsigned)((cx >> 2) - glo0), by = (unsigned)((cy >> 2) - glo1), bz = (unsigned)((cz >> 2) - glo2);
    const bool inbox = (bx < gd0) & (by < gd1) & (bz < gd2);
    const unsigned lin = inbox ? mad24s(mad24s(bz, gd1, by), gd0, bx) : 0u;  // < 2^24 blocks
    P.pos = (inbox && live) ? (((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3)) : -1;
    grid_issue_s(gblocks, lin * 16u, P.blk);
  };
  auto front_ring = [&](int par, Ahead& P) {
    const v3f pt = *reinterpret_cast<const v3f*>(pslot(par) + 16 * lane);
    v3f nr = {0.f, 0.f, 0.f};
    if constexpr (SV) nr = *reinterpret_cast<const v3f*>(nslot(par) + 16 * lane);
    front(pt.x, pt.y, pt.z, nr.x, nr.y, nr.z, true, P);
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
  auto cov_ring = [&](int par, double* a) {
    if constexpr (PK) {  // the six floats of the symmetric matrix: rows 1 and 2 of the chunk, one conflict-free ds_read_b96 each
      const char* c = cslot(par) + 16 * lane;
      const v3f ca = *reinterpret_cast<const v3f*>(c + kPtsSlotBytes), cb = *reinterpret_cast<const v3f*>(c + 2 * kPtsSlotBytes);
      a[0] = (double)ca.x, a[1] = (double)ca.y, a[2] = (double)ca.z, a[3] = (double)cb.x, a[4] = (double)cb.y, a[5] = (double)cb.z;
    } else {
      const char* c = cslot(par) + 48 * lane;
      const v3f c0 = *reinterpret_cast<const v3f*>(c), c1 = *reinterpret_cast<const v3f*>(c + 16), c2 = *reinterpret_cast<const v3f*>(c + 32);
      const float c9[9] = {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z};
      load_cov6(c9, a);
    }
  };

  Ahead Pc;  // the chunk whose record has landed (front half and lookup done)
  v4f head;
  v2d c01, c23, c45;
  double a[6];
  // (The counts below were validated with HIP 7.2.26015 / AMD clang 22.0.0git roc-7.2.0.  They are a property of the COMPILED code: csrc/count_waits.py walks the device
  //  assembly of every instantiation at build time and the Makefile stops when a compiler-placed wait or register touch appears inside the schedule.)
  // HAZARD the schedule below is built around: the destination registers of an asm-issued load hold nothing until the matching s_waitcnt, but
  // the compiler believes they are defined at the issue.  Any copy it places between the two -- a phi at a loop back-edge, or the operand copy in
  // front of one of TWO alternative wait statements -- reads the registers before the data lands (round 3: exactly that, wild record offsets,
  // a memory fault).  Therefore (i) the loop is rotated so that NOTHING asm-issued is in flight at its back-edge (the body ends with the
  // vmcnt(0) behind hop 2), and (ii) every register-tied wait is ONE unconditional statement; run-time alternatives only add an untied
  // `s_waitcnt` in front of it.
  if (n > 0) {
    // points first: only chunk 0's points (and normals) are in flight, so the first transform and hop 1 do not queue behind everybody's
    // covariances; those follow hop 1 (they are needed behind hop 2), the head of chunk 1 goes out before hop 2 and its covariances behind it
    vm_wait<0>();
    if constexpr (EXP == 1) asm volatile("" : : "v"(warm));  // (the registers the warm-up loads land in stay reserved until here)
    GP_TRACE(1);
    front_ring(0, Pc);  // in flight: H0
    dma_cov(0, 0);
    const bool has1 = n > 1;
    if (has1) dma_head(1, 1);  // in flight: H0, C0 x KC, head of chunk 1 (K - KC requests)
    else vm_wait<KC>();
    vm_wait_blk_n<K>(Pc.blk);
    GP_TRACE(2);
    bool hit = back_issue(Pc, head, c01, c23, c45);
    if (has1) dma_cov(1, 1);  // [C0 x KC, head 1, R0 x4, C1 x KC]
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vm_wait_rec<KC>(head, c01, c23, c45);  // the record, the covariances of chunk 0 and the head of chunk 1 (C1 may still travel: it is older than
                                          // the next hop 1, whose wait below retires it)
    // steady state, rotated: the body starts where the record of chunk j has landed and ends where the record of chunk j+1 has.
    //   front half of chunk j+1 (its hop 1 travels under the algebra below) -> covariance of chunk j out of LDS -> chunk j+2 requested into the
    //   places of chunk j -> algebra of chunk j -> wait hop 1 of chunk j+1 -> its hop 2 -> wait (everything)
    for (int j = 0;; j++) {
      const int par = j & 1;
      const bool more = j + 1 < n, more2 = j + 2 < n;  // wave-uniform
      Ahead Pn;
      if (more) front_ring(par ^ 1, Pn);
      cov_ring(par, a);
      if (more2) {  // chunk j+2 takes the places of chunk j, whose points and covariance have just been read
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_head(j + 2, par);
        dma_cov(j + 2, par);
      }
      if (hit) accumulate_core2<MODE, EXP == 2>(Tl, a, c01, c23, c45, Pc.ex + head.x, Pc.ey + head.y, Pc.ez + head.z, Pc.qx, Pc.qy, Pc.qz, acc);
      if constexpr (TRACE) {
        if (j == 0) GP_TRACE(3);
        if (j == 1) GP_TRACE(5);
      }
      if (!more) break;
      if (!more2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      vm_wait_blk_n<K>(Pn.blk);  // [H(j+1), chunk j+2 x K]
      if constexpr (TRACE) {
        if (j == 0) GP_TRACE(4);
      }
      hit = back_issue(Pn, head, c01, c23, c45);
      vm_wait_rec<0>(head, c01, c23, c45);  // the record, and chunk j+2, which the next front half reads: nothing is in flight at the back-edge
      Pc.ex = Pn.ex;
      Pc.ey = Pn.ey;
      Pc.ez = Pn.ez;
      Pc.qx = Pn.qx;
      Pc.qy = Pn.qy;
      Pc.qz = Pn.qz;
    }
    vm_wait<0>();
  }
  if (tail > 0) {
    // the points behind the wave's last full chunk (end of a factor): per-lane loads, same arithmetic.  The loads are issued from inline asm
    // like everything else here: a load hipcc tracks itself makes it guard registers of the ring path with vmcnt(0) waits of its own.
    const GP_GLOBAL float* points = as_global(f.points);
    const GP_GLOBAL float* covs = as_global(f.covs);
    const bool active = lane < tail;
    const size_t i = first + (size_t)n * kChunkPoints + (active ? lane : 0);
    v3f pt, nr = {0.f, 0.f, 0.f};
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
    if constexpr (SV) {
      const GP_GLOBAL float* normals = as_global(f.normals);
      asm volatile(
        "global_load_dwordx3 %0, %1, off\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(nr)
        : "v"(normals + 3 * i)
        : "memory");
    }
    const float c9[9] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w, cc};
    Ahead P;
    front(pt.x, pt.y, pt.z, nr.x, nr.y, nr.z, active, P);
    vm_wait_blk_n<0>(P.blk);
    const bool hit = back_issue(P, head, c01, c23, c45);
    vm_wait_rec<0>(head, c01, c23, c45);
    load_cov6(c9, a);
    if (hit) accumulate_core2<MODE>(Tl, a, c01, c23, c45, P.ex + head.x, P.ey + head.y, P.ez + head.z, P.qx, P.qy, P.qz, acc);
  }

  GP_TRACE(6);
  // ---- reduction (gp_vgicp_tile2.hpp): the wave's drained ring becomes a 32 x 64 f32 transposition buffer (row stride 66 floats), every
  // lane sums 32 values of one component (four f32 partial sums met in f64), lane pairs meet with one swap; the 4-wave sum goes through the
  // last 256 B of each wave's region; one 32-double partial per workgroup (fixed order: bit-reproducible) ----
  constexpr int kRowStrideF = 66;
  static_assert(32 * kRowStrideF * 4 + 32 * 8 <= kWaveBytes, "f32 transposition buffer + wave sums must fit the wave's LDS region");
  float* wtf = reinterpret_cast<float*>(wbase);
  double* wsums = reinterpret_cast<double*>(wbase + kWaveBytes - 32 * 8);
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
      if constexpr (W == 4) {
        const double* w0 = reinterpret_cast<const double*>(smem + 1 * kWaveBytes - 32 * 8);
        const double* w1 = reinterpret_cast<const double*>(smem + 2 * kWaveBytes - 32 * 8);
        const double* w2 = reinterpret_cast<const double*>(smem + 3 * kWaveBytes - 32 * 8);
        const double* w3 = reinterpret_cast<const double*>(smem + 4 * kWaveBytes - 32 * 8);
        sum = (w0[threadIdx.x] + w1[threadIdx.x]) + (w2[threadIdx.x] + w3[threadIdx.x]);
      } else {  // fixed pairwise tree over the W waves in wave order: ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)) ...
        double v[W];
#pragma unroll
        for (int k = 0; k < W; k++) v[k] = reinterpret_cast<const double*>(smem + (k + 1) * kWaveBytes - 32 * 8)[threadIdx.x];
#pragma unroll
        for (int w = 1; w < W; w <<= 1) {
#pragma unroll
          for (int k = 0; k < W; k += 2 * w) v[k] += v[k + w];
        }
        sum = v[0];
      }
    }
    GP_GLOBAL double* dst = (GP_GLOBAL double*)partials + (size_t)row * ACC_STRIDE + threadIdx.x;
    if (inl.arrive) {
      // fused finalize: the row goes out write-through (sc0 sc1: visible to every XCD once the store is acknowledged, no release fence --
      // MI355X_MICROARCH.md, inter-workgroup visibility), then ONE relaxed agent-scope add announces it
      asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(dst), "v"(sum) : "memory");
    } else {
      *dst = sum;
    }
  }
  if (inl.arrive && inl.rows_per_part > 0) {  // (kernel arguments: uniform over the launch) fused finalize by PARTS of one large factor's row list
    if constexpr (INL) {
      unsigned long long* tr = nullptr;
      if constexpr (TRACE) tr = trace ? trace + (size_t)tile_idx * 16 : nullptr;
      const int part = row / inl.rows_per_part;
      int* const last = reinterpret_cast<int*>(smem + W * kWaveBytes - 32 * 8 - 16);  // below the last wave's sums: nothing lives there any more
      if (threadIdx.x == 0) {
        if (tr) tr[12] = __builtin_amdgcn_s_memrealtime();
        // one counter per part, 4 KB apart: device-scope atomics on ONE line retire at ~12 ns apiece (MI355X_MICROARCH.md, fanin), 1024 of them would be 12 us
        const unsigned long long seen = __hip_atomic_fetch_add(inl.arrive + (size_t)part * kArriveStride, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last = seen + 1 == inl.arrive_target[part];
        if (tr) tr[13] = __builtin_amdgcn_s_memrealtime();
        if (*last) *reinterpret_cast<unsigned long long*>(last + 2) = __builtin_amdgcn_s_memrealtime();
      }
      __syncthreads();
      if (*last) {
        const int row_begin = part * inl.rows_per_part, row_count = min(inl.rows_per_part, inl.num_rows - row_begin);
        if constexpr (MODE == MODE_ERR)
          finalize_part_error(partials, row_begin, row_count, reinterpret_cast<double*>(smem), inl.fin_out + (size_t)part * inl.fin_stride, inl.fin_flags + part, inl.fin_seq);
        else
          finalize_part_rows(partials, row_begin, row_count, reinterpret_cast<double*>(smem), inl.fin_out + (size_t)part * inl.fin_stride, inl.fin_flags + part, inl.fin_seq, tr,
                             *reinterpret_cast<const unsigned long long*>(last + 2));
      }
    }
  } else if (W == 4 && inl.arrive) {  // (wide workgroups exist for planned single-factor launches only: the host never arms this form for them)
    // fused finalize by FACTOR (synchronous batched calls, small single factors): the workgroup that stores a factor's last row sums the factor's rows and
    // expands them into the record -- rigid_slice_total / rigid_wave_tree / rigid_expand_wave in the order of vgicp_finalize_rigid_kernel<1024>, whose 32
    // slices of rows are taken four to a thread here -- and hands record and completion word to the host while the other factors' tiles are still running:
    // no finalize launch, and the records' way over PCIe (a third of a 512-factor call) is hidden behind the tile kernel
    if constexpr (MODE == MODE_ERR) {
      // the error evaluation's by-factor form (round 4): the last arriver adds the factor's rows up in the order of vgicp_finalize_error_kernel (error_factor_total)
      int* const last = reinterpret_cast<int*>(smem + 4 * kWaveBytes - 32 * 8 - 16);
      if (threadIdx.x == 0) {
        unsigned long long* ctr = inl.arrive + (size_t)factor_idx * kFactorArriveStride;
        const unsigned long long seen = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool l = seen + 1 == (unsigned long long)f.tile_count;
        if (l) __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last = l;
      }
      __syncthreads();
      if (*last) {
        const double total = error_factor_total<true>(partials, f.tile_begin, f.tile_count, reinterpret_cast<double*>(smem));
        if (threadIdx.x == 0) {
          asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(inl.fin_out + (size_t)factor_idx * inl.fin_stride), "v"(total) : "memory");
          asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(inl.fin_flags + factor_idx), "v"(inl.fin_seq) : "memory");
        }
      }
    }
    if constexpr (MODE == MODE_LIN) {
      int* const last = reinterpret_cast<int*>(smem + 4 * kWaveBytes - 32 * 8 - 16);
      if (threadIdx.x == 0) {
        unsigned long long* ctr = inl.arrive + (size_t)factor_idx * kFactorArriveStride;
        const unsigned long long seen = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool l = seen + 1 == (unsigned long long)f.tile_count;
        if (l) __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every arrival of this launch is in: ready for the next one
        *last = l;
      }
      __syncthreads();
      if (*last) {
        double* wsum16 = reinterpret_cast<double*>(smem);                  // [16][32]: what the sixteen waves of the finalize kernel would have left
        RigidScratch& S = *reinterpret_cast<RigidScratch*>(smem + 4096);   // (both below the waves' own sums, which nobody reads any more)
        static_assert(4096 + sizeof(RigidScratch) <= kWaveBytes - 32 * 8, "scratch of the fused finalize must fit in front of wave 0's sums");
        const int comp = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const double* base = partials + (size_t)f.tile_begin * ACC_STRIDE + comp;
        const double s0 = rigid_slice_total<32, true>(base, f.tile_count, 4 * sl), s1 = rigid_slice_total<32, true>(base, f.tile_count, 4 * sl + 1);
        const double s2 = rigid_slice_total<32, true>(base, f.tile_count, 4 * sl + 2), s3 = rigid_slice_total<32, true>(base, f.tile_count, 4 * sl + 3);
        wsum16[(2 * sl) * 32 + comp] = s0 + s1;
        wsum16[(2 * sl + 1) * 32 + comp] = s2 + s3;
        __syncthreads();
        if (threadIdx.x < 64) {
          const int l64 = threadIdx.x;
          if (l64 < 32) S.sum[l64] = rigid_wave_tree<16>(wsum16, l64);
          GP_WAVE_SYNC();
          rigid_expand_wave(S, Tl, l64);
          double* out_rec = inl.fin_out + (size_t)factor_idx * inl.fin_stride;
          const double d0 = S.dst[l64], d1 = l64 + 64 < 122 ? S.dst[l64 + 64] : 0.0;
          asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(out_rec + l64), "v"(d0) : "memory");
          if (l64 + 64 < 122) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(out_rec + l64 + 64), "v"(d1) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (l64 == 0) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(inl.fin_flags + factor_idx), "v"(inl.fin_seq) : "memory");
        }
      }
    }
  }
  GP_TRACE(7);
  if constexpr (TRACE) {
    if (trace && threadIdx.x == 0) trace[(size_t)tile_idx * 16 + 11] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace gp
