// gp_binning.hip -- see gp_binning.hpp
#include "gp_binning.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "gp_sort.hpp"

namespace gp {

namespace {

constexpr unsigned kNoCell = 0x7fffffffu;  // cell_of[] of a skipped (non-finite) point

__device__ __forceinline__ bool point_cell(const float* __restrict__ points, size_t i, double inv_cell, int& cx, int& cy, int& cz) {
  const float x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
  const double ux = (double)x * inv_cell, uy = (double)y * inv_cell, uz = (double)z * inv_cell;
  // finite and inside the range a 32-bit cell coordinate can hold (with room for the >> 2 block arithmetic)
  const bool ok = fabs(ux) < 1.0e9 && fabs(uy) < 1.0e9 && fabs(uz) < 1.0e9;  // false for NaN / inf
  cx = ok ? fast_floor(ux) : 0;
  cy = ok ? fast_floor(uy) : 0;
  cz = ok ? fast_floor(uz) : 0;
  return ok;
}

// bounding box (block units) of the finite points.  ONE kernel, reduced by the HOST: a workgroup reduces its tiles of 4096 points (the sixteen loads of a thread in
// flight together; at most HostSlots::kSlots workgroups, which stride over the tiles) and leaves {min xyz, max xyz, -, seq} in its 32-byte slot of host-mapped memory,
// the sequence number stored behind the box; the host combines the slots as they arrive.  A workgroup without a finite point leaves the neutral box.
// (Tried before: the boxes folded into six device words with atomicMax and the last workgroup by ticket writing the result -- 488 atomics on one address are served one
// after the other, 16 us; then per-workgroup boxes + a one-workgroup reduce kernel, 10 us of which ~6 are the second kernel's boundary.)
constexpr int kBboxTile = 4096;
__global__ void __launch_bounds__(256) bins_bbox_kernel(const float* __restrict__ points, int n, double inv_cell, int* __restrict__ slots /* host-mapped */, int seq,
                                                        const FillJob zero_states) {
  run_fill_job(zero_states);  // (the state words of the kernels behind: gp_host.hpp, FillJob)
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  const size_t tiles = ((size_t)n + kBboxTile - 1) / kBboxTile;
  for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const size_t base = tile * kBboxTile;
    float p[kBboxTile / 256][3];
#pragma unroll
    for (int r = 0; r < kBboxTile / 256; r++) {
      const size_t i = min(base + (size_t)r * 256 + threadIdx.x, (size_t)n - 1);  // (a point seen twice does not change the box)
#pragma unroll
      for (int a = 0; a < 3; a++) p[r][a] = points[3 * i + a];
    }
#pragma unroll
    for (int r = 0; r < kBboxTile / 256; r++) {
      const double ux = (double)p[r][0] * inv_cell, uy = (double)p[r][1] * inv_cell, uz = (double)p[r][2] * inv_cell;
      const bool ok = fabs(ux) < 1.0e9 && fabs(uy) < 1.0e9 && fabs(uz) < 1.0e9;  // (point_cell's rule)
      const int c[3] = {fast_floor(ux) >> 2, fast_floor(uy) >> 2, fast_floor(uz) >> 2};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        lo[a] = ok ? min(lo[a], c[a]) : lo[a];
        hi[a] = ok ? max(hi[a], c[a]) : hi[a];
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = max(hi[a], __shfl_xor(hi[a], off, 64));
    }
  __shared__ int wave_box[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      wave_box[threadIdx.x >> 6][a] = lo[a];
      wave_box[threadIdx.x >> 6][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int v[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
      v[a] = wave_box[0][a];
#pragma unroll
      for (int w = 1; w < 4; w++) v[a] = a < 3 ? min(v[a], wave_box[w][a]) : max(v[a], wave_box[w][a]);
    }
    int4* slot = reinterpret_cast<int4*>(slots + 8 * (size_t)blockIdx.x);
    slot[0] = make_int4(v[0], v[1], v[2], v[3]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first half is in host memory before the half that carries the sequence number
    slot[1] = make_int4(v[4], v[5], 0, seq);
  }
}

// sort key of a point: (block index, bit inside the block) -- the order the cells are numbered in.  < 2^24 * 64 = 2^30.  The kernel also counts the keys' digits for
// every pass of the sort behind it (gp_sort.hpp: the sort needs no pass over the keys for that).
constexpr int kKeyTile = 4096;
template <int passes>  // (compile-time: with a run-time pass count the digit counts compile to a loop nest with a branch per count)
__global__ void __launch_bounds__(256) bins_key_kernel(const float* __restrict__ points, int n, double inv_cell, GridGeom g, unsigned* __restrict__ keys,
                                                       unsigned* __restrict__ hist, unsigned invalid_key, const FillJob zero_blocks) {
  run_fill_job(zero_blocks);  // (the block grid the cells kernel ORs its occupancy bits into)
  __shared__ SortHistLds l;
  GP_SORT_STAMP(blockIdx.x, 8);
  sort_hist_clear(l);
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kKeyTile;
  // two halves of eight points: the loads of a half are in flight together
#pragma unroll
  for (int half = 0; half < 2; half++) {
    float p[8][3];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const size_t i = min(base + (size_t)(half * 8 + r) * 256 + threadIdx.x, (size_t)n - 1);
#pragma unroll
      for (int a = 0; a < 3; a++) p[r][a] = points[3 * i + a];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const size_t i = base + (size_t)(half * 8 + r) * 256 + threadIdx.x;
      const double ux = (double)p[r][0] * inv_cell, uy = (double)p[r][1] * inv_cell, uz = (double)p[r][2] * inv_cell;
      const bool ok = fabs(ux) < 1.0e9 && fabs(uy) < 1.0e9 && fabs(uz) < 1.0e9;  // (point_cell's rule)
      const int cx = fast_floor(ux), cy = fast_floor(uy), cz = fast_floor(uz);
      const unsigned key = ok ? (unsigned)(grid_block_index(g, cx, cy, cz) * 64 + grid_bit(cx, cy, cz)) : invalid_key;
      if (i < (size_t)n) {
        keys[i] = key;
        sort_hist_count(l, key, passes);
      }
    }
  }
  GP_SORT_STAMP(blockIdx.x, 9);  // this wave's points done
  __syncthreads();
  GP_SORT_STAMP(blockIdx.x, 10);
  sort_hist_flush(l, passes, hist);
  GP_SORT_STAMP(blockIdx.x, 11);  // flush issued
}

// Cells = runs of equal keys in the sorted order.  TWO kernels (round 4; was: scan of the cell-start flags, a kernel writing the cells, scan of the block-start
// flags, a kernel writing the block list -- 61 us per 2 M points, with an 8 MB array of scanned flags written and read twice between them).  A workgroup takes a tile
// of 4096 sorted keys; bins_count_kernel counts the cells AND the blocks that start in it (one packed sum: cells in the low 31 bits, blocks above), stores the
// tile's word and adds it to its group's word (32 tiles per group).  bins_cells_kernel forms the flags again, takes its prefix = the groups in front + the tiles in
// front inside its own group (one entry per lane of the first wave), and writes, per point, the ordinal of its cell; at the first point of a cell: the cell's
// start, its block and its occupancy bit; at the first cell of a block: the block's base and its entry in the compact list of occupied blocks.
// The thread that sees the end of the binned points (the first skipped point, or the last point) knows all three counts -- binned points, cells, blocks -- and stores
// them for the host (host_counts[8..10], host-mapped words).
// (One kernel with the tiles waiting for their predecessors' words was tried first: the wait directly follows the publication, so every tile waits for the slowest
// tile in front of it -- 16 us median for the 488 tiles of 2 M points, of a 33 us kernel: scripts/probe/bins_probe.hip.  The sort's passes hide the same wait behind
// their ranking; here there is nothing to hide it behind, and the second read of the keys comes from the cache.)
// state (64-bit words, zeroed): group words [groups], behind them tile words [tiles].
constexpr int kCellsTile = 4096, kCellsPerThread = 16, kCellsGroup = 32;
inline size_t cells_tiles(long long n) { return (size_t)((n + kCellsTile - 1) / kCellsTile); }
inline size_t cells_groups(long long n) { return (cells_tiles(n) + kCellsGroup - 1) / kCellsGroup; }
inline size_t cells_state_words(long long n) { return cells_groups(n) + cells_tiles(n); }

// a thread's 16 consecutive sorted keys and their flags: bit k of cflag = a cell starts at element k, of bflag = a block starts there
struct CellKeys {
  unsigned key[kCellsPerThread];
  unsigned prev0, cflag, bflag;
  bool full, any, has_prev;
};
__device__ __forceinline__ void load_cell_keys(const unsigned* __restrict__ sorted_keys, int n, long long base, unsigned invalid_key, CellKeys& c) {
  c.full = base + kCellsPerThread <= (long long)n;
  if (c.full && (reinterpret_cast<uintptr_t>(sorted_keys) & 15) == 0) {
    const uint4* p = reinterpret_cast<const uint4*>(sorted_keys + base);
#pragma unroll
    for (int q = 0; q < kCellsPerThread / 4; q++) {
      const uint4 x = p[q];
      c.key[4 * q] = x.x, c.key[4 * q + 1] = x.y, c.key[4 * q + 2] = x.z, c.key[4 * q + 3] = x.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kCellsPerThread; k++) c.key[k] = base + k < (long long)n ? sorted_keys[base + k] : invalid_key;
  }
  c.any = base < (long long)n;
  c.has_prev = c.any && base > 0;
  c.prev0 = c.has_prev ? sorted_keys[base - 1] : 0u;
  c.cflag = 0, c.bflag = 0;
  unsigned prev = c.prev0;
  bool have = c.has_prev;
#pragma unroll
  for (int k = 0; k < kCellsPerThread; k++) {
    const bool in = base + k < (long long)n && c.key[k] != invalid_key;
    if (in && (!have || prev != c.key[k])) c.cflag |= 1u << k;
    if (in && (!have || (prev >> 6) != (c.key[k] >> 6))) c.bflag |= 1u << k;
    prev = c.key[k];
    have = true;
  }
}

__global__ void __launch_bounds__(256) bins_count_kernel(const unsigned* __restrict__ sorted_keys, int n, unsigned invalid_key, unsigned long long* __restrict__ state, int num_groups) {
  __shared__ unsigned long long wave_sum[4];
  const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  CellKeys c;
  load_cell_keys(sorted_keys, n, (long long)tile * kCellsTile + (long long)threadIdx.x * kCellsPerThread, invalid_key, c);
  unsigned long long sum = (unsigned long long)__popc(c.cflag) | ((unsigned long long)__popc(c.bflag) << 31);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) wave_sum[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    state[num_groups + tile] = total;
    if (total) atomicAdd(state + tile / kCellsGroup, total);  // (integer sum of <= 32 tiles: the order does not matter)
  }
}

__global__ void __launch_bounds__(256) bins_cells_kernel(const unsigned* __restrict__ sorted_keys, int n, unsigned invalid_key, const unsigned long long* __restrict__ state, int num_groups,
                                                         GridBlock* __restrict__ blocks, int* __restrict__ cell_start, unsigned* __restrict__ cell_of, int* __restrict__ cell_block,
                                                         int* __restrict__ occ_blocks, int* __restrict__ host_counts /* host-mapped */, int seq,
                                                         const unsigned* __restrict__ sort_state, unsigned sort_pass_words, int sort_passes) {
  __shared__ unsigned long long wave_sum[4];
  __shared__ unsigned long long tile_prefix;
  const int tile = blockIdx.x;
  // A sort pass that gave up a wait (gp_sort.hpp, draw_tile) voids this build: its output has holes -- whatever the pooled buffer held -- and NOTHING may be indexed with
  // such keys (`blocks[kk >> 6]`, `occ_blocks[bo]`, `cell_start[ord]` below were, out of bounds, before the host had seen the fault).  Every tile asks first
  // (the passes are complete: same answer everywhere, scalar loads) and leaves; tile 0 tells the host, which builds again through the one-class sort.
  if (radix_sort_faults(sort_state, sort_pass_words, sort_passes)) {
    if (tile == 0 && threadIdx.x == 0) {
      host_counts[11] = 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      host_counts[HostWords::kFlag] = seq;
    }
    return;
  }
  GP_SORT_STAMP(tile, 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)tile * kCellsTile + (long long)threadIdx.x * kCellsPerThread;
  // the tile's prefix: the groups in front, then the tiles in front inside the own group (asked for first: the answer arrives with the keys)
  unsigned long long prefix = 0;
  if (wave == 0) {
    const int group = tile / kCellsGroup, in_group = tile % kCellsGroup;
    const int entries = group + in_group;
    for (int e0 = 0; e0 < entries; e0 += 64) {
      const int e = e0 + lane;
      unsigned long long w = 0;
      if (e < entries) w = e < group ? state[e] : state[num_groups + (size_t)group * kCellsGroup + (e - group)];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
      prefix += w;
    }
  }
  CellKeys ck;
  load_cell_keys(sorted_keys, n, base, invalid_key, ck);
  unsigned (&key)[kCellsPerThread] = ck.key;
  const unsigned cflag = ck.cflag, bflag = ck.bflag, prev0 = ck.prev0;
  const bool full = ck.full, any = ck.any, has_prev = ck.has_prev;
  const unsigned long long mine = (unsigned long long)__popc(cflag) | ((unsigned long long)__popc(bflag) << 31);
  unsigned long long incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  if (threadIdx.x == 0) tile_prefix = prefix;
  __syncthreads();
  GP_SORT_STAMP(tile, 1);  // keys loaded, flags counted, prefix known
  unsigned long long wave_excl = 0;
#pragma unroll
  for (int w = 0; w < 4; w++)
    if (w < wave) wave_excl += wave_sum[w];
  GP_SORT_STAMP(tile, 2);
  if (!any) return;
  const unsigned long long excl = tile_prefix + wave_excl + incl - mine;
  const int cells0 = (int)(excl & 0x7fffffffull), blocks0 = (int)(excl >> 31);  // cells / blocks that start in front of this thread's first element
  // the ordinal of every element's cell: cells that start at or before it, minus one (straight-line: no branch per element)
  unsigned ord_out[kCellsPerThread];
#pragma unroll
  for (int k = 0; k < kCellsPerThread; k++) ord_out[k] = key[k] == invalid_key ? kNoCell : (unsigned)(cells0 + __popc(cflag & ((2u << k) - 1u)) - 1);
  if (full && (reinterpret_cast<uintptr_t>(cell_of) & 15) == 0) {
    uint4* p = reinterpret_cast<uint4*>(cell_of + base);
#pragma unroll
    for (int q = 0; q < kCellsPerThread / 4; q++) p[q] = make_uint4(ord_out[4 * q], ord_out[4 * q + 1], ord_out[4 * q + 2], ord_out[4 * q + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < kCellsPerThread; k++)
      if (base + k < (long long)n) cell_of[base + k] = ord_out[k];
  }
  GP_SORT_STAMP(tile, 3);  // ordinals stored
  // the cells that start here: one trip per set flag (a voxel map has a cell start every ~27 points, a search grid every ~4), not one per element
  for (unsigned rest = cflag; rest;) {
    const int k = __ffs(rest) - 1;
    rest &= rest - 1u;
    unsigned kk = key[0];
#pragma unroll
    for (int q = 1; q < kCellsPerThread; q++) kk = k == q ? key[q] : kk;  // (register array: select, no indexing)
    const int ord = cells0 + __popc(cflag & ((1u << k) - 1u));
    cell_start[ord] = (int)(base + k);
    cell_block[ord] = (int)(kk >> 6);
    atomicOr(&blocks[kk >> 6].bits, 1ull << (kk & 63u));  // one atomic per CELL (integer: the result does not depend on the order)
    if ((bflag >> k) & 1u) {
      const int bo = blocks0 + __popc(bflag & ((1u << k) - 1u));
      blocks[kk >> 6].base = ord;
      occ_blocks[bo] = (int)(kk >> 6);
    }
  }
  GP_SORT_STAMP(tile, 4);  // cells written
  // the end of the binned points: the first skipped point (they sort behind every cell), or the last point
  const int cells_end = cells0 + __popc(cflag), blocks_end = blocks0 + __popc(bflag);
  int end_at = -1;
  {
    bool prev_binned = has_prev ? prev0 != invalid_key : true;  // (element 0 of a cloud without a single binned point closes it at 0)
#pragma unroll
    for (int k = 0; k < kCellsPerThread; k++) {
      const bool in_range = base + k < (long long)n;
      if (in_range && key[k] == invalid_key && prev_binned && end_at < 0) end_at = (int)(base + k);
      if (in_range && base + k == (long long)n - 1 && key[k] != invalid_key) end_at = n;
      prev_binned = key[k] != invalid_key;
    }
  }
  if (end_at >= 0) {
    cell_start[cells_end] = end_at;
    host_counts[8] = end_at, host_counts[9] = cells_end, host_counts[10] = blocks_end;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counts are in host memory before the flag is stored (HostWords::wait_flag)
    host_counts[HostWords::kFlag] = seq;
  }
}

}  // namespace

namespace {
thread_local int g_inject_sort_faults = 0;  // test hook: the next so many builds of this thread see a faulted sort (gp_debug_inject_sort_fault)
thread_local bool g_inject_corrupt = false;  // ... whose first tile also leaves garbage keys behind (count < 0)
thread_local int g_sort_fallbacks = 0;      // builds of this thread that went through the one-class sort
int bin_points_once(const float* points_dev, int n, double inv_cell, hipStream_t s, PointBins* bins, bool* too_large, int ticket_classes, bool* sort_fault);
}  // namespace

void inject_sort_faults(int count) {
  g_inject_sort_faults = count < 0 ? -count : count;
  g_inject_corrupt = count < 0;
}
int sort_fallbacks() { return g_sort_fallbacks; }

int bin_points(const float* points_dev, int n, double inv_cell, hipStream_t s, PointBins* bins, bool* too_large) {
  bool fault = false;
  int classes = kSortTicketClasses;
  if (g_inject_sort_faults > 0) {
    g_inject_sort_faults--;
    classes = -kSortTicketClasses - (g_inject_corrupt ? kSortCorruptHook : 0);
  }
  GP_TRY(bin_points_once(points_dev, n, inv_cell, s, bins, too_large, classes, &fault));
  if (!fault) return GP_OK;
  // A tile of the sort waited for a predecessor that had not even been started (the device did not start workgroups in blockIdx order: gp_sort.hpp). Everything the
  // build derived from the keys is void; build again with the single ticket counter, which needs no such order.
  g_sort_fallbacks++;
  GP_HIP(hipStreamSynchronize(s));
  GP_TRY(bin_points_once(points_dev, n, inv_cell, s, bins, too_large, 1, &fault));
  if (fault) return fail(GP_ERROR_HIP, "bin_points: the radix sort made no progress (one-class form)");
  return GP_OK;
}

namespace {
int bin_points_once(const float* points_dev, int n, double inv_cell, hipStream_t s, PointBins* bins, bool* too_large, int ticket_classes, bool* sort_fault) {
  *sort_fault = false;
  *too_large = false;
  bins->num_cells = 0;
  bins->num_binned = 0;
  bins->num_blocks = 0;
  if (n <= 0) {
    GP_TRY(bins->cell_start.alloc_pooled(sizeof(int), s));
    GP_HIP(hipMemsetAsync(bins->cell_start.ptr, 0, sizeof(int), s));
    return GP_OK;
  }
  if (n >= (1 << 30)) return fail(GP_ERROR_INVALID_ARGUMENT, "bin_points: at most 2^30 - 1 points");
  // ---- the state words of every kernel of this build (per sort pass -- sized for the widest key --, the histograms, the cells kernel's): ONE fill ----
  DeviceArray states;
  HostWords hw;  // host-mapped: [0..5] bounding box, [8] binned points, [9] cells, [10] occupied blocks -- written by the kernels, read behind the synchronisations
  GP_TRY(HostWords::get(&hw));
  const size_t sort_words = radix_sort_state_words32(n, 32) /* 32-bit */, cells_words = cells_state_words(n) /* 64-bit */;
  const size_t sort_off = 0, cells_off = (sort_off + sort_words * 4 + 7) & ~size_t(7), state_bytes = (cells_off + cells_words * 8 + 255) & ~size_t(255);  // (a fill whose size is not a multiple of 16 B is two kernels)
  GP_TRY(states.alloc_async(state_bytes, s));
  char* st = states.as<char>();  // (zeroed by the bounding-box kernel on its way)
  // ---- bounding box ----
  HostSlots slots;
  GP_TRY(HostSlots::get(&slots));
  const int box_wgs = (int)std::min<size_t>(((size_t)n + kBboxTile - 1) / kBboxTile, (size_t)HostSlots::kSlots);
  const int seq_box = hw.next_seq();
  hipLaunchKernelGGL(bins_bbox_kernel, dim3(box_wgs), dim3(256), 0, s, points_dev, n, inv_cell, slots.dev, seq_box, fill_job(states.ptr, state_bytes, 0u));
  GP_HIP(hipGetLastError());
  // (what does not depend on the box is allocated while the device works on it)
  DeviceArray keys_b, vals_b;
  GP_TRY(bins->cell_of.alloc_pooled(sizeof(unsigned) * (size_t)n, s));
  GP_TRY(bins->order.alloc_pooled(sizeof(int) * (size_t)n, s));
  GP_TRY(keys_b.alloc_pooled(sizeof(unsigned) * (size_t)n, s));
  GP_TRY(vals_b.alloc_pooled(sizeof(int) * (size_t)n, s));
  GP_TRY(bins->cell_start.alloc_pooled(sizeof(int) * ((size_t)n + 1), s));
  GP_TRY(bins->cell_block.alloc_pooled(sizeof(int) * (size_t)n, s));
  GP_TRY(bins->occ_blocks.alloc_pooled(sizeof(int) * (size_t)n, s));  // at most one block per cell
  // the host combines the workgroups' boxes as they arrive (bounded spin, then the stream -- which also surfaces a failed kernel)
  int h_bbox[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  {
    const volatile int* hs = slots.host;
    const auto t_wait = std::chrono::steady_clock::now();
    bool synced = false;
    for (int w = 0; w < box_wgs; w++) {
      for (int spins = 0; hs[8 * w + 7] != seq_box; spins++) {
        if ((spins & 63) == 63 && !synced && std::chrono::steady_clock::now() - t_wait > std::chrono::microseconds(500)) {
          GP_HIP(hipStreamSynchronize(s));
          synced = true;
        } else if (synced && hs[8 * w + 7] != seq_box) {
          return fail(GP_ERROR_HIP, "bin_points: the bounding-box kernel finished without leaving its boxes");
        }
      }
      for (int a = 0; a < 3; a++) {
        h_bbox[a] = std::min(h_bbox[a], (int)hs[8 * w + a]);
        h_bbox[3 + a] = std::max(h_bbox[3 + a], (int)hs[8 * w + 3 + a]);
      }
    }
  }
  if (h_bbox[0] > h_bbox[3]) {  // no finite point at all
    GP_TRY(bins->cell_start.alloc_pooled(sizeof(int), s));
    GP_HIP(hipMemsetAsync(bins->cell_start.ptr, 0, sizeof(int), s));
    return GP_OK;
  }
  double nb = 1.0;
  for (int a = 0; a < 3; a++) {
    bins->geom.lo[a] = h_bbox[a];
    const long long d = (long long)h_bbox[3 + a] - (long long)h_bbox[a] + 1;
    bins->geom.dim[a] = (int)std::min<long long>(d, 1ll << 30);
    nb *= (double)d;
  }
  if (nb > (double)kMaxGridBlocks) {
    *too_large = true;
    return GP_OK;
  }
  bins->num_blocks = (long long)bins->geom.dim[0] * bins->geom.dim[1] * bins->geom.dim[2];
  // ---- keys = (block, bit), stable sort, cells = runs of equal keys ----
  GP_TRY(bins->blocks.alloc_pooled(sizeof(GridBlock) * (size_t)bins->num_blocks, s));  // (zeroed by the key kernel on its way)
  // keys of the binned points are below K = blocks * 64; skipped points carry the all-ones key of the narrowest width that exceeds them -- 2^b - 1 >= K -- so they
  // land behind every cell, and the sort runs over b bits (a search grid of 1 M points has K ~ 1.2e7: 24 bits, three passes; one bit more for the marker was a fourth)
  int key_bits = 7;
  while ((1ll << key_bits) - 1 < bins->num_blocks * 64) key_bits++;
  const unsigned invalid_key = (unsigned)((1ll << key_bits) - 1);
  bool in_b = false;
  unsigned* sort_state = reinterpret_cast<unsigned*>(st + sort_off);
  {
    const dim3 kgrid((n + kKeyTile - 1) / kKeyTile), kblock(256);
    unsigned* hist = radix_sort_hist(sort_state, n, key_bits);
    const FillJob zero_blocks = fill_job(bins->blocks.ptr, sizeof(GridBlock) * (size_t)bins->num_blocks, 0u);
    switch ((key_bits + 7) / 8) {
      case 1: hipLaunchKernelGGL(bins_key_kernel<1>, kgrid, kblock, 0, s, points_dev, n, inv_cell, bins->geom, bins->cell_of.as<unsigned>(), hist, invalid_key, zero_blocks); break;
      case 2: hipLaunchKernelGGL(bins_key_kernel<2>, kgrid, kblock, 0, s, points_dev, n, inv_cell, bins->geom, bins->cell_of.as<unsigned>(), hist, invalid_key, zero_blocks); break;
      case 3: hipLaunchKernelGGL(bins_key_kernel<3>, kgrid, kblock, 0, s, points_dev, n, inv_cell, bins->geom, bins->cell_of.as<unsigned>(), hist, invalid_key, zero_blocks); break;
      default: hipLaunchKernelGGL(bins_key_kernel<4>, kgrid, kblock, 0, s, points_dev, n, inv_cell, bins->geom, bins->cell_of.as<unsigned>(), hist, invalid_key, zero_blocks); break;
    }
  }
  GP_HIP(hipGetLastError());
  GP_TRY(radix_sort_pairs(bins->cell_of.as<unsigned>(), bins->order.as<int>(), keys_b.as<unsigned>(), vals_b.as<int>(), n, key_bits, true, sort_state, true, true, s, &in_b, ticket_classes));
  if (in_b) {
    bins->cell_of.swap(keys_b);
    bins->order.swap(vals_b);
  }
  // keys_b is free now: it receives the sorted points' cell ordinals while cell_of still holds the sorted keys
  // (no host round trip in the middle: the arrays the cell count would size are allocated for the worst case, one cell per point; the host learns cells, occupied
  // blocks and binned points together at the end)
  const int seq_cells = hw.next_seq();
  reinterpret_cast<volatile int*>(hw.host)[11] = 0;  // (the kernel only ever stores a 1 there)
  hipLaunchKernelGGL(bins_count_kernel, dim3((unsigned)cells_tiles(n)), dim3(256), 0, s, (const unsigned*)bins->cell_of.as<unsigned>(), n, invalid_key,
                     reinterpret_cast<unsigned long long*>(st + cells_off), (int)cells_groups(n));
  hipLaunchKernelGGL(bins_cells_kernel, dim3((unsigned)cells_tiles(n)), dim3(256), 0, s, (const unsigned*)bins->cell_of.as<unsigned>(), n, invalid_key,
                     (const unsigned long long*)reinterpret_cast<unsigned long long*>(st + cells_off), (int)cells_groups(n), bins->blocks.as<GridBlock>(),
                     bins->cell_start.as<int>(), keys_b.as<unsigned>(), bins->cell_block.as<int>(), bins->occ_blocks.as<int>(), hw.dev, seq_cells,
                     (const unsigned*)sort_state, (unsigned)radix_sort_pass_words(n), (key_bits + 7) / 8);
  GP_HIP(hipGetLastError());
  bins->cell_of.swap(keys_b);  // cell_of = ordinals of the sorted points
  // the counts are awaited, not the kernels: what the caller issues behind this is ordered by the stream, and the scratch arrays go back to the pool in ITS order
  GP_TRY(hw.wait_flag(seq_cells, s));
  keys_b.release_on(s);
  vals_b.release_on(s);
  states.release_on(s);
  if (reinterpret_cast<volatile int*>(hw.host)[11]) {
    *sort_fault = true;
    return GP_OK;
  }
  const int h_counts[3] = {reinterpret_cast<volatile int*>(hw.host)[9], reinterpret_cast<volatile int*>(hw.host)[10], reinterpret_cast<volatile int*>(hw.host)[8]};  // cells, occupied blocks, binned points
  bins->num_cells = h_counts[0];
  bins->num_occ_blocks = h_counts[0] > 0 ? h_counts[1] : 0;
  bins->num_binned = h_counts[2];
  return GP_OK;
}
}  // namespace

}  // namespace gp

extern "C" {
int gp_debug_inject_sort_fault(int count) {
  gp::inject_sort_faults(count);
  return GP_OK;
}
int gp_debug_sort_fallbacks(void) { return gp::sort_fallbacks(); }
}
