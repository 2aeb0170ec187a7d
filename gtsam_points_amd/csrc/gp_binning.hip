// gp_binning.hip -- see gp_binning.hpp
#include "gp_binning.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "gp_sort.hpp"

namespace gp {

namespace {

constexpr unsigned kInvalidKey = 0x7fffffffu;

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

// per-workgroup bounding box (block units) of the finite points: boxes[wg][6] = {min xyz, max xyz}; a workgroup without a finite
// point writes the neutral box
__global__ void __launch_bounds__(256) bins_bbox_kernel(const float* __restrict__ points, int n, double inv_cell, int* __restrict__ boxes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  if (i < (size_t)n) {
    int c[3];
    if (point_cell(points, i, inv_cell, c[0], c[1], c[2])) {
#pragma unroll
      for (int a = 0; a < 3; a++) lo[a] = hi[a] = c[a] >> 2;
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
  if (threadIdx.x < 6) {
    int v = wave_box[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? min(v, wave_box[w][threadIdx.x]) : max(v, wave_box[w][threadIdx.x]);
    boxes[6 * (size_t)blockIdx.x + threadIdx.x] = v;
  }
}

__global__ void __launch_bounds__(256) bins_bbox_reduce_kernel(const int* __restrict__ boxes, int nb, int* __restrict__ bbox) {
  __shared__ int part[256][6];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int b = threadIdx.x; b < nb; b += 256)
    for (int a = 0; a < 3; a++) {
      lo[a] = min(lo[a], boxes[6 * (size_t)b + a]);
      hi[a] = max(hi[a], boxes[6 * (size_t)b + 3 + a]);
    }
  for (int a = 0; a < 3; a++) {
    part[threadIdx.x][a] = lo[a];
    part[threadIdx.x][3 + a] = hi[a];
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int a = 0; a < 6; a++) part[threadIdx.x][a] = a < 3 ? min(part[threadIdx.x][a], part[threadIdx.x + w][a]) : max(part[threadIdx.x][a], part[threadIdx.x + w][a]);
    __syncthreads();
  }
  if (threadIdx.x < 6) bbox[threadIdx.x] = part[0][threadIdx.x];
}

// sort key of a point: (block index, bit inside the block) -- the order the cells are numbered in.  < 2^24 * 64 = 2^30
__global__ void __launch_bounds__(256) bins_key_kernel(const float* __restrict__ points, int n, double inv_cell, GridGeom g, unsigned* __restrict__ keys) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n) return;
  int cx, cy, cz;
  unsigned key = kInvalidKey;
  if (point_cell(points, i, inv_cell, cx, cy, cz)) key = (unsigned)(grid_block_index(g, cx, cy, cz) * 64 + grid_bit(cx, cy, cz));
  keys[i] = key;
}

// sorted keys -> 1 at the first point of every cell (0 elsewhere and on the skipped points behind the cells).  Round 4: the flags are not stored -- the scan and the
// kernels behind it evaluate them where they need them (one kernel, one 8 MB array and its read-back less per build)
struct CellStartFlag {
  const unsigned* sorted_keys;
  __device__ __forceinline__ int operator()(long long j) const {
    const unsigned k = sorted_keys[j];
    return (k != kInvalidKey && (j == 0 || sorted_keys[j - 1] != k)) ? 1 : 0;
  }
};
// cell c opens a block when it is the block's first cell (entries behind the last cell: 0)
struct BlockStartFlag {
  const int* num_cells;
  const int* cell_block;
  const GridBlock* blocks;
  __device__ __forceinline__ int operator()(long long c) const { return (c < *num_cells && blocks[cell_block[c]].base == (int)c) ? 1 : 0; }
};

// ordinal of every sorted point's cell (exclusive scan of the flags, + its own flag, - 1); at the first point of a cell: the cell's
// start, its occupancy bit, and -- at the first cell of a block -- the block's base.  total[0] = number of cells.
__global__ void __launch_bounds__(256) bins_finish_kernel(const unsigned* __restrict__ sorted_keys, const int* __restrict__ scanned, int n,
                                                          const int* __restrict__ total, GridBlock* __restrict__ blocks, int* __restrict__ cell_start,
                                                          unsigned* __restrict__ cell_of, int* __restrict__ cell_block, int* __restrict__ num_binned,
                                                          int* __restrict__ host_counts /* host-mapped: [8] binned points, [9] cells */) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= (size_t)n) return;
  const unsigned k = sorted_keys[j];
  const int num_cells = *total;
  if (j == 0) {
    num_binned[1] = num_cells;  // (kept for the kernels behind: the scan's slot is re-used)
    host_counts[9] = num_cells;
  }
  if (k == kInvalidKey) {
    cell_of[j] = kInvalidKey;
    if (j == 0 || sorted_keys[j - 1] != kInvalidKey) {
      cell_start[num_cells] = (int)j;
      *num_binned = (int)j;
      host_counts[8] = (int)j;
    }
    return;
  }
  const int flag = (j == 0 || sorted_keys[j - 1] != k) ? 1 : 0;  // (CellStartFlag)
  const int ord = scanned[j] + flag - 1;
  cell_of[j] = (unsigned)ord;
  if (flag) {
    cell_start[ord] = (int)j;
    cell_block[ord] = (int)(k >> 6);
    GridBlock* blk = blocks + (k >> 6);
    atomicOr(&blk->bits, 1ull << (k & 63u));  // one atomic per CELL (not per point); the bits of a block come from <= 64 cells
    if (j == 0 || (sorted_keys[j - 1] >> 6) != (k >> 6)) blk->base = ord;
  }
  if (j == (size_t)n - 1) {
    cell_start[num_cells] = n;
    *num_binned = n;
    host_counts[8] = n;
  }
}

// occupied blocks as a compact ascending list: cell c opens a block when it is the block's first cell
// occupied blocks as a compact ascending list (launched over all n positions: the cell count is still on the device)
__global__ void __launch_bounds__(256) bins_block_list_kernel(const int* __restrict__ num_cells, const int* __restrict__ cell_block, const GridBlock* __restrict__ blocks,
                                                              const int* __restrict__ scanned, int* __restrict__ occ_blocks, const int* __restrict__ scan_total,
                                                              int* __restrict__ num_occ_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0) *num_occ_out = *scan_total;
  if (c < *num_cells && blocks[cell_block[c]].base == c) occ_blocks[scanned[c]] = cell_block[c];
}

}  // namespace

int bin_points(const float* points_dev, int n, double inv_cell, hipStream_t s, PointBins* bins, bool* too_large) {
  *too_large = false;
  bins->num_cells = 0;
  bins->num_binned = 0;
  bins->num_blocks = 0;
  if (n <= 0) {
    GP_TRY(bins->cell_start.alloc_pooled(sizeof(int), s));
    GP_HIP(hipMemsetAsync(bins->cell_start.ptr, 0, sizeof(int), s));
    return GP_OK;
  }
  const int wgs = (n + 255) / 256;
  const bool dbg = getenv("GP_KNN_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  // ---- bounding box ----
  DeviceArray boxes, d_small;  // d_small (device): [8] binned points, [9] cells -- what the kernels behind read (a host-mapped word would be a PCIe read per wave)
  HostWords hw;  // host-mapped: [0..5] bounding box, [8] binned points, [9] cells, [10] occupied blocks -- written by the kernels, read behind the synchronisations
  GP_TRY(HostWords::get(&hw));
  GP_TRY(boxes.alloc_async(sizeof(int) * 6 * (size_t)wgs, s));
  GP_TRY(d_small.alloc_async(sizeof(int) * 16, s));
  hipLaunchKernelGGL(bins_bbox_kernel, dim3(wgs), dim3(256), 0, s, points_dev, n, inv_cell, boxes.as<int>());
  hipLaunchKernelGGL(bins_bbox_reduce_kernel, dim3(1), dim3(256), 0, s, (const int*)boxes.as<int>(), wgs, hw.dev);
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize(s));
  int h_bbox[6];
  for (int a = 0; a < 6; a++) h_bbox[a] = reinterpret_cast<volatile int*>(hw.host)[a];
  const double t1 = now();
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
  DeviceArray keys_b, vals_b, sort_scratch, scanned, scan_scratch, states;
  GP_TRY(bins->blocks.alloc_pooled(sizeof(GridBlock) * (size_t)bins->num_blocks, s));
  GP_HIP(hipMemsetAsync(bins->blocks.ptr, 0, sizeof(GridBlock) * (size_t)bins->num_blocks, s));
  GP_TRY(bins->cell_of.alloc_pooled(sizeof(unsigned) * (size_t)n, s));
  GP_TRY(bins->order.alloc_pooled(sizeof(int) * (size_t)n, s));
  GP_TRY(keys_b.alloc_pooled(sizeof(unsigned) * (size_t)n, s));
  GP_TRY(vals_b.alloc_pooled(sizeof(int) * (size_t)n, s));
  GP_TRY(sort_scratch.alloc_async(sizeof(int) * radix_sort_scratch_ints(n), s));
  GP_TRY(scanned.alloc_async(sizeof(int) * (size_t)n, s));
  GP_TRY(scan_scratch.alloc_async(sizeof(int) * scan_scratch_ints(n), s));
  hipLaunchKernelGGL(bins_key_kernel, dim3(wgs), dim3(256), 0, s, points_dev, n, inv_cell, bins->geom, bins->cell_of.as<unsigned>());
  GP_HIP(hipGetLastError());
  int bits = 6;
  while ((1ll << bits) < bins->num_blocks * 64) bits++;
  // skipped points carry kInvalidKey = 0x7fffffff: every pass sees all-ones digits, so they land behind every cell
  bool in_b = false;
  // the look-back states of every scan of this build (one per sort pass, two over the sorted points): ONE fill
  const int key_bits = std::min(bits + 1, 31);
  const size_t sort_words = radix_sort_state_words(n, key_bits), scan_words = onepass_state_words(n);
  GP_TRY(states.alloc_async(sizeof(unsigned long long) * (sort_words + 2 * scan_words), s));
  GP_HIP(hipMemsetAsync(states.ptr, 0, sizeof(unsigned long long) * (sort_words + 2 * scan_words), s));
  unsigned long long* st = states.as<unsigned long long>();
  GP_TRY(radix_sort_pairs(bins->cell_of.as<unsigned>(), bins->order.as<int>(), keys_b.as<unsigned>(), vals_b.as<int>(), n, key_bits, true, sort_scratch.as<int>(), s, &in_b, st));
  if (in_b) {
    bins->cell_of.swap(keys_b);
    bins->order.swap(vals_b);
  }
  // keys_b is free now: it receives the sorted keys' cell ordinals while cell_of still holds the sorted keys
  GP_TRY(exclusive_scan_of(CellStartFlag{bins->cell_of.as<unsigned>()}, scanned.as<int>(), n, scan_scratch.as<int>() + (int)(((long long)n + kScanThreads - 1) / kScanThreads), s,
                           st + sort_words));
  const int scan_blocks = (int)(((long long)n + kScanThreads - 1) / kScanThreads);
  const int* d_total = scan_scratch.as<int>() + scan_blocks;  // the scan's grand total = number of cells
  // (round 4: no host round trip in the middle -- the arrays the cell count would size are allocated for the worst case, one cell per point, and the kernels behind
  // read the count where the scan left it; the host learns cells, occupied blocks and binned points together at the end)
  const double t2 = now();
  const double t3 = t2;
  GP_TRY(bins->cell_start.alloc_pooled(sizeof(int) * ((size_t)n + 1), s));
  GP_TRY(bins->cell_block.alloc_pooled(sizeof(int) * (size_t)n, s));
  GP_TRY(bins->occ_blocks.alloc_pooled(sizeof(int) * (size_t)n, s));  // at most one block per cell
  hipLaunchKernelGGL(bins_finish_kernel, dim3(wgs), dim3(256), 0, s, (const unsigned*)bins->cell_of.as<unsigned>(), (const int*)scanned.as<int>(), n, d_total, bins->blocks.as<GridBlock>(), bins->cell_start.as<int>(), keys_b.as<unsigned>(), bins->cell_block.as<int>(),
                     d_small.as<int>() + 8, hw.dev);
  GP_HIP(hipGetLastError());
  bins->cell_of.swap(keys_b);  // cell_of = ordinals of the sorted points
  // the compact list of occupied blocks (flags / scanned are re-used: num_cells <= n; entries behind the last cell are flagged 0)
  // (bins_finish_kernel left the cell count in d_small[9]: the second scan re-uses the first one's slot)
  const int* d_cells = d_small.as<int>() + 9;
  GP_TRY(exclusive_scan_of(BlockStartFlag{d_cells, bins->cell_block.as<int>(), bins->blocks.as<GridBlock>()}, scanned.as<int>(), n, const_cast<int*>(d_total), s,
                           st + sort_words + scan_words));
  hipLaunchKernelGGL(bins_block_list_kernel, dim3(wgs), dim3(256), 0, s, d_cells, (const int*)bins->cell_block.as<int>(), (const GridBlock*)bins->blocks.as<GridBlock>(),
                     (const int*)scanned.as<int>(), bins->occ_blocks.as<int>(), d_total, hw.dev + 10);
  GP_HIP(hipGetLastError());
  // the occupied-block count = the second scan's total: bins_block_list_kernel copies it next to the others
  GP_HIP(hipStreamSynchronize(s));
  const int h_counts[3] = {reinterpret_cast<volatile int*>(hw.host)[9], reinterpret_cast<volatile int*>(hw.host)[10], reinterpret_cast<volatile int*>(hw.host)[8]};  // cells, occupied blocks, binned points
  bins->num_cells = h_counts[0];
  bins->num_occ_blocks = h_counts[0] > 0 ? h_counts[1] : 0;
  bins->num_binned = h_counts[2];
  if (dbg) fprintf(stderr, "bin_points: bbox %.0f us, sort issue %.0f us, wait %.0f us, finish %.0f us\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
  return GP_OK;
}

}  // namespace gp
