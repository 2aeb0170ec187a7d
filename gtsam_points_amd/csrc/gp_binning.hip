// gp_binning.hip -- see gp_binning.hpp
#include "gp_binning.hpp"

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

// occupancy bits.  Most points fall into a cell whose bit is already set: a plain load filters them out before the atomic
// (a stale read only costs a redundant atomicOr)
__global__ void __launch_bounds__(256) bins_mark_kernel(const float* __restrict__ points, int n, double inv_cell, GridGeom g, GridBlock* __restrict__ blocks) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n) return;
  int cx, cy, cz;
  if (!point_cell(points, i, inv_cell, cx, cy, cz)) return;
  unsigned long long* w = &blocks[grid_block_index(g, cx, cy, cz)].bits;
  const unsigned long long bit = 1ull << grid_bit(cx, cy, cz);
  if (!(*reinterpret_cast<volatile unsigned long long*>(w) & bit)) atomicOr(w, bit);
}

__global__ void __launch_bounds__(256) bins_count_kernel(long long num_blocks, GridBlock* __restrict__ blocks) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b < num_blocks) blocks[b].base = __popcll(blocks[b].bits);
}

__global__ void __launch_bounds__(256) bins_ordinal_kernel(const float* __restrict__ points, int n, double inv_cell, GridGeom g, const GridBlock* __restrict__ blocks,
                                                           unsigned* __restrict__ keys) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n) return;
  int cx, cy, cz;
  unsigned key = kInvalidKey;
  if (point_cell(points, i, inv_cell, cx, cy, cz)) {
    const GridBlock blk = blocks[grid_block_index(g, cx, cy, cz)];
    key = (unsigned)(blk.base + __popcll(blk.bits & ((1ull << grid_bit(cx, cy, cz)) - 1ull)));
  }
  keys[i] = key;
}

// sorted keys -> first position of every cell; every cell holds at least one point, so every entry is written exactly once
__global__ void __launch_bounds__(256) bins_starts_kernel(const unsigned* __restrict__ sorted_keys, int n, int num_cells, int* __restrict__ cell_start,
                                                          int* __restrict__ num_binned) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= (size_t)n) return;
  const unsigned k = sorted_keys[j];
  const unsigned prev = j > 0 ? sorted_keys[j - 1] : 0xffffffffu;
  if (k != prev) {
    if (k == kInvalidKey) {
      cell_start[num_cells] = (int)j;
      *num_binned = (int)j;
    } else {
      cell_start[k] = (int)j;
    }
  }
  if (j == (size_t)n - 1 && k != kInvalidKey) {
    cell_start[num_cells] = n;
    *num_binned = n;
  }
}

}  // namespace

int bin_points(const float* points_dev, int n, double inv_cell, hipStream_t s, PointBins* bins, bool* too_large) {
  *too_large = false;
  bins->num_cells = 0;
  bins->num_binned = 0;
  bins->num_blocks = 0;
  if (n <= 0) {
    GP_TRY(bins->cell_start.alloc(sizeof(int)));
    GP_HIP(hipMemsetAsync(bins->cell_start.ptr, 0, sizeof(int), s));
    return GP_OK;
  }
  const int wgs = (n + 255) / 256;
  // ---- bounding box ----
  DeviceArray boxes, d_small;
  GP_TRY(boxes.alloc_async(sizeof(int) * 6 * (size_t)wgs, s));
  GP_TRY(d_small.alloc_async(sizeof(int) * 16, s));
  hipLaunchKernelGGL(bins_bbox_kernel, dim3(wgs), dim3(256), 0, s, points_dev, n, inv_cell, boxes.as<int>());
  hipLaunchKernelGGL(bins_bbox_reduce_kernel, dim3(1), dim3(256), 0, s, (const int*)boxes.as<int>(), wgs, d_small.as<int>());
  GP_HIP(hipGetLastError());
  int h_bbox[6];
  GP_HIP(hipMemcpyAsync(h_bbox, d_small.ptr, sizeof(h_bbox), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  if (h_bbox[0] > h_bbox[3]) {  // no finite point at all
    GP_TRY(bins->cell_start.alloc(sizeof(int)));
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
  // ---- occupancy bits, bases ----
  GP_TRY(bins->blocks.alloc(sizeof(GridBlock) * (size_t)bins->num_blocks));
  GP_HIP(hipMemsetAsync(bins->blocks.ptr, 0, sizeof(GridBlock) * (size_t)bins->num_blocks, s));
  GridBlock* blocks = bins->blocks.as<GridBlock>();
  hipLaunchKernelGGL(bins_mark_kernel, dim3(wgs), dim3(256), 0, s, points_dev, n, inv_cell, bins->geom, blocks);
  hipLaunchKernelGGL(bins_count_kernel, dim3((unsigned)((bins->num_blocks + 255) / 256)), dim3(256), 0, s, bins->num_blocks, blocks);
  GP_HIP(hipGetLastError());
  DeviceArray scan_scratch;
  GP_TRY(scan_scratch.alloc_async(sizeof(int) * (size_t)(bins->num_blocks / kScanThreads + 8), s));
  int* base0 = &blocks[0].base;
  GP_TRY(exclusive_scan_strided(base0, 4, base0, 4, bins->num_blocks, scan_scratch.as<int>(), s));
  // total = the scan's grand total (kept behind the block sums, exclusive_scan_strided: scratch[nb])
  const int scan_blocks = (int)((bins->num_blocks + kScanThreads - 1) / kScanThreads);
  int h_cells = 0;
  GP_HIP(hipMemcpyAsync(&h_cells, scan_scratch.as<int>() + scan_blocks, sizeof(int), hipMemcpyDeviceToHost, s));
  // ---- ordinals + stable sort ----
  DeviceArray keys_b, vals_b, sort_scratch;
  GP_TRY(bins->cell_of.alloc(sizeof(unsigned) * (size_t)n));
  GP_TRY(bins->order.alloc(sizeof(int) * (size_t)n));
  GP_TRY(keys_b.alloc_async(sizeof(unsigned) * (size_t)n, s));
  GP_TRY(vals_b.alloc_async(sizeof(int) * (size_t)n, s));
  GP_TRY(sort_scratch.alloc_async(sizeof(int) * radix_sort_scratch_ints(n), s));
  hipLaunchKernelGGL(bins_ordinal_kernel, dim3(wgs), dim3(256), 0, s, points_dev, n, inv_cell, bins->geom, (const GridBlock*)blocks, bins->cell_of.as<unsigned>());
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize(s));  // the cell count decides the number of radix passes
  bins->num_cells = h_cells;
  // sort over the bits of num_cells: the invalid key (0x7fffffff) of skipped points must sort last, so they are re-keyed to
  // num_cells by sorting over enough bits to hold it... simpler: 31 bits are only needed when points were skipped; the common
  // case sorts ceil(log2(num_cells + 1)) bits and treats a key >= num_cells as "skipped" afterwards
  int bits = 1;
  while ((1ll << bits) <= (long long)h_cells) bits++;
  bool in_b = false;
  // skipped points carry kInvalidKey whose low `bits` bits are all ones = 2^bits - 1 >= num_cells: they land behind every cell
  GP_TRY(radix_sort_pairs(bins->cell_of.as<unsigned>(), bins->order.as<int>(), keys_b.as<unsigned>(), vals_b.as<int>(), n, bits, true, sort_scratch.as<int>(), s, &in_b));
  if (in_b) {
    GP_HIP(hipMemcpyAsync(bins->cell_of.ptr, keys_b.ptr, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(bins->order.ptr, vals_b.ptr, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  GP_TRY(bins->cell_start.alloc(sizeof(int) * ((size_t)h_cells + 1)));
  hipLaunchKernelGGL(bins_starts_kernel, dim3(wgs), dim3(256), 0, s, (const unsigned*)bins->cell_of.as<unsigned>(), n, h_cells, bins->cell_start.as<int>(), d_small.as<int>() + 8);
  GP_HIP(hipGetLastError());
  int h_binned = 0;
  GP_HIP(hipMemcpyAsync(&h_binned, d_small.as<int>() + 8, sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  bins->num_binned = h_binned;
  return GP_OK;
}

}  // namespace gp
