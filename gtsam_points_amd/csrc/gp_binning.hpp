// gp_binning.hpp -- deterministic binning of a point cloud into the cells of a uniform grid (gp_binning.hip).
//
// The structure behind both the Gaussian voxel-map build (cell = voxel, gp_voxelmap.hip) and the exact k-NN / GICP search
// (cell = search cell, gp_knn.hip):
//   * an occupancy-block grid over the cells (gp_device.hpp: GridBlock, 16 B per 4 x 4 x 4 cells): cell ordinal =
//     base + popcount(occupancy bits below the cell's bit), cells numbered in (block, bit) order;
//   * `order`: the point indices sorted by cell ordinal, STABLE (gp_sort.hpp) -- inside a cell ascending point index;
//   * `cell_start[ordinal]`: first position of the cell in `order` (cell_start[num_cells] = number of binned points).
// Non-finite points are left out (they sort behind all cells).  Everything is computed without atomics on sums or cursors, so
// the result is identical from run to run.
#pragma once

#include "gp_host.hpp"

namespace gp {

struct GridGeom {
  int lo[3];   // block coordinate (cell coordinate >> 2) of the low corner
  int dim[3];  // blocks per axis
};

__host__ __device__ __forceinline__ long long grid_block_index(const GridGeom& g, int cx, int cy, int cz) {
  const int bx = (cx >> 2) - g.lo[0], by = (cy >> 2) - g.lo[1], bz = (cz >> 2) - g.lo[2];
  return ((long long)bz * g.dim[1] + by) * g.dim[0] + bx;
}
__host__ __device__ __forceinline__ int grid_bit(int cx, int cy, int cz) { return ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3); }
// inverse of (block index, bit) -> cell coordinate
__host__ __device__ __forceinline__ void grid_cell_coord(const GridGeom& g, long long block, int bit, int& cx, int& cy, int& cz) {
  const int bx = (int)(block % g.dim[0]), by = (int)((block / g.dim[0]) % g.dim[1]), bz = (int)(block / ((long long)g.dim[0] * g.dim[1]));
  cx = ((bx + g.lo[0]) << 2) | (bit & 3);
  cy = ((by + g.lo[1]) << 2) | ((bit >> 2) & 3);
  cz = ((bz + g.lo[2]) << 2) | ((bit >> 4) & 3);
}

constexpr long long kMaxGridBlocks = 1ll << 24;  // 16 B each: at most 256 MB; a larger bounding box is reported, not built

struct PointBins {
  GridGeom geom{};
  long long num_blocks = 0;
  int num_cells = 0;
  int num_binned = 0;      // points with finite coordinates
  DeviceArray blocks;      // GridBlock[num_blocks]
  DeviceArray cell_start;  // int[num_cells + 1]
  DeviceArray order;       // int[n]: point indices, cell-major, ascending inside a cell; positions >= num_binned hold the skipped points
  DeviceArray cell_of;     // unsigned[n]: cell ordinal of order[j] (sorted keys); kept for the consumers' segmented passes
  DeviceArray cell_block;  // int[num_cells]: block index of every cell
  DeviceArray occ_blocks;  // int[num_occ_blocks]: indices of the occupied blocks, ascending (work list of block-tiled kernels)
  int num_occ_blocks = 0;
};

// Bins points_dev[n] (float xyz, 12-B stride) by floor(p * inv_cell) in double -- the CPU map's rule (util/fast_floor.hpp:12-15).
// Returns GP_OK with bins->num_cells >= 0, or GP_ERROR_INVALID_ARGUMENT with *too_large = true when the bounding box of the
// occupied cells needs more than kMaxGridBlocks blocks (nothing is built then; the caller falls back to its hashed structure).
// Synchronises the stream (twice: bounding box, cell count).
// The build's radix sort hands tiles out through 32 ticket classes, which is fast and rests on the device starting workgroups in blockIdx order (gp_sort.hpp); a
// sort that finds that order violated gives up instead of spinning, and the build runs again with the single ticket counter that needs no such order.
int bin_points(const float* points_dev, int n, double inv_cell, hipStream_t s, PointBins* bins, bool* too_large);
void inject_sort_faults(int count);  // test hook (thread-local): the next `count` builds see a faulted first sort
int sort_fallbacks();                // builds of this thread that went through the one-class sort

}  // namespace gp
