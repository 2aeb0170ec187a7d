// gp_voxelmap.hip -- GaussianVoxelMapGPU for gfx950.
//
// Replaces src/gtsam_points/types/gaussian_voxelmap_gpu.cu (thrust::for_each functors :25-172, insert :211-251,
// create_bucket_table :253-307, save/load :309-467, offload/reload :469-535, download_* :537-571) and the
// overlap/lookup half of gaussian_voxelmap_gpu_funcs.cu:156-236.
//
// Differences by design:
//   * voxel coordinates are floor(double(p) * (1.0/leaf)) -- the CPU map's rule (gaussian_voxelmap_cpu.cpp:59-61)
//   * statistics are accumulated with native f64 atomics relative to the voxel centre, not unordered f32 atomics
//   * besides the reference-visible arrays (num_points / voxel_means / voxel_covs / voxel_intensities) the map
//     keeps one aligned 64-B gather record per voxel (gp::VoxelRecord) that the VGICP kernels read
//   * no intermediate coordinate array: a bucket's representative point is re-floored when compared
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

#include "gp_binning.hpp"
#include "gp_host.hpp"
#include "gp_scan.hpp"

namespace gp {

__device__ __forceinline__ void point_coord(const float* __restrict__ points, int i, double inv_leaf, int& cx, int& cy, int& cz) {
  cx = fast_floor((double)points[3 * (size_t)i] * inv_leaf);
  cy = fast_floor((double)points[3 * (size_t)i + 1] * inv_leaf);
  cz = fast_floor((double)points[3 * (size_t)i + 2] * inv_leaf);
}

// voxel_bucket_assignment_kernel (gaussian_voxelmap_gpu.cu:37-75): claim a bucket per distinct voxel coordinate.
// rep[b] = index of the first point that claimed bucket b, or -1.
__global__ void __launch_bounds__(256) claim_buckets_kernel(const float* __restrict__ points, int n, int* __restrict__ rep, uint32_t num_buckets,
                                                            uint32_t mask, int max_scan, double inv_leaf, int* __restrict__ failures) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  point_coord(points, i, inv_leaf, cx, cy, cz);
  const uint64_t hash = coord_hash(cx, cy, cz);
  for (int j = 0; j < max_scan; j++) {
    const uint32_t b = bucket_index(hash, j, num_buckets, mask);
    const int old = atomicCAS(&rep[b], -1, i);
    if (old < 0) return;  // claimed an empty bucket
    int ox, oy, oz;
    point_coord(points, old, inv_leaf, ox, oy, oz);
    if (ox == cx && oy == cy && oz == cz) return;  // voxel already present
  }
  atomicAdd(failures, 1);  // probe chain exhausted: this point is dropped (gaussian_voxelmap_gpu.cu:67)
}

// voxel_coord_select_kernel (:77-90) + voxel id allocation (:57-60)
__global__ void __launch_bounds__(256) assign_voxels_kernel(const float* __restrict__ points, const int* __restrict__ rep, uint32_t num_buckets,
                                                            double inv_leaf, gp_voxel_bucket* __restrict__ buckets, int* __restrict__ voxel_coords,
                                                            int* __restrict__ num_voxels) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= num_buckets) return;
  const int r = rep[b];
  int4 out = make_int4(0, 0, 0, -1);
  if (r >= 0) {
    int cx, cy, cz;
    point_coord(points, r, inv_leaf, cx, cy, cz);
    const int v = atomicAdd(num_voxels, 1);
    out = make_int4(cx, cy, cz, v);
    voxel_coords[3 * (size_t)v] = cx;
    voxel_coords[3 * (size_t)v + 1] = cy;
    voxel_coords[3 * (size_t)v + 2] = cz;
  }
  reinterpret_cast<int4*>(buckets)[b] = out;
}

// accumulate_points_kernel (:92-152): sums[v] = { sum(p - centre) (3), sum upper(C) (6) } in double, count, max intensity
__global__ void __launch_bounds__(256) accumulate_kernel(const float* __restrict__ points, const float* __restrict__ covs,
                                                         const float* __restrict__ intensities, int n, VoxelMapView map, double* __restrict__ sums,
                                                         int* __restrict__ counts, unsigned int* __restrict__ intensity_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double px = (double)points[3 * (size_t)i], py = (double)points[3 * (size_t)i + 1], pz = (double)points[3 * (size_t)i + 2];
  const int cx = fast_floor(px * map.inv_leaf), cy = fast_floor(py * map.inv_leaf), cz = fast_floor(pz * map.inv_leaf);
  const int v = lookup_voxel(map, cx, cy, cz);
  if (v < 0) return;  // dropped at table build
  double ox, oy, oz;
  voxel_center(map, cx, cy, cz, ox, oy, oz);
  const float* c = covs + 9 * (size_t)i;
  double* s = sums + 9 * (size_t)v;
  unsafeAtomicAdd(s + 0, px - ox);
  unsafeAtomicAdd(s + 1, py - oy);
  unsafeAtomicAdd(s + 2, pz - oz);
  unsafeAtomicAdd(s + 3, (double)c[0]);  // xx
  unsafeAtomicAdd(s + 4, 0.5 * ((double)c[3] + (double)c[1]));  // xy: symmetric part of the column-major 3x3 (the input itself when symmetric)
  unsafeAtomicAdd(s + 5, 0.5 * ((double)c[6] + (double)c[2]));  // xz
  unsafeAtomicAdd(s + 6, (double)c[4]);                         // yy
  unsafeAtomicAdd(s + 7, 0.5 * ((double)c[7] + (double)c[5]));  // yz
  unsafeAtomicAdd(s + 8, (double)c[8]);  // zz
  atomicAdd(counts + v, 1);
  if (intensities) atomicMax(intensity_bits + v, __float_as_uint(intensities[i]));  // max intensity (:138-139)
}

// finalize_voxels_kernel (:154-172): divide by the count; emit the gather record and the reference-visible arrays
__global__ void __launch_bounds__(256) finalize_kernel(int num_voxels, double leaf, const int* __restrict__ voxel_coords, const double* __restrict__ sums,
                                                       const int* __restrict__ counts, VoxelRecord* __restrict__ records,
                                                       float* __restrict__ voxel_means, float* __restrict__ voxel_covs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_voxels) return;
  const int n = counts[v];
  const double inv_n = 1.0 / (double)n;
  const double* s = sums + 9 * (size_t)v;
  const double lx = s[0] * inv_n, ly = s[1] * inv_n, lz = s[2] * inv_n;
  VoxelRecord rec;
  rec.mean_local[0] = (float)lx;
  rec.mean_local[1] = (float)ly;
  rec.mean_local[2] = (float)lz;
  rec.num_points = n;
  for (int k = 0; k < 6; k++) rec.cov[k] = s[3 + k] / (double)n;
  records[v] = rec;
  const double ox = ((double)voxel_coords[3 * (size_t)v] + 0.5) * leaf;
  const double oy = ((double)voxel_coords[3 * (size_t)v + 1] + 0.5) * leaf;
  const double oz = ((double)voxel_coords[3 * (size_t)v + 2] + 0.5) * leaf;
  voxel_means[3 * (size_t)v] = (float)(ox + lx);
  voxel_means[3 * (size_t)v + 1] = (float)(oy + ly);
  voxel_means[3 * (size_t)v + 2] = (float)(oz + lz);
  float* c = voxel_covs + 9 * (size_t)v;
  c[0] = (float)rec.cov[0];
  c[1] = c[3] = (float)rec.cov[1];
  c[2] = c[6] = (float)rec.cov[2];
  c[4] = (float)rec.cov[3];
  c[5] = c[7] = (float)rec.cov[4];
  c[8] = (float)rec.cov[5];
}

// line table: claim the first free key slot of the home line (front to back), else walk to the next line
__global__ void __launch_bounds__(256) line_claim_kernel(int num_voxels, const int* __restrict__ voxel_coords, gp_voxel_bucket* __restrict__ lines, uint32_t lmask) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_voxels) return;
  const int cx = voxel_coords[3 * (size_t)v], cy = voxel_coords[3 * (size_t)v + 1], cz = voxel_coords[3 * (size_t)v + 2];
  uint32_t l = coord_hash32(cx, cy, cz) & lmask;
  for (;;) {
    gp_voxel_bucket* line = lines + 4 * (size_t)l;
    for (int t = 0; t < 4; t++) {
      if (atomicCAS(&line[t].voxel_index, -1, v) == -1) {
        line[t].coord[0] = cx;
        line[t].coord[1] = cy;
        line[t].coord[2] = cz;
        return;
      }
    }
    l = (l + 1) & lmask;
  }
}


// ---- occupancy-block grid (gp_device.hpp: GridBlock; geometry helpers in gp_binning.hpp) ----

// bounding box of the voxel coordinates in block units: bbox[0..2] = min, bbox[3..5] = max (wave reduce + one atomic per wave)
__global__ void __launch_bounds__(256) coords_bbox_kernel(int num_voxels, const int* __restrict__ voxel_coords, int* __restrict__ bbox) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int u = v < num_voxels ? v : num_voxels - 1;
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) lo[a] = hi[a] = voxel_coords[3 * (size_t)u + a] >> 2;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = max(hi[a], __shfl_xor(hi[a], off, 64));
    }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      atomicMin(bbox + a, lo[a]);
      atomicMax(bbox + 3 + a, hi[a]);
    }
  }
}

__global__ void __launch_bounds__(256) grid_mark_kernel(int num_voxels, const int* __restrict__ voxel_coords, GridGeom g, GridBlock* __restrict__ blocks) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_voxels) return;
  const int cx = voxel_coords[3 * (size_t)v], cy = voxel_coords[3 * (size_t)v + 1], cz = voxel_coords[3 * (size_t)v + 2];
  atomicOr(&blocks[grid_block_index(g, cx, cy, cz)].bits, 1ull << grid_bit(cx, cy, cz));
}

__global__ void __launch_bounds__(256) grid_count_kernel(long long num_blocks, GridBlock* __restrict__ blocks) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < num_blocks) blocks[b].base = __popcll(blocks[b].bits);
}

// new index of every voxel = base of its block + number of occupied bits below its own; coordinates move to the new order
__global__ void __launch_bounds__(256) grid_renumber_kernel(int num_voxels, const int* __restrict__ coords_old, GridGeom g, const GridBlock* __restrict__ blocks,
                                                            int* __restrict__ perm, int* __restrict__ coords_new) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_voxels) return;
  const int cx = coords_old[3 * (size_t)v], cy = coords_old[3 * (size_t)v + 1], cz = coords_old[3 * (size_t)v + 2];
  const GridBlock blk = blocks[grid_block_index(g, cx, cy, cz)];
  const int nv = blk.base + __popcll(blk.bits & ((1ull << grid_bit(cx, cy, cz)) - 1ull));
  perm[v] = nv;
  coords_new[3 * (size_t)nv] = cx;
  coords_new[3 * (size_t)nv + 1] = cy;
  coords_new[3 * (size_t)nv + 2] = cz;
}

__global__ void __launch_bounds__(256) buckets_renumber_kernel(uint32_t num_buckets, gp_voxel_bucket* __restrict__ buckets, const int* __restrict__ perm) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= num_buckets) return;
  const int v = buckets[b].voxel_index;
  if (v >= 0) buckets[b].voxel_index = perm[v];
}

// ---- binned build (gp_binning.hpp): per-voxel statistics as an ORDERED segmented sum ----------------------------------------
// Sixteen lanes per voxel, sixteen voxels per workgroup (rounds 2-3: one wave per voxel -- the median voxel of a scan holds three points, the mean 10-30, so three
// quarters of a wave's lanes idled through nine six-step butterflies: 114 us per 2 M points, the largest kernel of the map build).  The rows of a voxel's points are
// a random gather through the sort order (12 + 36 B from wherever the point lies in the caller's arrays); when the lanes of a voxel fetched their own rows, the wave
// waited for its longest voxel with most lanes idle (68 us per 2 M points at 2.3 TB/s of fabric traffic: latency-bound).  Now the WHOLE workgroup gathers the rows of
// its sixteen voxels -- one contiguous range of the sort order, every lane busy, two rows per lane in flight -- into LDS, 512 rows at a time, and the voxels' lanes
// sum from LDS.  Lane l of a voxel adds the voxel's points l, l + 16, ... in ascending order (stable sort: ascending point index), one fixed butterfly joins the
// sixteen lanes at the end: the statistics are bit-identical from run to run (no atomics) and do not depend on where the 512-row batches fall.
// sums relative to the voxel centre in f64, like accumulate_kernel; outputs as finalize_kernel.
constexpr int kStatsBatch = 512;
__global__ void __launch_bounds__(256) segmented_stats_kernel(const float* __restrict__ points, const float* __restrict__ covs, const float* __restrict__ intensities,
                                                              int num_voxels, const int* __restrict__ cell_start, const int* __restrict__ order, double inv_leaf, double leaf,
                                                              VoxelRecord* __restrict__ records, int* __restrict__ num_points, float* __restrict__ voxel_means,
                                                              float* __restrict__ voxel_covs, float* __restrict__ voxel_intensities, int* __restrict__ voxel_coords,
                                                              const gp::FillJob fill_buckets) {
  gp::run_fill_job(fill_buckets);  // (the bucket table the insertion kernel behind this one claims its slots in: 0xff = empty)
  constexpr int kGroup = 16;
  __shared__ float rows[kStatsBatch][12];  // x y z, then the covariance's nine entries
  __shared__ float inten[kStatsBatch];
  const int lane = threadIdx.x & (kGroup - 1);
  const int v0 = blockIdx.x * (256 / kGroup);
  const int v1 = min(v0 + 256 / kGroup, num_voxels);
  const int v = v0 + (threadIdx.x / kGroup);
  const bool live = v < num_voxels;
  const int p0 = cell_start[v0], p1 = cell_start[v1];
  const int b = live ? cell_start[v] : p1, e = live ? cell_start[v + 1] : p1;
  int cx = 0, cy = 0, cz = 0;
  double ox = 0.0, oy = 0.0, oz = 0.0;  // the voxel's centre, from its first point -- taken from the batch that holds it (two dependent loads less in front of the gather)
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float imax = 0.0f;  // max intensity (:138-139): intensities are non-negative upstream (atomicMax on the float bits), 0 when absent
  for (int batch = p0; batch < p1; batch += kStatsBatch) {
    const int cnt = min(kStatsBatch, p1 - batch);
    float row[kStatsBatch / 256][12];
    float rit[kStatsBatch / 256];
#pragma unroll
    for (int q = 0; q < kStatsBatch / 256; q++) {
      const int r = q * 256 + (int)threadIdx.x;
      rit[q] = 0.0f;
      if (r < cnt) {
        const size_t i = (size_t)order[batch + r];
#pragma unroll
        for (int k = 0; k < 3; k++) row[q][k] = points[3 * i + k];
#pragma unroll
        for (int k = 0; k < 9; k++) row[q][3 + k] = covs[9 * i + k];
        if (intensities) rit[q] = intensities[i];
      }
    }
#pragma unroll
    for (int q = 0; q < kStatsBatch / 256; q++) {
      const int r = q * 256 + (int)threadIdx.x;
      if (r < cnt) {
#pragma unroll
        for (int k = 0; k < 12; k++) rows[r][k] = row[q][k];
        inten[r] = rit[q];
      }
    }
    __syncthreads();
    const int lo = max(b, batch), hi = min(e, batch + cnt);
    if (live && b >= batch && b < batch + cnt) {
      const float* f = rows[b - batch];
      cx = fast_floor((double)f[0] * inv_leaf), cy = fast_floor((double)f[1] * inv_leaf), cz = fast_floor((double)f[2] * inv_leaf);
      ox = ((double)cx + 0.5) * leaf, oy = ((double)cy + 0.5) * leaf, oz = ((double)cz + 0.5) * leaf;
    }
    for (int j = lo + ((lane - (lo - b)) & (kGroup - 1)); j < hi; j += kGroup) {
      const float* c = rows[j - batch];
      acc[0] += (double)c[0] - ox;
      acc[1] += (double)c[1] - oy;
      acc[2] += (double)c[2] - oz;
      acc[3] += (double)c[3];
      acc[4] += 0.5 * ((double)c[6] + (double)c[4]);  // symmetric part of the column-major 3x3 (the input itself when symmetric)
      acc[5] += 0.5 * ((double)c[9] + (double)c[5]);
      acc[6] += (double)c[7];
      acc[7] += 0.5 * ((double)c[10] + (double)c[8]);
      acc[8] += (double)c[11];
      imax = fmaxf(imax, inten[j - batch]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int off = kGroup / 2; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, kGroup);
  }
#pragma unroll
  for (int off = kGroup / 2; off > 0; off >>= 1) imax = fmaxf(imax, __shfl_xor(imax, off, kGroup));
  if (!live) return;
  if (lane != 0) return;
  const int n = e - b;
  const double inv_n = 1.0 / (double)n;
  VoxelRecord rec;
  const double lx = acc[0] * inv_n, ly = acc[1] * inv_n, lz = acc[2] * inv_n;
  rec.mean_local[0] = (float)lx;
  rec.mean_local[1] = (float)ly;
  rec.mean_local[2] = (float)lz;
  rec.num_points = n;
  for (int k = 0; k < 6; k++) rec.cov[k] = acc[3 + k] / (double)n;
  records[v] = rec;
  num_points[v] = n;
  voxel_means[3 * (size_t)v] = (float)(ox + lx);
  voxel_means[3 * (size_t)v + 1] = (float)(oy + ly);
  voxel_means[3 * (size_t)v + 2] = (float)(oz + lz);
  float* c = voxel_covs + 9 * (size_t)v;
  c[0] = (float)rec.cov[0];
  c[1] = c[3] = (float)rec.cov[1];
  c[2] = c[6] = (float)rec.cov[2];
  c[4] = (float)rec.cov[3];
  c[5] = c[7] = (float)rec.cov[4];
  c[8] = (float)rec.cov[5];
  voxel_intensities[v] = imax;
  voxel_coords[3 * (size_t)v] = cx;
  voxel_coords[3 * (size_t)v + 1] = cy;
  voxel_coords[3 * (size_t)v + 2] = cz;
}

// reference-visible bucket table from the list of DISTINCT voxels: claim the first free bucket of the probe chain
// (reference hash + max_bucket_scan_count rule, so lookup_voxel finds it); failed[0] += points of a voxel whose chain is full
__global__ void __launch_bounds__(256) insert_voxels_kernel(int num_voxels, const int* __restrict__ voxel_coords, const int* __restrict__ num_points,
                                                            gp_voxel_bucket* __restrict__ buckets, uint32_t num_buckets, uint32_t mask, int max_scan,
                                                            int* __restrict__ failed) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_voxels) return;
  const int cx = voxel_coords[3 * (size_t)v], cy = voxel_coords[3 * (size_t)v + 1], cz = voxel_coords[3 * (size_t)v + 2];
  const uint64_t hash = coord_hash(cx, cy, cz);
  for (int j = 0; j < max_scan; j++) {
    gp_voxel_bucket* b = buckets + bucket_index(hash, j, num_buckets, mask);
    if (atomicCAS(&b->voxel_index, -1, v) == -1) {
      b->coord[0] = cx;
      b->coord[1] = cy;
      b->coord[2] = cz;
      return;
    }
  }
  *failed = 1;  // (a flag, not a count: the caller doubles the table until nothing fails; host-mapped on the binned build's path)
}

// lookup_voxels_kernel (cuda/kernels/lookup_voxels.cuh:34-60): voxel index of delta * p, or -1
__global__ void __launch_bounds__(256) lookup_kernel(const float* __restrict__ points, const float* __restrict__ normals, int n, VoxelMapView map,
                                                     const double* __restrict__ pose, int* __restrict__ out, int* __restrict__ hit_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int v = -1;
  if (i < n) {
    const Pose T = load_pose(pose);
    const double px = (double)points[3 * (size_t)i], py = (double)points[3 * (size_t)i + 1], pz = (double)points[3 * (size_t)i + 2];
    const double qx = T.r00 * px + T.r01 * py + T.r02 * pz + T.tx;
    const double qy = T.r10 * px + T.r11 * py + T.r12 * pz + T.ty;
    const double qz = T.r20 * px + T.r21 * py + T.r22 * pz + T.tz;
    bool rejected = !finite3(qx, qy, qz);  // a NaN / inf return lies in no voxel
    if (normals && !rejected) rejected = surface_rejected(T, qx, qy, qz, normals + 3 * (size_t)i);
    if (!rejected) v = lookup_voxel(map, fast_floor(qx * map.inv_leaf), fast_floor(qy * map.inv_leaf), fast_floor(qz * map.inv_leaf));
    if (out) out[i] = v;
  }
  if (hit_count) {
    const unsigned long long hits = __ballot(v >= 0);
    if ((threadIdx.x & 63) == 0 && hits) atomicAdd(hit_count, __popcll(hits));
  }
}

}  // namespace gp

gp::VoxelMapView gp_voxelmap::view() const {
  gp::VoxelMapView v;
  v.gblocks = has_grid ? gblocks.as<gp::GridBlock>() : nullptr;
  for (int a = 0; a < 3; a++) {
    v.glo[a] = glo[a];
    v.gdim[a] = gdim[a];
  }
  v.plines = plines.as<gp_voxel_bucket>();
  v.plmask = plmask;
  v.pad_ = 0;
  v.buckets = buckets.as<gp_voxel_bucket>();
  v.records = records.as<gp::VoxelRecord>();
  v.voxel_coords = voxel_coords.as<int>();
  v.num_buckets = (uint32_t)info.num_buckets;
  v.bucket_mask = (info.num_buckets > 0 && (info.num_buckets & (info.num_buckets - 1)) == 0) ? (uint32_t)(info.num_buckets - 1) : 0u;
  v.max_scan = info.max_bucket_scan_count;
  v.num_voxels = info.num_voxels;
  v.inv_leaf = 1.0 / resolution;
  v.leaf = resolution;
  return v;
}

namespace {

constexpr int kBlock = 256;
inline int grid_for(size_t n) { return (int)((n + kBlock - 1) / kBlock); }

// GaussianVoxelData (types/gaussian_voxel_data.hpp:11-54): 56-byte on-disk record
struct VoxelDataRecord {
  int coord[3];
  int num_points;
  float mean[3];
  float cov[6];  // xx xy xz yy yz zz
  float intensity;
};
static_assert(sizeof(VoxelDataRecord) == 56, "GaussianVoxelData must be 56 B");

int alloc_voxel_arrays(gp_voxelmap* m, int V) {
  hipStream_t s = m->stream;
  GP_TRY(m->records.alloc_pooled(sizeof(gp::VoxelRecord) * (size_t)V, s));
  GP_TRY(m->num_points.alloc_pooled(sizeof(int) * (size_t)V, s));
  GP_TRY(m->voxel_means.alloc_pooled(sizeof(float) * 3 * (size_t)V, s));
  GP_TRY(m->voxel_covs.alloc_pooled(sizeof(float) * 9 * (size_t)V, s));
  GP_TRY(m->voxel_intensities.alloc_pooled(sizeof(float) * (size_t)V, s));
  return GP_OK;
}

}  // namespace

// (re)build the pipeline kernel's line table from the voxel list; called at the end of insert / assign / reload
static int build_private_table(gp_voxelmap* m, hipStream_t s) {
  const int V = m->info.num_voxels;
  uint32_t lines = 256;
  while (lines < 2u * (uint32_t)std::max(V, 1)) lines <<= 1;  // 4 key slots per line: load factor <= 1/8
  m->plmask = lines - 1;
  GP_TRY(m->plines.alloc_pooled(64 * (size_t)lines, s));
  GP_HIP(hipMemsetAsync(m->plines.ptr, 0xff, 64 * (size_t)lines, s));
  if (V > 0) {
    hipLaunchKernelGGL(gp::line_claim_kernel, dim3((V + 255) / 256), dim3(256), 0, s, V, m->voxel_coords.as<int>(), m->plines.as<gp_voxel_bucket>(), m->plmask);
    GP_HIP(hipGetLastError());
  }
  m->private_built = true;
  return GP_OK;
}

int gp_voxelmap::ensure_private_table() {
  std::lock_guard<std::mutex> lock(private_mutex);
  if (private_built) return GP_OK;
  if (!loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "voxel map is not loaded on the GPU");
  GP_TRY(build_private_table(this, stream));
  GP_HIP(hipStreamSynchronize(stream));
  return GP_OK;
}


using gp::kMaxGridBlocks;  // gp_binning.hpp: at most 2^24 blocks (256 MB) per map; a larger box keeps the hashed tables only

// geometry of the block grid from the block-unit bounding box; false when the box exceeds the budget
static bool grid_geometry(const int bbox[6], gp::GridGeom* g, long long* num_blocks) {
  double nb = 1.0;
  for (int a = 0; a < 3; a++) {
    g->lo[a] = bbox[a];
    const long long d = (long long)bbox[3 + a] - (long long)bbox[a] + 1;
    if (d <= 0 || d > (1ll << 30)) return false;
    g->dim[a] = (int)d;
    nb *= (double)d;
  }
  if (nb > (double)kMaxGridBlocks) return false;
  *num_blocks = (long long)g->dim[0] * g->dim[1] * g->dim[2];
  return true;
}

// device build of the occupancy-block grid: bounding box -> occupancy bits -> prefix sums -> voxels renumbered in
// (block, bit) order.  On return voxel_coords and the reference bucket table carry the NEW numbering (nothing else has been
// computed per voxel yet).  Synchronises the stream.
static int build_grid_device(gp_voxelmap* m, hipStream_t s) {
  m->has_grid = false;
  m->gblocks.release();
  const int V = m->info.num_voxels;
  if (V <= 0) return GP_OK;
  gp::DeviceArray d_bbox;
  GP_TRY(d_bbox.alloc(sizeof(int) * 6));
  int h_bbox[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  GP_HIP(hipMemcpyAsync(d_bbox.ptr, h_bbox, sizeof(h_bbox), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(gp::coords_bbox_kernel, dim3((V + 255) / 256), dim3(256), 0, s, V, m->voxel_coords.as<int>(), d_bbox.as<int>());
  GP_HIP(hipGetLastError());
  GP_HIP(hipMemcpyAsync(h_bbox, d_bbox.ptr, sizeof(h_bbox), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  gp::GridGeom g;
  long long nb = 0;
  if (!grid_geometry(h_bbox, &g, &nb)) return GP_OK;
  GP_TRY(m->gblocks.alloc(sizeof(gp::GridBlock) * (size_t)nb));
  GP_HIP(hipMemsetAsync(m->gblocks.ptr, 0, sizeof(gp::GridBlock) * (size_t)nb, s));
  gp::GridBlock* blocks = m->gblocks.as<gp::GridBlock>();
  hipLaunchKernelGGL(gp::grid_mark_kernel, dim3((V + 255) / 256), dim3(256), 0, s, V, m->voxel_coords.as<int>(), g, blocks);
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(gp::grid_count_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, nb, blocks);
  GP_HIP(hipGetLastError());
  gp::DeviceArray scratch, perm, coords_new;
  GP_TRY(scratch.alloc(sizeof(int) * gp::scan_scratch_ints(nb)));
  GP_TRY(perm.alloc(sizeof(int) * (size_t)V));
  GP_TRY(coords_new.alloc(sizeof(int) * 3 * (size_t)V));
  int* base0 = &blocks[0].base;
  GP_TRY(gp::exclusive_scan_strided(base0, 4, base0, 4, nb, scratch.as<int>(), s));
  hipLaunchKernelGGL(gp::grid_renumber_kernel, dim3((V + 255) / 256), dim3(256), 0, s, V, m->voxel_coords.as<int>(), g, (const gp::GridBlock*)blocks, perm.as<int>(),
                     coords_new.as<int>());
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(gp::buckets_renumber_kernel, dim3(grid_for((size_t)m->info.num_buckets)), dim3(kBlock), 0, s, (uint32_t)m->info.num_buckets,
                     m->buckets.as<gp_voxel_bucket>(), (const int*)perm.as<int>());
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize(s));  // perm / scratch die with this scope
  m->voxel_coords.swap(coords_new);
  for (int a = 0; a < 3; a++) {
    m->glo[a] = g.lo[a];
    m->gdim[a] = g.dim[a];
  }
  m->has_grid = true;
  return GP_OK;
}

// host build of the same structure (load / assign path): returns the permutation old index -> new index
static bool build_grid_host(int V, const int* coords, gp::GridGeom* g, std::vector<gp::GridBlock>* blocks, std::vector<int>* perm) {
  if (V <= 0) return false;
  int bbox[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int v = 0; v < V; v++)
    for (int a = 0; a < 3; a++) {
      bbox[a] = std::min(bbox[a], coords[3 * (size_t)v + a] >> 2);
      bbox[3 + a] = std::max(bbox[3 + a], coords[3 * (size_t)v + a] >> 2);
    }
  long long nb = 0;
  if (!grid_geometry(bbox, g, &nb)) return false;
  blocks->assign((size_t)nb, gp::GridBlock{0ull, 0, 0});
  for (int v = 0; v < V; v++) {
    const int* c = coords + 3 * (size_t)v;
    (*blocks)[(size_t)gp::grid_block_index(*g, c[0], c[1], c[2])].bits |= 1ull << gp::grid_bit(c[0], c[1], c[2]);
  }
  int run = 0;
  for (auto& b : *blocks) {
    b.base = run;
    run += __builtin_popcountll(b.bits);
  }
  if (run != V) return false;  // duplicate coordinates in the input: no canonical numbering
  perm->resize((size_t)V);
  for (int v = 0; v < V; v++) {
    const int* c = coords + 3 * (size_t)v;
    const gp::GridBlock& b = (*blocks)[(size_t)gp::grid_block_index(*g, c[0], c[1], c[2])];
    (*perm)[(size_t)v] = b.base + __builtin_popcountll(b.bits & ((1ull << gp::grid_bit(c[0], c[1], c[2])) - 1ull));
  }
  return true;
}


static inline gp_voxelmap* ext(gp_voxelmap* m) { return m; }
static inline const gp_voxelmap* ext(const gp_voxelmap* m) { return m; }

extern "C" {

int gp_voxelmap_create(double resolution, int init_num_buckets, int max_bucket_scan_count, double target_points_drop_rate, gp_stream_t stream,
                       gp_voxelmap_t** out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_create: null out");
  if (!(resolution > 0.0) || init_num_buckets <= 0 || max_bucket_scan_count <= 0)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_create: resolution, init_num_buckets and max_bucket_scan_count must be positive");
  auto* m = new gp_voxelmap;
  m->resolution = resolution;
  m->init_num_buckets = init_num_buckets;
  m->target_points_drop_rate = target_points_drop_rate;
  m->stream = (hipStream_t)stream;
  m->info.num_voxels = 0;
  m->info.num_buckets = init_num_buckets;
  m->info.max_bucket_scan_count = max_bucket_scan_count;
  m->info.voxel_resolution = (float)resolution;
  *out = m;
  return GP_OK;
}

int gp_voxelmap_destroy(gp_voxelmap_t* map) {
  if (!map) return GP_OK;
  // the arrays go back to the device's memory pool: nothing may still be reading them (what hipFree would have waited for)
  (void)hipDeviceSynchronize();
  delete ext(map);
  return GP_OK;
}

// the binned build (default): gp_binning.hpp gives the occupancy-block grid, the voxel numbering (block order) and the points
// sorted by voxel (stable); the statistics are an ordered segmented sum, the reference-visible bucket table is filled from the
// list of distinct voxels.  Deterministic: two builds of the same cloud give bit-identical maps.
static int insert_binned(gp_voxelmap* m, gp::PointBins& bins, const float* points_dev, const float* covs_dev, const float* intensities_dev, hipStream_t s) {
  const int V = bins.num_cells;
  m->info.num_voxels = V;
  GP_TRY(alloc_voxel_arrays(m, V));
  GP_TRY(m->voxel_coords.alloc_pooled(sizeof(int) * 3 * (size_t)std::max(V, 1), s));
  // the bucket table of the first insertion attempt (below) is allocated here already: the statistics kernel fills it with "empty" on its way (gp_host.hpp, FillJob)
  // the doubling sequence is entered where the voxels fit at a load factor <= 1/3: at 1/2 .. 2/3 some probe chain among 10^5 voxels exceeds max_bucket_scan_count almost
  // surely, and the failed attempt (fill + insertion + a synchronisation) was 45 us of the 2 M-point build (profiles/r04_build_timeline.txt)
  // (GP_TUNE_BUCKET_LOAD, per map: 33 % by default = up to twice the entries of round 3's 1.5 V, 16 B each -- 4.2 MB instead of 2.1 MB for the 73.7 k voxels of the
  // bench map; info.num_buckets / memory_usage_gpu report it.  The reference's own sequence starts at init_num_buckets and doubles until the drop rate is met.)
  int64_t num_buckets = m->init_num_buckets;
  while (num_buckets * (int64_t)m->bucket_load_percent < 100 * (int64_t)V) num_buckets *= 2;
  if (num_buckets > (int64_t(1) << 30)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_insert: bucket table would exceed 2^30 entries");
  GP_TRY(m->buckets.ensure_pooled(sizeof(gp_voxel_bucket) * (size_t)num_buckets, s));
  static_assert(sizeof(gp_voxel_bucket) == 16, "FillJob granules");
  hipLaunchKernelGGL(gp::segmented_stats_kernel, dim3((V + 15) / 16), dim3(256), 0, s, points_dev, covs_dev, intensities_dev, V, (const int*)bins.cell_start.as<int>(),
                     (const int*)bins.order.as<int>(), 1.0 / m->resolution, m->resolution, m->records.as<gp::VoxelRecord>(), m->num_points.as<int>(), m->voxel_means.as<float>(),
                     m->voxel_covs.as<float>(), m->voxel_intensities.as<float>(), m->voxel_coords.as<int>(),
                     gp::fill_job(m->buckets.ptr, sizeof(gp_voxel_bucket) * (size_t)num_buckets, 0xffffffffu));
  GP_HIP(hipGetLastError());
  // occupancy-block grid: taken over from the bins
  m->gblocks.swap(bins.blocks);
  for (int a = 0; a < 3; a++) {
    m->glo[a] = bins.geom.lo[a];
    m->gdim[a] = bins.geom.dim[a];
  }
  m->has_grid = true;
  // reference-visible bucket table (create_bucket_table, :253-307): the reference doubles the table until the fraction of points
  // whose probe chain is exhausted is <= target_points_drop_rate and then drops those points; here the table is doubled along the
  // same sequence until every voxel is placed, so no point is ever dropped (the CPU map, the parity target, drops none either)
  // (round 4: the "a voxel found no bucket" flag is a host-mapped word the kernel stores to -- no fill, no copy kernel; a table that turns out too small is the rare
  // path and starts over)
  gp::HostWords hw;
  GP_TRY(gp::HostWords::get(&hw));
  m->private_built = false;  // (the hashed kernel family's line table is built when a batch first asks for it: gp_voxelmap::ensure_private_table)
  m->plines.release();
  for (bool first = true;; num_buckets *= 2, first = false) {
    if (num_buckets > (int64_t(1) << 30)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_insert: bucket table would exceed 2^30 entries");
    if (!first) {  // (a table that turned out too small: the rare path)
      GP_TRY(m->buckets.ensure_pooled(sizeof(gp_voxel_bucket) * (size_t)num_buckets, s));
      GP_HIP(hipMemsetAsync(m->buckets.ptr, 0xff, sizeof(gp_voxel_bucket) * (size_t)num_buckets, s));
    }
    reinterpret_cast<volatile int*>(hw.host)[12] = 0;
    const uint32_t mask = ((num_buckets & (num_buckets - 1)) == 0) ? (uint32_t)(num_buckets - 1) : 0u;
    hipLaunchKernelGGL(gp::insert_voxels_kernel, dim3(grid_for((size_t)V)), dim3(kBlock), 0, s, V, (const int*)m->voxel_coords.as<int>(), (const int*)m->num_points.as<int>(),
                       m->buckets.as<gp_voxel_bucket>(), (uint32_t)num_buckets, mask, m->info.max_bucket_scan_count, hw.dev + 12);
    GP_HIP(hipGetLastError());
    m->info.num_buckets = (int)num_buckets;
    GP_TRY(hw.finish(s));  // :250 (the build is complete: a flag kernel behind it, polled -- gp_host.hpp)
    if (reinterpret_cast<volatile int*>(hw.host)[12] == 0) break;
  }
  return GP_OK;
}

int gp_voxelmap_insert(gp_voxelmap_t* map, const float* points_dev, const float* covs_dev, const float* intensities_dev, int n) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_insert: null map");
  if (!points_dev || !covs_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "error: GPU points/covs not allocated!!");  // gaussian_voxelmap_gpu.cu:212-215
  if (n < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_insert: negative size");
  auto* m = ext(map);
  hipStream_t s = m->stream;
  const double inv_leaf = 1.0 / m->resolution;
  // re-inserting into a map that already holds arrays releases them to the block cache ("any stream may take them"): kernels of other streams
  // that still read the old map must have finished.  A fresh map has nothing to wait for.
  if (m->buckets.ptr) GP_HIP(hipDeviceSynchronize());
  m->offloaded = false;
  m->generation++;
  if (n > 0 && !m->force_hashed_build) {
    gp::PointBins bins;
    bool too_large = false;
    GP_TRY(gp::bin_points(points_dev, n, inv_leaf, s, &bins, &too_large));
    if (!too_large && bins.num_cells > 0) return insert_binned(m, bins, points_dev, covs_dev, intensities_dev, s);
  }
  // ---- hashed build: an empty cloud, or a cloud whose bounding box is too large for the block grid.  The reference's own
  // scheme: claim buckets with atomicCAS, allocate voxel ids, accumulate with atomics (not bit-reproducible) ----

  // ---- create_bucket_table (:253-307): double the table until the drop rate is met ----
  gp::DeviceArray rep, counters;
  GP_TRY(counters.alloc(sizeof(int) * 2));
  int h_counters[2] = {0, 0};
  // the reference starts at init_num_buckets and doubles (one full pass over the points per attempt); a table needs more slots
  // than voxels and a LiDAR map rarely has fewer voxels than points / 16, so the doubling sequence is entered further up when
  // the cloud is large (same sequence, fewer wasted passes: 5 -> 2 for the 2 M-point bench map)
  int64_t num_buckets = m->init_num_buckets;
  while (num_buckets < n / 16) num_buckets *= 2;
  for (;; num_buckets *= 2) {
    if (num_buckets > (int64_t(1) << 30)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_insert: bucket table would exceed 2^30 entries");
    GP_TRY(rep.ensure(sizeof(int) * (size_t)num_buckets));
    GP_HIP(hipMemsetAsync(rep.ptr, 0xff, sizeof(int) * (size_t)num_buckets, s));
    GP_HIP(hipMemsetAsync(counters.ptr, 0, sizeof(int) * 2, s));
    const uint32_t mask = ((num_buckets & (num_buckets - 1)) == 0) ? (uint32_t)(num_buckets - 1) : 0u;
    if (n > 0) {
      hipLaunchKernelGGL(gp::claim_buckets_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, points_dev, n, rep.as<int>(), (uint32_t)num_buckets, mask,
                         m->info.max_bucket_scan_count, inv_leaf, counters.as<int>() + 1);
      GP_HIP(hipGetLastError());
    }
    GP_HIP(hipMemcpyAsync(h_counters, counters.ptr, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    if (h_counters[1] == 0 || (double)h_counters[1] / (double)n <= m->target_points_drop_rate) break;  // :288
  }
  m->info.num_buckets = (int)num_buckets;
  GP_TRY(m->buckets.alloc(sizeof(gp_voxel_bucket) * (size_t)num_buckets));
  // upper bound of the voxel count = number of claimed buckets <= min(n, num_buckets)
  const size_t max_voxels = (size_t)std::min<int64_t>(std::max(n, 1), num_buckets);
  GP_TRY(m->voxel_coords.alloc(sizeof(int) * 3 * max_voxels));
  hipLaunchKernelGGL(gp::assign_voxels_kernel, dim3(grid_for((size_t)num_buckets)), dim3(kBlock), 0, s, points_dev, rep.as<int>(), (uint32_t)num_buckets,
                     inv_leaf, m->buckets.as<gp_voxel_bucket>(), m->voxel_coords.as<int>(), counters.as<int>());
  GP_HIP(hipGetLastError());
  GP_HIP(hipMemcpyAsync(h_counters, counters.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  const int V = h_counters[0];
  m->info.num_voxels = V;
  // occupancy-block grid + canonical voxel numbering (block order); the bucket table and voxel_coords follow it
  GP_TRY(build_grid_device(m, s));

  // ---- accumulate + finalize (:218-250) ----
  GP_TRY(alloc_voxel_arrays(m, V));
  gp::DeviceArray sums;
  GP_TRY(sums.alloc(sizeof(double) * 9 * (size_t)std::max(V, 1)));
  GP_HIP(hipMemsetAsync(sums.ptr, 0, sizeof(double) * 9 * (size_t)std::max(V, 1), s));
  GP_HIP(hipMemsetAsync(m->num_points.ptr, 0, sizeof(int) * (size_t)std::max(V, 1), s));
  GP_HIP(hipMemsetAsync(m->voxel_intensities.ptr, 0, sizeof(float) * (size_t)std::max(V, 1), s));
  if (n > 0 && V > 0) {
    hipLaunchKernelGGL(gp::accumulate_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, points_dev, covs_dev, intensities_dev, n, m->view(), sums.as<double>(),
                       m->num_points.as<int>(), m->voxel_intensities.as<unsigned int>());
    GP_HIP(hipGetLastError());
    hipLaunchKernelGGL(gp::finalize_kernel, dim3(grid_for(V)), dim3(kBlock), 0, s, V, m->resolution, m->voxel_coords.as<int>(), sums.as<double>(),
                       m->num_points.as<int>(), m->records.as<gp::VoxelRecord>(), m->voxel_means.as<float>(), m->voxel_covs.as<float>());
    GP_HIP(hipGetLastError());
  }
  GP_TRY(build_private_table(m, s));
  GP_HIP(hipStreamSynchronize(s));  // :250
  return GP_OK;
}

int gp_voxelmap_info_get(const gp_voxelmap_t* map, gp_voxelmap_info* info) {
  if (!map || !info) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_info_get: null");
  *info = map->info;
  return GP_OK;
}

double gp_voxelmap_resolution(const gp_voxelmap_t* map) { return map ? map->resolution : 0.0; }

int gp_voxelmap_views_get(const gp_voxelmap_t* map, gp_voxelmap_views* views) {
  if (!map || !views) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_views_get: null");
  const bool on = map->loaded();
  views->buckets = on ? map->buckets.as<gp_voxel_bucket>() : nullptr;
  views->num_points = on ? map->num_points.as<int>() : nullptr;
  views->voxel_means = on ? map->voxel_means.as<float>() : nullptr;
  views->voxel_covs = on ? map->voxel_covs.as<float>() : nullptr;
  views->voxel_intensities = on ? map->voxel_intensities.as<float>() : nullptr;
  return GP_OK;
}

int gp_voxelmap_download(const gp_voxelmap_t* map, gp_voxel_bucket* buckets, int* num_points, float* means, float* covs, float* intensities) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_download: null map");
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "gp_voxelmap_download: voxel map is not on the GPU");
  const size_t V = (size_t)map->info.num_voxels, B = (size_t)map->info.num_buckets;
  hipStream_t s = map->stream;
  if (buckets) GP_HIP(hipMemcpyAsync(buckets, map->buckets.ptr, sizeof(gp_voxel_bucket) * B, hipMemcpyDeviceToHost, s));
  if (num_points && V) GP_HIP(hipMemcpyAsync(num_points, map->num_points.ptr, sizeof(int) * V, hipMemcpyDeviceToHost, s));
  if (means && V) GP_HIP(hipMemcpyAsync(means, map->voxel_means.ptr, sizeof(float) * 3 * V, hipMemcpyDeviceToHost, s));
  if (covs && V) GP_HIP(hipMemcpyAsync(covs, map->voxel_covs.ptr, sizeof(float) * 9 * V, hipMemcpyDeviceToHost, s));
  if (intensities && V) GP_HIP(hipMemcpyAsync(intensities, map->voxel_intensities.ptr, sizeof(float) * V, hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

int gp_voxelmap_download_f64(const gp_voxelmap_t* map, int* coords, int* num_points, double* means, double* covs) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_download_f64: null map");
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "gp_voxelmap_download_f64: voxel map is not on the GPU");
  const size_t V = (size_t)map->info.num_voxels;
  if (V == 0) return GP_OK;
  std::vector<gp::VoxelRecord> recs(V);
  std::vector<int> h_coords(3 * V);
  hipStream_t s = map->stream;
  GP_HIP(hipMemcpyAsync(recs.data(), map->records.ptr, sizeof(gp::VoxelRecord) * V, hipMemcpyDeviceToHost, s));
  GP_HIP(hipMemcpyAsync(h_coords.data(), ext(map)->voxel_coords.ptr, sizeof(int) * 3 * V, hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  for (size_t v = 0; v < V; v++) {
    if (coords) memcpy(coords + 3 * v, h_coords.data() + 3 * v, sizeof(int) * 3);
    if (num_points) num_points[v] = recs[v].num_points;
    if (means)
      for (int k = 0; k < 3; k++) means[3 * v + k] = ((double)h_coords[3 * v + k] + 0.5) * map->resolution + (double)recs[v].mean_local[k];
    if (covs) {
      double* c = covs + 9 * v;
      const double* r = recs[v].cov;
      c[0] = r[0];
      c[1] = c[3] = r[1];
      c[2] = c[6] = r[2];
      c[4] = r[3];
      c[5] = c[7] = r[4];
      c[8] = r[5];
    }
  }
  return GP_OK;
}

// host half of GaussianVoxelMapGPU::load (gaussian_voxelmap_gpu.cu:413-466): probe limit = max_bucket_scan_count,
// table starts at 8192*4 buckets and doubles until every voxel fits
int gp_voxelmap_assign(gp_voxelmap_t* map, int num_voxels, const int* coords, const int* num_points, const float* means, const float* covs6,
                       const float* intensities) {
  if (!map || num_voxels < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_assign: bad arguments");
  if (num_voxels > 0 && (!coords || !num_points || !means || !covs6)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_assign: null arrays");
  auto* m = ext(map);
  const size_t V = (size_t)num_voxels;
  // occupancy-block grid + canonical (block-order) numbering: the input arrays are permuted before anything is derived from them
  gp::GridGeom geom{};
  std::vector<gp::GridBlock> h_blocks;
  std::vector<int> perm, p_coords, p_np;
  std::vector<float> p_means, p_covs6, p_int;
  const bool grid = build_grid_host(num_voxels, coords, &geom, &h_blocks, &perm);
  if (grid) {
    p_coords.resize(3 * V);
    p_np.resize(V);
    p_means.resize(3 * V);
    p_covs6.resize(6 * V);
    if (intensities) p_int.resize(V);
    for (size_t v = 0; v < V; v++) {
      const size_t nv = (size_t)perm[v];
      memcpy(p_coords.data() + 3 * nv, coords + 3 * v, sizeof(int) * 3);
      p_np[nv] = num_points[v];
      memcpy(p_means.data() + 3 * nv, means + 3 * v, sizeof(float) * 3);
      memcpy(p_covs6.data() + 6 * nv, covs6 + 6 * v, sizeof(float) * 6);
      if (intensities) p_int[nv] = intensities[v];
    }
    coords = p_coords.data();
    num_points = p_np.data();
    means = p_means.data();
    covs6 = p_covs6.data();
    if (intensities) intensities = p_int.data();
  }
  std::vector<gp_voxel_bucket> h_buckets;
  const int max_scan = m->info.max_bucket_scan_count;
  auto assign_buckets = [&](int64_t nb) {
    h_buckets.assign((size_t)nb, gp_voxel_bucket{{0, 0, 0}, -1});
    const uint32_t mask = ((nb & (nb - 1)) == 0) ? (uint32_t)(nb - 1) : 0u;
    for (size_t i = 0; i < V; i++) {
      const uint64_t hash = gp::coord_hash(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
      bool inserted = false;
      for (int j = 0; j < max_scan; j++) {
        auto& b = h_buckets[gp::bucket_index(hash, j, (uint32_t)nb, mask)];
        if (b.voxel_index < 0) {
          b.coord[0] = coords[3 * i];
          b.coord[1] = coords[3 * i + 1];
          b.coord[2] = coords[3 * i + 2];
          b.voxel_index = (int)i;
          inserted = true;
          break;
        }
      }
      if (!inserted) return false;
    }
    return true;
  };
  int64_t nb = 8192 * 4;
  for (; nb < (int64_t(1) << 30); nb *= 2)
    if (assign_buckets(nb)) break;
  if (nb >= (int64_t(1) << 30)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_assign: could not build the bucket table");

  std::vector<gp::VoxelRecord> recs(V);
  std::vector<float> h_covs(9 * V), h_int(V, 0.0f);
  for (size_t v = 0; v < V; v++) {
    for (int k = 0; k < 3; k++) {
      const double centre = ((double)coords[3 * v + k] + 0.5) * m->resolution;
      recs[v].mean_local[k] = (float)((double)means[3 * v + k] - centre);
    }
    recs[v].num_points = num_points[v];
    for (int k = 0; k < 6; k++) recs[v].cov[k] = (double)covs6[6 * v + k];
    float* c = h_covs.data() + 9 * v;
    const float* r = covs6 + 6 * v;
    c[0] = r[0];
    c[1] = c[3] = r[1];
    c[2] = c[6] = r[2];
    c[4] = r[3];
    c[5] = c[7] = r[4];
    c[8] = r[5];
    if (intensities) h_int[v] = intensities[v];
  }
  m->info.num_voxels = num_voxels;
  m->info.num_buckets = (int)nb;
  m->offloaded = false;
  m->generation++;
  hipStream_t s = m->stream;
  GP_TRY(m->buckets.alloc(sizeof(gp_voxel_bucket) * (size_t)nb));
  GP_TRY(alloc_voxel_arrays(m, num_voxels));
  GP_TRY(m->voxel_coords.alloc(sizeof(int) * 3 * std::max<size_t>(V, 1)));
  GP_HIP(hipMemcpyAsync(m->buckets.ptr, h_buckets.data(), sizeof(gp_voxel_bucket) * (size_t)nb, hipMemcpyHostToDevice, s));
  if (V) {
    GP_HIP(hipMemcpyAsync(m->records.ptr, recs.data(), sizeof(gp::VoxelRecord) * V, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(m->num_points.ptr, num_points, sizeof(int) * V, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(m->voxel_means.ptr, means, sizeof(float) * 3 * V, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(m->voxel_covs.ptr, h_covs.data(), sizeof(float) * 9 * V, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(m->voxel_intensities.ptr, h_int.data(), sizeof(float) * V, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(m->voxel_coords.ptr, coords, sizeof(int) * 3 * V, hipMemcpyHostToDevice, s));
  }
  m->has_grid = false;
  m->gblocks.release();
  if (grid) {
    GP_TRY(m->gblocks.alloc(sizeof(gp::GridBlock) * h_blocks.size()));
    GP_HIP(hipMemcpyAsync(m->gblocks.ptr, h_blocks.data(), sizeof(gp::GridBlock) * h_blocks.size(), hipMemcpyHostToDevice, s));
    for (int a = 0; a < 3; a++) {
      m->glo[a] = geom.lo[a];
      m->gdim[a] = geom.dim[a];
    }
    m->has_grid = true;
  }
  GP_HIP(hipStreamSynchronize(s));  // the staging vectors die with this scope
  m->private_built = false;
  if (!m->has_grid) {
    GP_TRY(build_private_table(m, s));
    GP_HIP(hipStreamSynchronize(s));
  }
  return GP_OK;
}

int gp_voxelmap_save_compact(const gp_voxelmap_t* map, const char* path) {
  if (!map || !path) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_save_compact: null");
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "gp_voxelmap_save_compact: voxel map is not on the GPU");
  const size_t V = (size_t)map->info.num_voxels;
  std::vector<int> h_coords(3 * V), h_np(V);
  std::vector<float> h_means(3 * V), h_covs(9 * V), h_int(V);
  hipStream_t s = map->stream;
  if (V) {
    GP_HIP(hipMemcpyAsync(h_coords.data(), ext(map)->voxel_coords.ptr, sizeof(int) * 3 * V, hipMemcpyDeviceToHost, s));
    GP_TRY(gp_voxelmap_download(map, nullptr, h_np.data(), h_means.data(), h_covs.data(), h_int.data()));
  }
  std::vector<VoxelDataRecord> serial(V);
  for (size_t v = 0; v < V; v++) {
    auto& r = serial[v];
    memcpy(r.coord, h_coords.data() + 3 * v, sizeof(int) * 3);
    r.num_points = h_np[v];
    memcpy(r.mean, h_means.data() + 3 * v, sizeof(float) * 3);
    const float* c = h_covs.data() + 9 * v;
    r.cov[0] = c[0];
    r.cov[1] = c[3];
    r.cov[2] = c[6];
    r.cov[3] = c[4];
    r.cov[4] = c[7];
    r.cov[5] = c[8];
    r.intensity = h_int[v];
  }
  std::ofstream ofs(path, std::ios::binary);
  if (!ofs) return gp::fail(GP_ERROR_IO, std::string("error: failed to open ") + path);
  // header of gaussian_voxelmap_gpu.cu:358-366
  ofs << "compact " << 1 << std::endl;
  ofs << "resolution " << map->resolution << std::endl;
  ofs << "lru_count " << 0 << std::endl;
  ofs << "lru_cycle " << 1 << std::endl;
  ofs << "lru_thresh " << 1 << std::endl;
  ofs << "voxel_bytes " << sizeof(VoxelDataRecord) << std::endl;
  ofs << "num_voxels " << serial.size() << std::endl;
  ofs.write(reinterpret_cast<const char*>(serial.data()), (std::streamsize)(sizeof(VoxelDataRecord) * serial.size()));
  return ofs ? GP_OK : gp::fail(GP_ERROR_IO, std::string("error: failed to write ") + path);
}

int gp_voxelmap_load(const char* path, gp_stream_t stream, gp_voxelmap_t** out) {
  if (!path || !out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_load: null");
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs) return gp::fail(GP_ERROR_IO, std::string("error: failed to open ") + path);
  std::string token;
  bool compact = false;
  double resolution = 0.0;
  int lru = 0, voxel_bytes = 0, num_voxels = 0;
  ifs >> token >> compact;
  ifs >> token >> resolution;
  ifs >> token >> lru;
  ifs >> token >> lru;
  ifs >> token >> lru;
  ifs >> token >> voxel_bytes;
  ifs >> token >> num_voxels;
  std::getline(ifs, token);
  if (!ifs || voxel_bytes != (int)sizeof(VoxelDataRecord) || num_voxels < 0) return gp::fail(GP_ERROR_IO, std::string("error: malformed voxel map file ") + path);
  std::vector<VoxelDataRecord> flat((size_t)num_voxels);
  ifs.read(reinterpret_cast<char*>(flat.data()), (std::streamsize)(sizeof(VoxelDataRecord) * flat.size()));
  if (!ifs && num_voxels > 0) return gp::fail(GP_ERROR_IO, std::string("error: truncated voxel map file ") + path);
  const size_t V = flat.size();
  std::vector<int> coords(3 * V), np(V);
  std::vector<float> means(3 * V), covs6(6 * V), ints(V);
  for (size_t v = 0; v < V; v++) {
    memcpy(coords.data() + 3 * v, flat[v].coord, sizeof(int) * 3);
    np[v] = flat[v].num_points;
    memcpy(means.data() + 3 * v, flat[v].mean, sizeof(float) * 3);
    memcpy(covs6.data() + 6 * v, flat[v].cov, sizeof(float) * 6);
    ints[v] = flat[v].intensity;
  }
  gp_voxelmap_t* m = nullptr;
  GP_TRY(gp_voxelmap_create(resolution, 8192, 10, 0.1, stream, &m));  // :451
  int rc = gp_voxelmap_assign(m, num_voxels, coords.data(), np.data(), means.data(), covs6.data(), ints.data());
  if (rc != GP_OK) {
    gp_voxelmap_destroy(m);
    return rc;
  }
  *out = m;
  return GP_OK;
}

// replica of a map on another device (multi-GPU sharding: a target map referenced from several shards).  Arrays travel
// device-to-device with hipMemcpyPeerAsync; the clone owns its memory and carries the same voxel numbering.
int gp_voxelmap_clone_to_device(const gp_voxelmap_t* map, int device, gp_stream_t stream_on_device, gp_voxelmap_t** out) {
  if (!map || !out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_clone_to_device: null");
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "gp_voxelmap_clone_to_device: voxel map is not on the GPU");
  int src_device = 0, saved = 0;
  GP_HIP(hipGetDevice(&saved));
  {
    hipPointerAttribute_t attr{};
    src_device = (hipPointerGetAttributes(&attr, map->buckets.ptr) == hipSuccess) ? attr.device : saved;
    (void)hipGetLastError();
  }
  GP_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream_on_device;
  auto* m = new gp_voxelmap;
  m->resolution = map->resolution;
  m->init_num_buckets = map->init_num_buckets;
  m->target_points_drop_rate = map->target_points_drop_rate;
  m->stream = s;
  m->info = map->info;
  m->plmask = map->plmask;
  m->has_grid = map->has_grid;
  for (int a = 0; a < 3; a++) {
    m->glo[a] = map->glo[a];
    m->gdim[a] = map->gdim[a];
  }
  int rc = GP_OK;
  auto copy = [&](gp::DeviceArray& dst, const gp::DeviceArray& src, size_t bytes) {
    if (rc != GP_OK) return;
    if ((rc = dst.alloc(bytes)) != GP_OK) return;
    if (bytes == 0) return;
    const hipError_t e = hipMemcpyPeerAsync(dst.ptr, device, src.ptr, src_device, bytes, s);
    if (e != hipSuccess) rc = gp::hip_fail(e, "hipMemcpyPeerAsync", __FILE__, __LINE__);
  };
  const size_t V = (size_t)map->info.num_voxels, B = (size_t)map->info.num_buckets;
  copy(m->buckets, map->buckets, sizeof(gp_voxel_bucket) * B);
  copy(m->records, map->records, sizeof(gp::VoxelRecord) * V);
  copy(m->num_points, map->num_points, sizeof(int) * V);
  copy(m->voxel_means, map->voxel_means, sizeof(float) * 3 * V);
  copy(m->voxel_covs, map->voxel_covs, sizeof(float) * 9 * V);
  copy(m->voxel_intensities, map->voxel_intensities, sizeof(float) * V);
  copy(m->voxel_coords, map->voxel_coords, sizeof(int) * 3 * V);
  if (map->private_built) copy(m->plines, map->plines, 64 * ((size_t)map->plmask + 1));
  m->private_built = map->private_built;
  if (map->has_grid) copy(m->gblocks, map->gblocks, sizeof(gp::GridBlock) * (size_t)map->gdim[0] * map->gdim[1] * map->gdim[2]);
  if (rc == GP_OK) {
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = gp::hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
  }
  (void)hipSetDevice(saved);
  if (rc != GP_OK) {
    delete m;
    return rc;
  }
  m->generation = 1;
  *out = m;
  return GP_OK;
}

size_t gp_voxelmap_memory_usage_gpu(const gp_voxelmap_t* map) {
  if (!map) return 0;
  // reference formula (gaussian_voxelmap_gpu.cu:469-472) + the gather records, coordinates and line table this implementation adds
  return (size_t)map->info.num_voxels * (sizeof(int) + sizeof(float) * 3 + sizeof(float) * 9 + sizeof(gp::VoxelRecord) + sizeof(int) * 3) +
         (size_t)map->info.num_buckets * sizeof(gp_voxel_bucket) + (map->private_built ? ((size_t)map->plmask + 1) * 4 * sizeof(gp_voxel_bucket) : 0) +
         (map->has_grid ? (size_t)map->gdim[0] * map->gdim[1] * map->gdim[2] * sizeof(gp::GridBlock) : 0);
}

int gp_voxelmap_loaded_on_gpu(const gp_voxelmap_t* map) { return map && map->loaded() ? 1 : 0; }
int gp_voxelmap_has_block_grid(const gp_voxelmap_t* map) { return map && map->has_grid ? 1 : 0; }
// per map: GP_TUNE_MAP_BUILD = 1 builds with the reference-shaped hashed scheme (the fallback of clouds too large for the block grid; A/B and tests)
int gp_voxelmap_set_tuning(gp_voxelmap_t* map, int key, int value) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_set_tuning: null map");
  if (key == GP_TUNE_BUCKET_LOAD) {
    if (value < 5 || value > 90) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_set_tuning: GP_TUNE_BUCKET_LOAD is a load factor in per cent, 5 .. 90");
    ext(map)->bucket_load_percent = value;
    return GP_OK;
  }
  if (key != GP_TUNE_MAP_BUILD) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_set_tuning: the keys of a voxel map are GP_TUNE_MAP_BUILD and GP_TUNE_BUCKET_LOAD");
  map->force_hashed_build = value != 0;
  return GP_OK;
}

static int to_host(std::vector<char>& dst, const gp::DeviceArray& src, size_t bytes, hipStream_t s) {
  dst.resize(bytes);
  if (bytes) GP_HIP(hipMemcpyAsync(dst.data(), src.ptr, bytes, hipMemcpyDeviceToHost, s));
  return GP_OK;
}

static int to_device(gp::DeviceArray& dst, const std::vector<char>& src, hipStream_t s) {
  GP_TRY(dst.alloc(src.size()));
  if (!src.empty()) GP_HIP(hipMemcpyAsync(dst.ptr, src.data(), src.size(), hipMemcpyHostToDevice, s));
  return GP_OK;
}

int gp_voxelmap_offload(gp_voxelmap_t* map, gp_stream_t stream) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_offload: null");
  auto* m = ext(map);
  if (!m->loaded()) return GP_ERROR_NOT_LOADED;  // reference returns false (:477-479)
  hipStream_t s = (hipStream_t)stream;
  const size_t V = (size_t)m->info.num_voxels, B = (size_t)m->info.num_buckets;
  GP_TRY(to_host(m->h_buckets, m->buckets, sizeof(gp_voxel_bucket) * B, s));
  GP_TRY(to_host(m->h_records, m->records, sizeof(gp::VoxelRecord) * V, s));
  GP_TRY(to_host(m->h_num_points, m->num_points, sizeof(int) * V, s));
  GP_TRY(to_host(m->h_means, m->voxel_means, sizeof(float) * 3 * V, s));
  GP_TRY(to_host(m->h_covs, m->voxel_covs, sizeof(float) * 9 * V, s));
  GP_TRY(to_host(m->h_intensities, m->voxel_intensities, sizeof(float) * V, s));
  GP_TRY(to_host(m->h_coords, m->voxel_coords, sizeof(int) * 3 * V, s));
  const size_t G = m->has_grid ? (size_t)m->gdim[0] * m->gdim[1] * m->gdim[2] : 0;
  GP_TRY(to_host(m->h_gblocks, m->gblocks, sizeof(gp::GridBlock) * G, s));
  // the arrays go back to the per-thread block cache tagged "any stream may take them": a factor kernel on ANOTHER stream may still be reading
  // the map (the caller offloads while a linearise is in flight on the factor's stream), so every stream of the device is drained first -- what
  // hipFree used to imply; an offload is a rare, millisecond-scale operation
  GP_HIP(hipDeviceSynchronize());
  m->buckets.release();
  m->records.release();
  m->num_points.release();
  m->voxel_means.release();
  m->voxel_covs.release();
  m->voxel_intensities.release();
  m->voxel_coords.release();
  m->plines.release();
  m->private_built = false;
  m->gblocks.release();
  m->offloaded = true;
  m->generation++;
  return GP_OK;
}

int gp_voxelmap_reload(gp_voxelmap_t* map, gp_stream_t stream) {
  if (!map) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_reload: null");
  auto* m = ext(map);
  if (!m->offloaded) return GP_ERROR_INVALID_ARGUMENT;  // already on the GPU: reference returns false (:509-511)
  hipStream_t s = (hipStream_t)stream;
  GP_TRY(to_device(m->buckets, m->h_buckets, s));
  GP_TRY(to_device(m->records, m->h_records, s));
  GP_TRY(to_device(m->num_points, m->h_num_points, s));
  GP_TRY(to_device(m->voxel_means, m->h_means, s));
  GP_TRY(to_device(m->voxel_covs, m->h_covs, s));
  GP_TRY(to_device(m->voxel_intensities, m->h_intensities, s));
  GP_TRY(to_device(m->voxel_coords, m->h_coords, s));
  if (m->has_grid) GP_TRY(to_device(m->gblocks, m->h_gblocks, s));
  m->private_built = false;
  if (!m->has_grid) GP_TRY(build_private_table(m, s));
  GP_HIP(hipStreamSynchronize(s));
  m->offloaded = false;
  m->generation++;
  return GP_OK;
}

int gp_voxelmap_lookup(const gp_voxelmap_t* map, const float* points_dev, const float* normals_dev, int n, const double delta[16], int* voxel_indices_dev,
                       gp_stream_t stream) {
  if (!map || !points_dev || !delta || !voxel_indices_dev || n < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_lookup: bad arguments");
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "gp_voxelmap_lookup: voxel map is not on the GPU");
  if (n == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  gp::DeviceArray pose;
  GP_TRY(pose.alloc(sizeof(double) * 16));
  GP_HIP(hipMemcpyAsync(pose.ptr, delta, sizeof(double) * 16, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(gp::lookup_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, points_dev, normals_dev, n, map->view(), pose.as<double>(), voxel_indices_dev,
                     (int*)nullptr);
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

int gp_voxelmap_overlap(const gp_voxelmap_t* map, const float* points_dev, int n, const double delta[16], int* num_hits, gp_stream_t stream) {
  if (!map || !points_dev || !delta || !num_hits || n < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_voxelmap_overlap: bad arguments");
  if (!map->loaded()) return gp::fail(GP_ERROR_NOT_LOADED, "gp_voxelmap_overlap: voxel map is not on the GPU");
  *num_hits = 0;
  if (n == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  gp::DeviceArray scratch;
  GP_TRY(scratch.alloc(sizeof(double) * 16 + sizeof(int) * 4));
  int* d_count = reinterpret_cast<int*>(scratch.as<double>() + 16);
  GP_HIP(hipMemcpyAsync(scratch.ptr, delta, sizeof(double) * 16, hipMemcpyHostToDevice, s));
  GP_HIP(hipMemsetAsync(d_count, 0, sizeof(int), s));
  hipLaunchKernelGGL(gp::lookup_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, points_dev, (const float*)nullptr, n, map->view(), scratch.as<double>(),
                     (int*)nullptr, d_count);
  GP_HIP(hipGetLastError());
  GP_HIP(hipMemcpyAsync(num_hits, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

}  // extern "C"
