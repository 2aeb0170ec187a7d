// gp_knn.hip -- exact k-nearest-neighbour search on a cell-sorted point grid, covariance estimation, and a GICP
// linearisation that uses it (BASELINE.json configs[4]).
//
// Replaces (reference, CPU only -- there is no GPU counterpart upstream):
//   ann/small_kdtree.hpp:124-186,437-474 + ann/knn_result.hpp:89-109   exact k-NN (kd-tree)      -> uniform-grid shell search
//   features/covariance_estimation.cpp:18-77                          estimate_covariances      -> gp_estimate_covariances
//   factors/impl/integrated_gicp_factor_impl.hpp:132-296              GICP correspondences+H/b  -> gp_gicp_factor_*
//
// Exactness: a query visits the cells of growing cubes around its own cell and stops after radius r once it holds k
// neighbours whose k-th squared distance is <= d_safe(r)^2, where d_safe(r) = r*h + (distance from the query to the
// nearest face of its own cell) is a lower bound on the distance to every unvisited point.  Distances are computed in
// f64 on the f32 inputs (as the reference does on PointCloudCPU's doubles), so the neighbour SET equals the kd-tree's
// except for exact ties at the k-th distance (where the reference's own result depends on traversal order).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gp_binning.hpp"
#include "gp_scan.hpp"
#include "gp_host.hpp"
#include "gp_vgicp_tile.hpp"

namespace gp {

constexpr unsigned long long kEmptyKey = ~0ull;

__host__ __device__ __forceinline__ unsigned long long pack_cell(int x, int y, int z) {
  return ((unsigned long long)(unsigned)(x + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(y + (1 << 20)) << 21) | (unsigned long long)(unsigned)(z + (1 << 20));
}

// cell coordinate of the hashed grid: 21-bit fields.  Coordinates beyond +-2^20 cells are clamped to the border cells and a
// non-finite coordinate goes to cell 0: such points sit in a cell that is never FARTHER from a query than their true cell, so the
// shell search still meets them in time (distances always come from the real coordinates; NaN / inf distances are never selected)
__host__ __device__ __forceinline__ int hashed_cell(double u) {
  if (!(fabs(u) < 1.0e9)) return 0;
  const int c = fast_floor(u);
  const int lim = (1 << 20) - 3;
  return c < -lim ? -lim : (c > lim ? lim : c);
}

__host__ __device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

struct GridView {
  const unsigned long long* keys;  // [slots] packed cell coordinate or kEmptyKey
  const int* start;                // [slots + 1] first sorted point of the cell stored at this slot
  const float4* sorted;            // [n] (x, y, z, original index as int bits), cell-sorted
  uint32_t mask;
  int n;
  double inv_h, h;
  int lo[3], hi[3];  // bounding box of the occupied cells: bounds the cube radius of any query
};

__device__ __forceinline__ int grid_find(const GridView& g, unsigned long long key) {
  uint32_t s = hash_key(key) & g.mask;
  for (;;) {
    const unsigned long long k = g.keys[s];
    if (k == key) return (int)s;
    if (k == kEmptyKey) return -1;
    s = (s + 1) & g.mask;
  }
}

// ---- grid build -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grid_insert_kernel(const float* __restrict__ points, int n, double inv_h, unsigned long long* __restrict__ keys,
                                                          int* __restrict__ counts, int* __restrict__ point_slot, uint32_t mask, int* __restrict__ bbox) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = i < n ? i : n - 1;  // the tail lanes repeat the last point so that the wave-wide min/max below needs no masking
  const int cx = hashed_cell((double)points[3 * (size_t)j] * inv_h), cy = hashed_cell((double)points[3 * (size_t)j + 1] * inv_h),
            cz = hashed_cell((double)points[3 * (size_t)j + 2] * inv_h);
  // bounding box of the occupied cells (bounds every query's cube radius): wave min/max, one row per workgroup, reduced by
  // bbox_reduce_kernel (atomics on six shared words serialise ~100 k operations per level: measured 1 ms)
  int lo[3] = {cx, cy, cz}, hi[3] = {cx, cy, cz};
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
    bbox[6 * (size_t)blockIdx.x + threadIdx.x] = v;
  }
  if (i >= n) return;
  const unsigned long long key = pack_cell(cx, cy, cz);
  uint32_t s = hash_key(key) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&keys[s], kEmptyKey, key);
    if (old == kEmptyKey || old == key) break;
    s = (s + 1) & mask;
  }
  point_slot[i] = (int)s;
  atomicAdd(&counts[s], 1);
}

// per-workgroup boxes [nb][6] -> one box
__global__ void __launch_bounds__(256) bbox_reduce_kernel(const int* __restrict__ block_boxes, int nb, int* __restrict__ bbox) {
  __shared__ int part[256][6];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int b = threadIdx.x; b < nb; b += 256)
    for (int a = 0; a < 3; a++) {
      lo[a] = min(lo[a], block_boxes[6 * (size_t)b + a]);
      hi[a] = max(hi[a], block_boxes[6 * (size_t)b + 3 + a]);
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

// exclusive scan of counts[0..m) -> start[0..m], three small kernels (block sums, scan of block sums, add)
constexpr int kScanBlock = 1024;
__global__ void __launch_bounds__(kScanBlock) scan_block_kernel(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ block_sums, int m) {
  __shared__ int lds[kScanBlock];
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  const int v = i < m ? in[i] : 0;
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    const int t = threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
    __syncthreads();
    lds[threadIdx.x] += t;
    __syncthreads();
  }
  if (i < m) out[i] = lds[threadIdx.x] - v;  // exclusive
  if (threadIdx.x == kScanBlock - 1) block_sums[blockIdx.x] = lds[threadIdx.x];
}
__global__ void __launch_bounds__(kScanBlock) scan_sums_kernel(int* __restrict__ block_sums, int nb, int* __restrict__ total) {
  __shared__ int lds[kScanBlock];
  int carry = 0;
  for (int base = 0; base < nb; base += kScanBlock) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
      const int t = threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
      __syncthreads();
      lds[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) block_sums[i] = carry + lds[threadIdx.x] - v;
    const int last = lds[kScanBlock - 1];
    __syncthreads();
    carry += last;
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(kScanBlock) scan_add_kernel(int* __restrict__ out, const int* __restrict__ block_sums, int m, const int* __restrict__ total) {
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  if (i < m) out[i] += block_sums[blockIdx.x];
  if (i == 0) out[m] = *total;
}

__global__ void __launch_bounds__(256) grid_scatter_kernel(const float* __restrict__ points, int n, const int* __restrict__ point_slot, const int* __restrict__ start,
                                                           int* __restrict__ cursor, float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = point_slot[i];
  const int pos = start[s] + atomicAdd(&cursor[s], 1);
  sorted[pos] = make_float4(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2], __int_as_float(i));
}

// ---- exact k-NN -------------------------------------------------------------------------------------------------
template <int KMAX, bool FULL = false>  // FULL: the list always holds exactly KMAX neighbours (k == KMAX): straight-line insertion only
struct TopK {
  // d / idx are only ever indexed with compile-time constants (unrolled loops + predicates): a run-time index such as d[k - 1]
  // would send both arrays to scratch memory (160 B per lane for KMAX = 10) and turn every comparison into a memory access
  double d[KMAX];
  int idx[KMAX];
  double bound;  // = d[k - 1]: the current k-th distance (or the caller's max_sq_dist while fewer than k are held)
  int k, found;
  __device__ void init(int k_, double max_sq_dist) {
    k = k_;
    found = 0;
    bound = max_sq_dist;
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
      d[j] = max_sq_dist;
      idx[j] = -1;
    }
  }
  __device__ double worst() const { return bound; }
  // neighbours held.  FULL lists do not count their insertions (two instructions in the hottest block of the search): an entry is held iff its index is valid
  __device__ int count() const {
    if constexpr (FULL) {
      int c = 0;
#pragma unroll
      for (int j = 0; j < KMAX; j++) c += idx[j] >= 0 ? 1 : 0;
      return c;
    } else {
      return found;
    }
  }
  // KnnResult::push (ann/knn_result.hpp:89-109): strict '<', earlier-visited ties win
  __device__ void push(int index, double dist) {
    if (!(dist < bound)) return;
    if constexpr (FULL) {
      // full list (the common case: covariance estimation asks for exactly KMAX): straight-line code.  c[j] = dist < d[j] is monotone in j (the list is
      // sorted), the new entry j is d[j-1] where c[j-1], the candidate where c[j] alone, d[j] otherwise -- for the distances that is
      // max(d[j-1], min(dist, d[j])), for the indices two selects on the same masks: 10 compares + 20 min/max + 20 selects, no exec-mask regions
      // (the position-by-position form below compiles to ten of them plus a scalar branch tree for the bound: ~70 vector and ~80 scalar / branch
      // instructions per insertion, executed by the whole wave whenever one lane inserts)
      bool c[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; j++) c[j] = dist < d[j];
      // (v_min_f64 / v_max_f64 through asm: fmin / fmax make hipcc quiet every operand first -- `v_max_f64 x, x, x`, eleven more f64 instructions per
      // insertion -- and no operand here is a NaN: squared distances of finite points and the finite sentinel of init())
      auto min64 = [](double a, double b) {
        double r;
        asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
      };
      auto max64 = [](double a, double b) {
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
      };
#pragma unroll
      for (int j = KMAX - 1; j >= 1; j--) {
        idx[j] = c[j - 1] ? idx[j - 1] : (c[j] ? index : idx[j]);
        d[j] = max64(d[j - 1], min64(dist, d[j]));
      }
      idx[0] = c[0] ? index : idx[0];
      d[0] = min64(dist, d[0]);
      bound = d[KMAX - 1];
      return;  // (`found` is not kept up in this form: count() reads it off the list)
    }
    bool placed = false;
#pragma unroll
    for (int j = KMAX - 1; j >= 0; j--) {
      if (j < k && !placed) {
        if (j > 0 && dist < d[j - 1]) {
          d[j] = d[j - 1];
          idx[j] = idx[j - 1];
        } else {
          d[j] = dist;
          idx[j] = index;
          placed = true;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KMAX; j++)
      if (j == k - 1) bound = d[j];
    found = found + 1 < k ? found + 1 : k;
  }
};

// f32 filter bound for "exact squared distance < worst": the f32 differences are off by <= m per axis, so the f32 squared distance is at most
// worst (1 + 1e-6) + 4 sqrt(worst) m + 4 m^2, and 4 sqrt(w) m <= w / 1024 + 4096 m^2 (AM-GM) spares the square root -- it sat behind every insertion
// with its IEEE refinement, ~20 instructions; the filter admits candidates within 0.1 % of the bound instead, the f64 comparison decides as before
__device__ __forceinline__ float loosened_bound(double worst, float m2x4100) { return (float)worst * 1.000978f + m2x4100; }

// LiDAR density varies by three orders of magnitude between the near and the far field, so one cell size cannot be right
// everywhere: the structure keeps up to kMaxLevels grids (cell size x4 per level) and every query runs the same exact
// search on the finest level whose 3x3x3 neighbourhood already holds enough points.  Exactness does not depend on the choice.
constexpr int kMaxLevels = 3;
struct MultiGridView {
  GridView lv[kMaxLevels];
  int num_levels;
};

template <int KMAX, bool FULL>
__device__ __forceinline__ void knn_query(const GridView& g, double qx, double qy, double qz, TopK<KMAX, FULL>& top);

__device__ __forceinline__ int count27(const GridView& g, double qx, double qy, double qz) {
  const int cx = hashed_cell(qx * g.inv_h), cy = hashed_cell(qy * g.inv_h), cz = hashed_cell(qz * g.inv_h);
  int c = 0;
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int s = grid_find(g, pack_cell(cx + dx, cy + dy, cz + dz));
        if (s >= 0) c += g.start[s + 1] - g.start[s];
      }
  return c;
}

template <int KMAX, bool FULL>
__device__ __forceinline__ void knn_query_multi(const MultiGridView& mg, double qx, double qy, double qz, int want, TopK<KMAX, FULL>& top) {
  int level = mg.num_levels - 1;
  for (int l = 0; l + 1 < mg.num_levels; l++) {
    if (count27(mg.lv[l], qx, qy, qz) >= want) {
      level = l;
      break;
    }
  }
  knn_query<KMAX, FULL>(mg.lv[level], qx, qy, qz, top);
}

template <int KMAX, bool FULL>
__device__ __forceinline__ void knn_query(const GridView& g, double qx, double qy, double qz, TopK<KMAX, FULL>& top) {
  if (!(fabs(qx) < 1.0e300 && fabs(qy) < 1.0e300 && fabs(qz) < 1.0e300)) return;  // non-finite query: no neighbours
  const int cx = hashed_cell(qx * g.inv_h), cy = hashed_cell(qy * g.inv_h), cz = hashed_cell(qz * g.inv_h);
  // distance from the query to the nearest face of its own cell
  const double fx = qx * g.inv_h - (double)cx, fy = qy * g.inv_h - (double)cy, fz = qz * g.inv_h - (double)cz;
  const double face = fmin(fmin(fmin(fx, 1.0 - fx), fmin(fy, 1.0 - fy)), fmin(fz, 1.0 - fz)) * g.h;
  // cube radius after which every occupied cell has been visited from this query
  const int rmax = max(max(max(abs(cx - g.lo[0]), abs(cx - g.hi[0])), max(abs(cy - g.lo[1]), abs(cy - g.hi[1]))), max(abs(cz - g.lo[2]), abs(cz - g.hi[2])));
  for (int r = 0; r <= rmax; r++) {
    for (int dz = -r; dz <= r; dz++)
      for (int dy = -r; dy <= r; dy++) {
        const bool shell_yz = (dz == -r || dz == r || dy == -r || dy == r);
        const int step = (shell_yz || r == 0) ? 1 : 2 * r;  // interior rows: only dx = -r and dx = +r belong to the shell
        for (int dx = -r; dx <= r; dx += step) {
          const int s = grid_find(g, pack_cell(cx + dx, cy + dy, cz + dz));
          if (s < 0) continue;
          const int b = g.start[s], e = g.start[s + 1];
          for (int p = b; p < e; p++) {
            const float4 v = g.sorted[p];
            const double ddx = (double)v.x - qx, ddy = (double)v.y - qy, ddz = (double)v.z - qz;
            top.push(__float_as_int(v.w), ddx * ddx + ddy * ddy + ddz * ddz);
          }
        }
      }
    const double safe = (double)r * g.h + face;
    if (top.worst() <= safe * safe) return;  // every unvisited point is farther than the current k-th (or than max_sq_dist)
    if (top.count() >= g.n) return;            // the whole cloud has been seen (clouds smaller than k)
  }
}

// ---- search over the binned structure (gp_binning.hpp): occupancy-block grid over the cells + cell-sorted points -------------------
// A query walks the cube shells around its cell like knn_query above, but reads ONE 16-B block entry per 4 x 4 x 4 cells instead of
// probing a hash table per cell, visits only occupied cells (bit scan), and filters candidates with an f32 distance before the f64
// distance that decides (the reference compares doubles): the shell loop of a typical query touches <= 8 block entries.
struct BinGridView {
  const GridBlock* blocks;
  const int* cell_start;  // [num_cells + 1]
  const float4* sorted;   // [n] (x, y, z, original index as int bits), cell-major, ascending index inside a cell
  GridGeom geom;
  double inv_h, h;
  int n;  // binned (finite) points
  const unsigned long long* super;  // [sdim[2]][sdim[1]][sdim[0]] occupancy masks of 4 x 4 x 4 blocks, relative block coordinate >> 2
  int sdim[3];
  unsigned long long* counters;  // measurement build only (gp_debug_knn_counters): {queries, f32 distances, f64 distances, block entries, cells}
};

// 4-bit mask of the cells x = 4 * b + {0, 1, 2, 3} inside [c - r, c + r]
__device__ __forceinline__ unsigned axis_mask(int b, int c, int r) {
  int lo = c - r - 4 * b, hi = c + r - 4 * b;
  lo = lo < 0 ? 0 : lo;
  hi = hi > 3 ? 3 : hi;
  return lo > hi ? 0u : (((2u << hi) - 1u) & ~((1u << lo) - 1u));
}
// 64-bit cell mask of a block from its per-axis 4-bit masks (bit = z * 16 + y * 4 + x)
__device__ __forceinline__ unsigned long long cube_mask(unsigned mx, unsigned my, unsigned mz) {
  const unsigned long long X = (unsigned long long)mx * 0x1111111111111111ull;
  const unsigned y4 = (my & 1u) | ((my & 2u) << 3) | ((my & 4u) << 6) | ((my & 8u) << 9);  // bit y -> bit 4 y
  const unsigned long long Y = (unsigned long long)(y4 * 0xFu) * 0x0001000100010001ull;
  const unsigned long long z1 = (unsigned long long)mz;
  const unsigned long long Z = ((z1 | (z1 << 15) | (z1 << 30) | (z1 << 45)) & 0x0001000100010001ull) * 0xFFFFull;
  return X & Y & Z;
}

// max_shells: how many shells beyond the first one that reaches the box this call may walk before it gives up (returns false: the
// caller retries on a coarser level); returns true when the search is complete (bound met, or every point seen)
constexpr int kDeferShell = 2;      // (covariance search with the cooperative pass) a query with fewer than k points within this many cells of its cell is deferred
constexpr int kRangeCap = 16;       // candidate ranges a lane collects before it scans them (flat scan of knn_query_bins)
constexpr int kFlatWidth = 4;       // candidates whose loads a lane has in flight per trip of the flat scan (8: 16 registers spilled, 0.748 vs 0.754 ms per call: no gain)
constexpr int kRangeStride = 128;   // int2 entries between two slots of one lane's list = threads of the workgroups that use it

// (round 5: a query the fine shells cannot settle -- knn_query_any's `sparse` -- is not continued on the coarser levels lane by lane but handed to covariance_far_kernel)
template <int KMAX, bool FULL, bool FLAT = false>
__device__ __forceinline__ bool knn_query_bins(const BinGridView& g, double qx, double qy, double qz, TopK<KMAX, FULL>& top, int max_shells, int2* rl = nullptr,
                                               bool* sparse = nullptr) {
  const double ux = qx * g.inv_h, uy = qy * g.inv_h, uz = qz * g.inv_h;
  if (!(fabs(ux) < 1.0e9 && fabs(uy) < 1.0e9 && fabs(uz) < 1.0e9)) return true;  // non-finite query: no neighbours
  const int c[3] = {fast_floor(ux), fast_floor(uy), fast_floor(uz)};
  const double fx = ux - (double)c[0], fy = uy - (double)c[1], fz = uz - (double)c[2];
  const double face = fmin(fmin(fmin(fx, 1.0 - fx), fmin(fy, 1.0 - fy)), fmin(fz, 1.0 - fz)) * g.h;
  int lo[3], hi[3], r0 = 0, rmax = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = g.geom.lo[a] * 4;
    hi[a] = (g.geom.lo[a] + g.geom.dim[a]) * 4 - 1;
    r0 = max(r0, max(lo[a] - c[a], c[a] - hi[a]));              // first shell that reaches the box
    rmax = max(rmax, max(abs(c[a] - lo[a]), abs(c[a] - hi[a])));  // shell that covers it
  }
  const float qxf = (float)qx, qyf = (float)qy, qzf = (float)qz;
  const float fxf = (float)fx, fyf = (float)fy, fzf = (float)fz, h2f = (float)(g.h * g.h);  // (cell-box pruning below)
  // |f32 difference - exact difference| <= margin per axis (rounding of q to float + the subtraction), generously
  const float margin = (fabsf(qxf) + fabsf(qyf) + fabsf(qzf) + 1.0f) * 2.4e-7f;
  const float m2x4100 = 4100.0f * margin * margin;
  auto loosened = [&](double worst) { return loosened_bound(worst, m2x4100); };  // (+inf while fewer than k neighbours are held and no distance bound was given)
  float accept = loosened(top.worst());
  unsigned n_f32 = 0, n_f64 = 0, n_blk = 0, n_cell = 0;  // work counters: only read when g.counters is set (measurement runs)
  // candidates of one cell: the loads of four consecutive points are issued together (a lane's loads miss L1 more often than not, and
  // one round trip per point was the whole cost of this search), the tests follow in point order
  auto test_point = [&](const float4 v) {
    const float dxf = v.x - qxf, dyf = v.y - qyf, dzf = v.z - qzf;
    if (dxf * dxf + dyf * dyf + dzf * dzf <= accept) {
      n_f64++;
      const double ddx = (double)v.x - qx, ddy = (double)v.y - qy, ddz = (double)v.z - qz;
      top.push(__float_as_int(v.w), ddx * ddx + ddy * ddy + ddz * ddz);
      accept = loosened(top.worst());
    }
  };
  auto scan_range = [&](int pb, int pe) {
    int p = pb;
    for (; p + 4 <= pe; p += 4) {
      const float4 v0 = g.sorted[p], v1 = g.sorted[p + 1], v2 = g.sorted[p + 2], v3 = g.sorted[p + 3];
      test_point(v0);
      test_point(v1);
      test_point(v2);
      test_point(v3);
    }
    if (p < pe) {
      const int last = pe - 1;
      const float4 v0 = g.sorted[p], v1 = g.sorted[min(p + 1, last)], v2 = g.sorted[min(p + 2, last)];
      test_point(v0);
      if (p + 1 < pe) test_point(v1);
      if (p + 2 < pe) test_point(v2);
    }
  };
  // FLAT scan (rl != nullptr: a per-lane list of candidate ranges in LDS, kRangeCap entries, lane stride kRangeStride).  Scanning a cell the moment the walk
  // finds it keeps the lanes of a wave out of step -- they find their cells at different points of the nested block / cell loops, and the wave runs the point
  // loop once per (lane group, cell): ~590 executions of the candidate test per wave for ~185 candidates per lane.  With the list, a shell's cells are only
  // COLLECTED by the walk; then every lane streams through its ranges in one loop, four candidates per trip, all lanes busy until their own list ends.  A
  // lane's candidates keep their order, so the result is the same list, bit for bit.
  int rl_count = 0;
  auto flush_ranges = [&]() {
    int ri = 0, p = 0, pe = 0;
    auto next_range = [&]() {
      p = 0;
      pe = 0;
      while (ri < rl_count) {
        const int2 rg = rl[ri * kRangeStride];
        ri++;
        // round 4: a cell collected while the list was not full yet (or the bound still loose) is looked at again when its turn comes: by then a dense cell in front of
        // it has usually brought the k-th distance down to centimetres, and a cell whose box is farther than that holds nothing of interest -- the queries that needed
        // shell 1 because their own cell held fewer than k points used to scan all 26 neighbours in full (up to 465 candidates on a lane, the launch's longest waves)
        if (__int_as_float(rg.y & (int)0xffff0000) > accept) continue;
        p = rg.x;
        pe = rg.x + (rg.y & 0xffff);
        break;
      }
    };
    next_range();
    while (p < pe) {
      int a[kFlatWidth];
      bool k[kFlatWidth];
#pragma unroll
      for (int q = 0; q < kFlatWidth; q++) {
        a[q] = 0;
        k[q] = p < pe;
        if (k[q]) {
          a[q] = p;
          p++;
          if (p == pe) next_range();
        }
      }
      float4 v[kFlatWidth];
#pragma unroll
      for (int q = 0; q < kFlatWidth; q++) v[q] = g.sorted[a[q]];
#pragma unroll
      for (int q = 0; q < kFlatWidth; q++)
        if (k[q]) test_point(v[q]);
    }
    rl_count = 0;
  };
  // box2: squared distance of the cell's box from the query (0 for the own cell), already scaled down by the slack of the collection-time test; kept with the range as
  // the upper 16 bits of a float -- truncated, i.e. rounded DOWN: the re-test at scan time can only keep more than the exact value would -- beside a 16-bit count
  auto visit_range = [&](int pb, int pe, float box2) {
    if constexpr (!FLAT) {
      scan_range(pb, pe);
    } else {
      while (pe > pb) {
        const int cnt = min(pe - pb, 0xffff);
        rl[rl_count * kRangeStride] = make_int2(pb, cnt | (__float_as_int(box2) & (int)0xffff0000));
        rl_count++;
        pb += cnt;
        if (__builtin_amdgcn_ballot_w64(rl_count >= kRangeCap) != 0ull) flush_ranges();  // (some lane's list is full: the lanes that are here scan what they hold)
      }
    }
  };
  const int rlast = (max_shells < rmax - r0) ? r0 + max_shells : rmax;
  for (int r = r0; r <= rlast; r++) {
    int b0[3], b1[3];
    bool any = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int x0 = max(c[a] - r, lo[a]), x1 = min(c[a] + r, hi[a]);
      any = any && x0 <= x1;
      b0[a] = x0 >> 2;
      b1[a] = x1 >> 2;
    }
    // Round 5: NEAR CELLS FIRST for the queries that reach shell 1 with a list that is not full (own cell < k points).  Next to a dense surface those were the launch's
    // longest waves after the far field: with no bound yet they collected all 26 neighbours in walk order and scanned 250-470 candidates per lane, the far corner cells in
    // full before the near face cell had filled the list (profiles/r04_c5_wavelog.txt: 500-545 us per wave against a mean of 107).  They walk the shell TWICE: pass 0
    // takes the cells whose box lies within half a cell edge of the query (the octant it leans to: <= 7 cells), the list is scanned, and pass 1 meets the rest with the
    // k-th distance those brought -- most of it fails the box test below before its range is even looked up.  Queries whose list is full walk once, as before (walking
    // everybody twice: the same lists, 5 % more wave time; one walk with the near ranges sorted to the front of the 16-entry list: no gain, the near cell is often not
    // among the first 16 -- profiles/r05_c5_near_first.txt).  Same candidates, same k smallest; only exact ties in distance could tell the visiting orders apart.
    const int passes = (FLAT && r == 1 && top.count() < top.k) ? 2 : 1;
    const float near2 = 0.25f * h2f;
    if (any)
     for (int pass = 0; pass < passes; pass++) {
      // only blocks that touch the shell are visited: a z-slab of blocks that lies inside the previous cube along z contributes
      // its y-border rows, and such a row its two x-border blocks (surface, not volume, per shell)
      for (int bz = b0[2]; bz <= b1[2]; bz++) {
        const unsigned mz = axis_mask(bz, c[2], r), mz1 = r > 0 ? axis_mask(bz, c[2], r - 1) : 0u;
        const bool zin = mz1 == 0xFu;
        for (int by = b0[1]; by <= b1[1]; by++) {
          const unsigned my = axis_mask(by, c[1], r), my1 = r > 0 ? axis_mask(by, c[1], r - 1) : 0u;
          const bool yin = zin && my1 == 0xFu;
          const int xstep = (yin && b1[0] > b0[0]) ? b1[0] - b0[0] : 1;  // interior row: first and last block only
          for (int bx = b0[0]; bx <= b1[0]; bx += xstep) {
            const unsigned mx = axis_mask(bx, c[0], r), mx1 = r > 0 ? axis_mask(bx, c[0], r - 1) : 0u;
            if (mx1 == 0xFu && my1 == 0xFu && mz1 == 0xFu) continue;  // the whole block lies inside the previous cube
            const size_t bi = ((size_t)(bz - g.geom.lo[2]) * (size_t)g.geom.dim[1] + (size_t)(by - g.geom.lo[1])) * (size_t)g.geom.dim[0] + (size_t)(bx - g.geom.lo[0]);
            const int4 raw = *reinterpret_cast<const int4*>(g.blocks + bi);
            n_blk++;
            const unsigned long long bits = ((unsigned long long)(unsigned)raw.y << 32) | (unsigned long long)(unsigned)raw.x;
            if (bits == 0ull) continue;  // an empty block (most of what a far-field query walks): nothing to mask
            unsigned long long m = bits & cube_mask(mx, my, mz) & ~cube_mask(mx1, my1, mz1);  // occupied cells of this shell
            while (m) {
              const int bit = __ffsll((long long)m) - 1;
              m &= m - 1ull;
              float box2 = 0.0f;
              if (r > 0) {
                // a cell whose box is farther from the query than the current k-th neighbour holds nothing of interest (the corners of a shell's cube
                // usually are): box distance in cell units, f32 with slack -- the test only ever SKIPS, and only cells every point of which fails the
                // list's own strict comparison
                // (relative to the query's own cell: small integers and the query's position inside its cell, exact to 1e-7 whatever the coordinates)
                const float rx = (float)(4 * bx + (bit & 3) - c[0]) - fxf, ry = (float)(4 * by + ((bit >> 2) & 3) - c[1]) - fyf, rz = (float)(4 * bz + (bit >> 4) - c[2]) - fzf;
                const float ex = fmaxf(fmaxf(rx, -rx - 1.0f), 0.0f), ey = fmaxf(fmaxf(ry, -ry - 1.0f), 0.0f), ez = fmaxf(fmaxf(rz, -rz - 1.0f), 0.0f);
                box2 = (ex * ex + ey * ey + ez * ez) * h2f * 0.9999f;
                if (passes == 2 && (box2 <= near2) != (pass == 0)) continue;  // (not this pass's)
                if (box2 > accept) continue;
              }
              const int ord = raw.z + __popcll(bits & ((1ull << bit) - 1ull));
              const int pb = g.cell_start[ord], pe = g.cell_start[ord + 1];
              n_cell++;
              n_f32 += (unsigned)(pe - pb);
              visit_range(pb, pe, box2);
            }
          }
        }
      }
      if constexpr (FLAT) {
        if (pass + 1 < passes) flush_ranges();
      }
     }
    if constexpr (FLAT) flush_ranges();
    const double safe = (double)r * g.h + face;
    const bool done = top.worst() <= safe * safe   // every unvisited point is farther than the current k-th (or than max_sq_dist)
                      || top.count() >= g.n;         // the whole cloud has been seen (clouds smaller than k)
    // (round 5: fewer than k points within kDeferShell cells of the query's cell -- the cells are the wrong tool here, and one lane walking on keeps its wave's other
    // 63 waiting: the caller hands the query to covariance_far_kernel)
    if (sparse && !done && r >= kDeferShell && r < rlast && top.count() < top.k) {
      *sparse = true;
      return false;
    }

    if (done || r == rlast) {
      if (g.counters) {
#ifdef GP_KNN_WAVELOG  // per-wave rows instead of the global counters (which serialise the launch): sum and maximum over the lanes of the candidates, the last shell
        unsigned long long* wl = g.counters + 8 + 8 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        atomicAdd(wl + 3, (unsigned long long)n_f32 | ((unsigned long long)n_f64 << 32));
        atomicMax(wl + 2, (unsigned long long)n_f32 | ((unsigned long long)r << 32));
#else
        atomicAdd(g.counters + 0, 1ull);
        atomicAdd(g.counters + 1, (unsigned long long)n_f32);
        atomicAdd(g.counters + 2, (unsigned long long)n_f64);
        atomicAdd(g.counters + 3, (unsigned long long)n_blk);
        atomicAdd(g.counters + 4, (unsigned long long)n_cell);
#endif
      }
      return done || rlast >= rmax;
    }
  }
  return rlast >= rmax;  // (r0 > rlast: nothing to walk)
}

// First stage of a 1-NN search (GICP correspondences): the 2 x 2 x 2 cells nearest to the query -- its own cell and, per axis, the
// neighbour on the side the query leans to.  Every point within min over the axes of max(f, 1 - f) >= 1/2 cells (f = the query's
// position inside its cell) is in there, and a matched point's neighbour is a few centimetres away, so this settles almost every query with 8 cells instead of the
// 27 of shells 0 + 1.  The 8 block entries are requested together, then the 8 cell ranges, then the points four at a time: three
// dependent round trips in front of the point scan instead of one per block, cell and point.  Returns true when the bound is met;
// otherwise the caller walks the shells with the list as it stands (a point pushed twice cannot displace itself in a 1-NN list).
template <int KMAX, bool FULL>
__device__ __forceinline__ bool knn_query_octant(const BinGridView& g, double qx, double qy, double qz, TopK<KMAX, FULL>& top) {
  static_assert(KMAX == 1, "duplicates are harmless only in a 1-NN list");
  const double ux = qx * g.inv_h, uy = qy * g.inv_h, uz = qz * g.inv_h;
  if (!(fabs(ux) < 1.0e9 && fabs(uy) < 1.0e9 && fabs(uz) < 1.0e9)) return true;  // non-finite query: no neighbours
  const int c[3] = {fast_floor(ux), fast_floor(uy), fast_floor(uz)};
  const double f[3] = {ux - (double)c[0], uy - (double)c[1], uz - (double)c[2]};
  int o[3];
  double reach = 1.0e300;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    o[a] = f[a] < 0.5 ? -1 : 1;
    reach = fmin(reach, fmax(f[a], 1.0 - f[a]));  // distance (cells) to the nearer end of the two-cell span along this axis
  }
  const float qxf = (float)qx, qyf = (float)qy, qzf = (float)qz;
  const float margin = (fabsf(qxf) + fabsf(qyf) + fabsf(qzf) + 1.0f) * 2.4e-7f;  // as in knn_query_bins
  const float m2x4100 = 4100.0f * margin * margin;
  auto loosened = [&](double worst) { return loosened_bound(worst, m2x4100); };
  float accept = loosened(top.worst());
  unsigned n_f32 = 0, n_f64 = 0, n_cell = 0;
  int4 e[8];
  int bit[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int cx = c[0] + ((i & 1) ? o[0] : 0), cy = c[1] + ((i & 2) ? o[1] : 0), cz = c[2] + ((i & 4) ? o[2] : 0);
    const int bx = (cx >> 2) - g.geom.lo[0], by = (cy >> 2) - g.geom.lo[1], bz = (cz >> 2) - g.geom.lo[2];
    bit[i] = (cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4);
    const bool in = bx >= 0 && bx < g.geom.dim[0] && by >= 0 && by < g.geom.dim[1] && bz >= 0 && bz < g.geom.dim[2];
    e[i] = in ? *reinterpret_cast<const int4*>(g.blocks + ((size_t)bz * (size_t)g.geom.dim[1] + (size_t)by) * (size_t)g.geom.dim[0] + (size_t)bx) : make_int4(0, 0, 0, 0);
  }
  int pb[8], pe[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned long long bits = ((unsigned long long)(unsigned)e[i].y << 32) | (unsigned long long)(unsigned)e[i].x;
    const bool occ = (bits >> bit[i]) & 1ull;
    const int ord = e[i].z + __popcll(bits & ((1ull << bit[i]) - 1ull));
    pb[i] = occ ? g.cell_start[ord] : 0;
    pe[i] = occ ? g.cell_start[ord + 1] : 0;
  }
  auto test_point = [&](const float4 v) {
    const float dxf = v.x - qxf, dyf = v.y - qyf, dzf = v.z - qzf;
    if (dxf * dxf + dyf * dyf + dzf * dzf <= accept) {
      n_f64++;
      const double ddx = (double)v.x - qx, ddy = (double)v.y - qy, ddz = (double)v.z - qz;
      top.push(__float_as_int(v.w), ddx * ddx + ddy * ddy + ddz * ddz);
      accept = loosened(top.worst());
    }
  };
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (pe[i] > pb[i]) {
      n_cell++;
      n_f32 += (unsigned)(pe[i] - pb[i]);
      const int last = pe[i] - 1;
      for (int p = pb[i]; p < pe[i]; p += 4) {
        const float4 v0 = g.sorted[p], v1 = g.sorted[min(p + 1, last)], v2 = g.sorted[min(p + 2, last)], v3 = g.sorted[min(p + 3, last)];
        test_point(v0);  // (the clamped repeats of the last point are harmless in a 1-NN list)
        test_point(v1);
        test_point(v2);
        test_point(v3);
      }
    }
  }
  if (g.counters) {
    atomicAdd(g.counters + 5, 1ull);
    atomicAdd(g.counters + 1, (unsigned long long)n_f32);
    atomicAdd(g.counters + 2, (unsigned long long)n_f64);
    atomicAdd(g.counters + 3, 8ull);
    atomicAdd(g.counters + 4, (unsigned long long)n_cell);
  }
  const double safe = reach * g.h;
  return top.worst() <= safe * safe;
}

// The same exact search one and two levels up WITHOUT further sorted copies: the 4 x 4 x 4-cell blocks of the grid are the cells of a
// grid with four times the cell size, and because the points are sorted by (block, cell) a block's points are ONE contiguous range
// of the sorted array -- [cell_start[base], cell_start[base + popcount(bits)]).  Queries whose neighbourhood is too sparse for the
// fine shells (far field of a LiDAR scan) walk cube shells of blocks here, surface only.  SUPER: the cells are 4 x 4 x 4 BLOCKS
// (16 x the cell size) and an entry is the 64-bit occupancy mask of its blocks (BinGridView::super) -- isolated points walk hundreds
// of shells' worth of empty space in a few dozen 8-byte loads this way.  Entries of an x-row are contiguous in memory and are
// requested four at a time: one round trip per (mostly empty) entry was what these walks cost.
// Returns true when the search is complete (bound met, every point seen, or the box exhausted), false after max_shells + 1 shells.
template <int KMAX, bool SUPER, bool FULL>
__device__ __forceinline__ bool knn_query_coarse(const BinGridView& g, double qx, double qy, double qz, TopK<KMAX, FULL>& top, int max_shells) {
  const double unit = (SUPER ? 16.0 : 4.0) * g.h, inv_unit = (SUPER ? 0.0625 : 0.25) * g.inv_h;
  // SUPER coordinates are relative to the grid's first block (the grid origin is not a multiple of four blocks)
  const double ux = qx * inv_unit - (SUPER ? 0.25 * (double)g.geom.lo[0] : 0.0), uy = qy * inv_unit - (SUPER ? 0.25 * (double)g.geom.lo[1] : 0.0),
               uz = qz * inv_unit - (SUPER ? 0.25 * (double)g.geom.lo[2] : 0.0);
  if (!(fabs(ux) < 1.0e9 && fabs(uy) < 1.0e9 && fabs(uz) < 1.0e9)) return true;
  const int c[3] = {fast_floor(ux), fast_floor(uy), fast_floor(uz)};
  const double fx = ux - (double)c[0], fy = uy - (double)c[1], fz = uz - (double)c[2];
  const double face = fmin(fmin(fmin(fx, 1.0 - fx), fmin(fy, 1.0 - fy)), fmin(fz, 1.0 - fz)) * unit;
  int lo[3], hi[3], r0 = 0, rmax = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = SUPER ? 0 : g.geom.lo[a];
    hi[a] = SUPER ? g.sdim[a] - 1 : g.geom.lo[a] + g.geom.dim[a] - 1;
    r0 = max(r0, max(lo[a] - c[a], c[a] - hi[a]));
    rmax = max(rmax, max(abs(c[a] - lo[a]), abs(c[a] - hi[a])));
  }
  const int dimx = SUPER ? g.sdim[0] : g.geom.dim[0], dimy = SUPER ? g.sdim[1] : g.geom.dim[1];
  const float qxf = (float)qx, qyf = (float)qy, qzf = (float)qz;
  const float margin = (fabsf(qxf) + fabsf(qyf) + fabsf(qzf) + 1.0f) * 2.4e-7f;  // as in knn_query_bins
  const float m2x4100 = 4100.0f * margin * margin;
  auto loosened = [&](double worst) { return loosened_bound(worst, m2x4100); };
  float accept = loosened(top.worst());
  unsigned n_f32 = 0, n_f64 = 0, n_blk = 0;
  auto test_point = [&](const float4 v) {
    const float dxf = v.x - qxf, dyf = v.y - qyf, dzf = v.z - qzf;
    if (dxf * dxf + dyf * dyf + dzf * dzf <= accept) {
      n_f64++;
      const double ddx = (double)v.x - qx, ddy = (double)v.y - qy, ddz = (double)v.z - qz;
      top.push(__float_as_int(v.w), ddx * ddx + ddy * ddy + ddz * ddz);
      accept = loosened(top.worst());
    }
  };
  auto scan_block = [&](const int4 raw) {
    const unsigned long long bits = ((unsigned long long)(unsigned)raw.y << 32) | (unsigned long long)(unsigned)raw.x;
    if (bits == 0ull) return;
    const int pb = g.cell_start[raw.z], pe = g.cell_start[raw.z + __popcll(bits)];
    n_f32 += (unsigned)(pe - pb);
    int p = pb;
    for (; p + 4 <= pe; p += 4) {  // four loads in flight (a block holds a few hundred points at most)
      const float4 v0 = g.sorted[p], v1 = g.sorted[p + 1], v2 = g.sorted[p + 2], v3 = g.sorted[p + 3];
      test_point(v0);
      test_point(v1);
      test_point(v2);
      test_point(v3);
    }
    for (; p < pe; p++) test_point(g.sorted[p]);
  };
  auto scan_super = [&](unsigned long long m, int sx, int sy, int sz) {  // occupied blocks of superblock (sx, sy, sz)
    while (m) {
      const int bit = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      const int bx = 4 * sx + (bit & 3), by = 4 * sy + ((bit >> 2) & 3), bz = 4 * sz + (bit >> 4);
      // a block farther away than the current k-th neighbour holds nothing of interest
      const double e = 4.0 * g.h;
      const double x0 = (double)(g.geom.lo[0] + bx) * e, y0 = (double)(g.geom.lo[1] + by) * e, z0 = (double)(g.geom.lo[2] + bz) * e;
      const double ddx = fmax(fmax(x0 - qx, qx - (x0 + e)), 0.0), ddy = fmax(fmax(y0 - qy, qy - (y0 + e)), 0.0), ddz = fmax(fmax(z0 - qz, qz - (z0 + e)), 0.0);
      if (ddx * ddx + ddy * ddy + ddz * ddz > top.worst()) continue;
      n_blk++;
      scan_block(*reinterpret_cast<const int4*>(g.blocks + ((size_t)bz * (size_t)g.geom.dim[1] + (size_t)by) * (size_t)g.geom.dim[0] + (size_t)bx));
    }
  };
  // entries xa .. xb (step `step`) of one x-row
  auto visit_row = [&](int xa, int xb, int step, int y, int z) {
    const size_t row0 = ((size_t)(z - lo[2]) * (size_t)dimy + (size_t)(y - lo[1])) * (size_t)dimx;
    for (int x = xa; x <= xb; x += 4 * step) {
      if constexpr (SUPER) {
        unsigned long long e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) e[i] = g.super[row0 + (size_t)(min(x + i * step, xb) - lo[0])];
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (x + i * step <= xb) scan_super(e[i], x + i * step, y, z);
      } else {
        int4 e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) e[i] = *reinterpret_cast<const int4*>(g.blocks + row0 + (size_t)(min(x + i * step, xb) - lo[0]));
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (x + i * step <= xb) {
            n_blk++;
            scan_block(e[i]);
          }
      }
    }
  };
  bool done = false;
  const int rlast = (max_shells < rmax - r0) ? r0 + max_shells : rmax;
  for (int r = r0; r <= rlast && !done; r++) {
    const int z0 = max(c[2] - r, lo[2]), z1 = min(c[2] + r, hi[2]);
    const int y0 = max(c[1] - r, lo[1]), y1 = min(c[1] + r, hi[1]);
    const int x0 = max(c[0] - r, lo[0]), x1 = min(c[0] + r, hi[0]);
    if (z0 <= z1 && y0 <= y1 && x0 <= x1) {
      for (int z = z0; z <= z1; z++) {
        const bool zface = z == c[2] - r || z == c[2] + r;
        for (int y = y0; y <= y1; y++) {
          if (zface || y == c[1] - r || y == c[1] + r || r == 0) {
            visit_row(x0, x1, 1, y, z);
          } else {  // interior row of the cube: only its two end entries belong to the shell
            const int xa = c[0] - r >= lo[0] ? c[0] - r : c[0] + r, xb = c[0] + r <= hi[0] ? c[0] + r : c[0] - r;
            if (xa >= lo[0] && xa <= hi[0] && xb >= xa) visit_row(xa, xb, xb > xa ? xb - xa : 1, y, z);
          }
        }
      }
    }
    const double safe = (double)r * unit + face;
    done = top.worst() <= safe * safe || top.count() >= g.n;
  }
  if (g.counters) {
    atomicAdd(g.counters + 0, 1ull);
    atomicAdd(g.counters + 1, (unsigned long long)n_f32);
    atomicAdd(g.counters + 2, (unsigned long long)n_f64);
    atomicAdd(g.counters + 3, (unsigned long long)n_blk);
  }
  return done || rlast >= rmax;
}

// what a search runs on: the binned structure, or -- for clouds whose bounding box is too large for it -- the hashed multi-level grid
// LiDAR density spans three orders of magnitude between the near and the far field: a query first tries the shells 0 and 1 of the
// cells (<= 8 block entries); when that does not settle it (sparse neighbourhood) it starts over on the blocks taken as cells four
// times the size, and then on the superblocks (knn_query_coarse), which it walks until the bound is met.  (More binned levels, cell size x4 each, can be stacked in
// between -- gp_debug_set_knn_structure -- but building them costs more than they save.)  Every stage is an exact search, so the
// staging affects speed only.
struct SearchView {
  int binned;      // number of binned levels (0: hashed fallback)
  int fine_shells; // shells beyond the first one that a query walks on the finest cells before it starts over on a coarser level (4 in rounds 2-3)
  int block_stage; // shells of BLOCKS (cells four times the size) a query walks between the fine shells and the superblocks; 0 = round 3's staging (none)
  BinGridView bins[kMaxLevels];
  MultiGridView hashed;
};

// stage 0: cell shells 0 .. 4 (occupied cells only: work-efficient while the neighbourhood is a few cells wide); stage 1: superblock
// shells -- blocks as cells, those beyond the current k-th distance skipped -- until the bound is met or the box is exhausted
template <int KMAX, bool FULL = false, bool FLAT = false>
__device__ __forceinline__ void knn_query_any(const SearchView& g, double qx, double qy, double qz, int want, TopK<KMAX, FULL>& top, bool skip_fine = false, int2* rl = nullptr,
                                              bool* sparse = nullptr) {
  if (g.binned) {
    const int k = top.k;
    const double bound = top.worst();  // the caller's max_sq_dist (nothing has been pushed yet)
#ifdef GP_KNN_WAVELOG  // measurement build (scripts/r04_c5_wavelog.py): per-wave stamps of the stages, rows of 8 uint64 behind the 8 work counters
    unsigned long long* wl = g.bins[0].counters ? g.bins[0].counters + 8 + 8 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) : nullptr;
    const bool wlane = (threadIdx.x & 63) == 0;
#define GP_WL(slot, value) do { if (wl && wlane) wl[slot] = (value); } while (0)
#define GP_WL_ALL(slot, value) do { const unsigned long long v_ = (value); if (wl && wlane) wl[slot] = v_; } while (0)
#else
#define GP_WL(slot, value) do { } while (0)
#define GP_WL_ALL(slot, value) do { } while (0)
#endif
    GP_WL(0, __builtin_amdgcn_s_memrealtime());
    // (skip_fine: the row-tiled pass has scanned the shells 0 and 1 of the finest level, which therefore cannot settle the query; the
    // walk still starts there -- the list is not carried over -- but goes on to shell 4 at once)
    if constexpr (KMAX == 1) {
      if (knn_query_octant<KMAX, FULL>(g.bins[0], qx, qy, qz, top)) return;
    }
    bool settled = false;
    for (int l = 0; l < g.binned && !settled; l++) {
      if (l > 0) top.init(k, bound);
      settled = knn_query_bins<KMAX, FULL, FLAT>(g.bins[l], qx, qy, qz, top, (l + 1 < g.binned && !skip_fine) ? 1 : g.fine_shells, rl, (l + 1 == g.binned) ? sparse : nullptr);
      if (sparse && *sparse) return;
    }
    GP_WL_ALL(5, (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(!settled)));
    GP_WL(1, __builtin_amdgcn_s_memrealtime());
    if (settled) return;
    if (sparse) {  // round 5: what the fine shells do not settle is searched by a whole wave (covariance_far_kernel), not by this lane with 63 others waiting
      *sparse = true;
      return;
    }
    // round 4: sparse neighbourhoods (the far field of a LiDAR scan: one point per cell) first try the BLOCKS as cells -- shells 0 .. block_stage of a grid four
    // times as coarse, 27 entries for the first two, a few points each -- before they start over on the superblocks, whose first shell alone scans every point
    // within 4-12 m of the query: those queries were the launch's tail (a hundred 64-query chunks of 350-460 us in a launch whose balanced length was 334 us)
    if (g.block_stage > 0) {
      top.init(k, bound);
      settled = knn_query_coarse<KMAX, false, FULL>(g.bins[g.binned - 1], qx, qy, qz, top, g.block_stage);
    }
    if (settled) return;
    top.init(k, bound);
    knn_query_coarse<KMAX, true, FULL>(g.bins[g.binned - 1], qx, qy, qz, top, 0x3fffffff);
  } else {
    knn_query_multi<KMAX, FULL>(g.hashed, qx, qy, qz, want, top);
  }
}

// non-finite points have no neighbours: identity covariance, counted as "short" (covariance_estimation.cpp:27-31)
__global__ void __launch_bounds__(256) nonfinite_identity_kernel(const float* __restrict__ points, int n, float* __restrict__ covs, int* __restrict__ num_short) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = points[3 * (size_t)i], y = points[3 * (size_t)i + 1], z = points[3 * (size_t)i + 2];
  if (fabsf(x) < 3.0e38f && fabsf(y) < 3.0e38f && fabsf(z) < 3.0e38f) return;
  for (int j = 0; j < 9; j++) covs[9 * (size_t)i + j] = (j % 4 == 0) ? 1.0f : 0.0f;
  atomicAdd(num_short, 1);
}

// superblock occupancy: bit (bx & 3) + 4 (by & 3) + 16 (bz & 3) of entry (bx >> 2, by >> 2, bz >> 2), relative block coordinates
__global__ void super_mark_kernel(const int* __restrict__ occ_blocks, int num, GridGeom geom, int sdim0, int sdim1, unsigned long long* __restrict__ super,
                                  const FillJob caller_zero) {
  run_fill_job(caller_zero);  // (words the CALLER's kernels behind this one want zeroed: gp_estimate_covariances' counters)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  const long long b = occ_blocks[i];
  const int bx = (int)(b % geom.dim[0]), by = (int)((b / geom.dim[0]) % geom.dim[1]), bz = (int)(b / ((long long)geom.dim[0] * geom.dim[1]));
  atomicOr(super + ((size_t)(bz >> 2) * sdim1 + (by >> 2)) * sdim0 + (bx >> 2), 1ull << ((bx & 3) | ((by & 3) << 2) | ((bz & 3) << 4)));
}

__global__ void __launch_bounds__(256) gather_sorted_kernel(const float* __restrict__ points, const int* __restrict__ order, int n, float4* __restrict__ sorted,
                                                           const FillJob zero_super) {
  run_fill_job(zero_super);  // (the superblock masks the kernel behind this one ORs into: gp_host.hpp, FillJob)
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const size_t i = (size_t)order[j];
  sorted[j] = make_float4(points[3 * i], points[3 * i + 1], points[3 * i + 2], __int_as_float((int)i));
}

template <int KMAX>
__global__ void __launch_bounds__(128) knn_kernel(SearchView g, const float* __restrict__ queries, int nq, int k, double max_sq_dist, int* __restrict__ indices,
                                                  double* __restrict__ sq_dists, int* __restrict__ num_found) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= nq) return;
  TopK<KMAX> top;
  top.init(k, max_sq_dist);
  knn_query_any<KMAX>(g, (double)queries[3 * (size_t)i], (double)queries[3 * (size_t)i + 1], (double)queries[3 * (size_t)i + 2], 2 * k, top);
#pragma unroll
  for (int j = 0; j < KMAX; j++)
    if (j < k) {
      indices[(size_t)i * k + j] = j < top.found ? top.idx[j] : -1;
      if (sq_dists) sq_dists[(size_t)i * k + j] = top.d[j];
    }
  if (num_found) num_found[i] = top.found;
}

// ---- Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3d>::computeDirect, restated from the published closed-form algorithm -----
__device__ __forceinline__ void eig3_roots(const double* m /*col-major sym*/, double* roots) {
  const double s_inv3 = 1.0 / 3.0, s_sqrt3 = 1.7320508075688772;
  const double c0 = m[0] * m[4] * m[8] + 2.0 * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  const double c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const double c2 = m[0] + m[4] + m[8];
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = a_over_3 < 0.0 ? 0.0 : a_over_3;
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = q < 0.0 ? 0.0 : q;
  const double rho = sqrt(a_over_3);
  const double theta = atan2(sqrt(q), half_b) * s_inv3;
  const double cos_theta = cos(theta), sin_theta = sin(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 + 2.0 * rho * cos_theta;
}

__device__ __forceinline__ void eig3_extract_kernel(const double* mat, double* res, double* representative) {
  int i0 = 0;
  double best = fabs(mat[0]);
  if (fabs(mat[4]) > best) {
    best = fabs(mat[4]);
    i0 = 1;
  }
  if (fabs(mat[8]) > best) i0 = 2;
  for (int r = 0; r < 3; r++) representative[r] = mat[i0 * 3 + r];
  const int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
  const double* a = representative;
  const double* b1 = mat + 3 * i1;
  const double* b2 = mat + 3 * i2;
  const double c0[3] = {a[1] * b1[2] - a[2] * b1[1], a[2] * b1[0] - a[0] * b1[2], a[0] * b1[1] - a[1] * b1[0]};
  const double c1[3] = {a[1] * b2[2] - a[2] * b2[1], a[2] * b2[0] - a[0] * b2[2], a[0] * b2[1] - a[1] * b2[0]};
  const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2], n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  if (n0 > n1) {
    const double s = 1.0 / sqrt(n0);
    for (int r = 0; r < 3; r++) res[r] = c0[r] * s;
  } else {
    const double s = 1.0 / sqrt(n1);
    for (int r = 0; r < 3; r++) res[r] = c1[r] * s;
  }
}

__device__ __forceinline__ void eig3_direct(const double* mat /*col-major, lower triangle referenced*/, double* evals, double* evecs) {
  const double eps = 2.220446049250313e-16;
  const double shift = (mat[0] + mat[4] + mat[8]) / 3.0;
  double scaled[9] = {mat[0] - shift, mat[1], mat[2], mat[1], mat[4] - shift, mat[5], mat[2], mat[5], mat[8] - shift};
  double scale = 0.0;
  for (int i = 0; i < 9; i++) scale = fmax(scale, fabs(scaled[i]));
  if (scale > 0.0)
    for (int i = 0; i < 9; i++) scaled[i] /= scale;
  eig3_roots(scaled, evals);
  if ((evals[2] - evals[0]) <= eps) {
    for (int i = 0; i < 9; i++) evecs[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    double tmp[9];
    for (int i = 0; i < 9; i++) tmp[i] = scaled[i];
    double d0 = evals[2] - evals[1];
    const double d1 = evals[1] - evals[0];
    int k = 0, l = 2;
    if (d0 > d1) {
      k = 2;
      l = 0;
      d0 = d1;
    }
    tmp[0] -= evals[k];
    tmp[4] -= evals[k];
    tmp[8] -= evals[k];
    eig3_extract_kernel(tmp, evecs + 3 * k, evecs + 3 * l);
    if (d0 <= 2.0 * eps * d1) {
      double* ck = evecs + 3 * k;
      double* cl = evecs + 3 * l;
      const double dot = ck[0] * cl[0] + ck[1] * cl[1] + ck[2] * cl[2];
      for (int r = 0; r < 3; r++) cl[r] -= dot * cl[r];
      const double nn = sqrt(cl[0] * cl[0] + cl[1] * cl[1] + cl[2] * cl[2]);
      for (int r = 0; r < 3; r++) cl[r] /= nn;
    } else {
      double dummy[3];
      for (int i = 0; i < 9; i++) tmp[i] = scaled[i];
      tmp[0] -= evals[l];
      tmp[4] -= evals[l];
      tmp[8] -= evals[l];
      eig3_extract_kernel(tmp, evecs + 3 * l, dummy);
    }
    const double* c2 = evecs + 6;
    const double* c0 = evecs;
    const double c1[3] = {c2[1] * c0[2] - c2[2] * c0[1], c2[2] * c0[0] - c2[0] * c0[2], c2[0] * c0[1] - c2[1] * c0[0]};
    const double nn = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    for (int r = 0; r < 3; r++) evecs[3 + r] = c1[r] / nn;
  }
  for (int i = 0; i < 3; i++) evals[i] = evals[i] * scale + shift;
}

__device__ __forceinline__ void inverse3_general(const double* a /*col-major*/, double* inv) {
  auto A = [&](int r, int c) { return a[c * 3 + r]; };
  const double c00 = A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1), c10 = A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2), c20 = A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0);
  const double invdet = 1.0 / (A(0, 0) * c00 + A(0, 1) * c10 + A(0, 2) * c20);
  inv[0] = c00 * invdet;
  inv[1] = c10 * invdet;
  inv[2] = c20 * invdet;
  inv[3] = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * invdet;
  inv[4] = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * invdet;
  inv[5] = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * invdet;
  inv[6] = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * invdet;
  inv[7] = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * invdet;
  inv[8] = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * invdet;
}

// estimate_covariances (features/covariance_estimation.cpp:18-77): k-NN (query included) -> sample covariance ->
// V diag(1e-3, 1, 1) V^-1.  Fewer than k neighbours -> identity (:27-31).
// sample covariance of the k neighbours -> V diag(1e-3, 1, 1) V^-1 (features/covariance_estimation.cpp:33-53)
template <int KMAX, bool FULL>
__device__ __forceinline__ void covariance_from_neighbours(const TopK<KMAX, FULL>& top, const float* __restrict__ points, int k, float* __restrict__ out) {
  double sp[3] = {0, 0, 0}, spp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < KMAX; j++)
    if (j < k) {
      const size_t nb = (size_t)top.idx[j];
      const double p[3] = {(double)points[3 * nb], (double)points[3 * nb + 1], (double)points[3 * nb + 2]};
      for (int r = 0; r < 3; r++) sp[r] += p[r];
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) spp[c * 3 + r] += p[r] * p[c];
    }
  double cov[9];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) cov[c * 3 + r] = (spp[c * 3 + r] - (sp[r] / (double)k) * sp[c]) / (double)k;  // :43
  double evals[3], V[9], Vinv[9];
  eig3_direct(cov, evals, V);
  inverse3_general(V, Vinv);
  const double lam[3] = {1e-3, 1.0, 1.0};
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) {
      double s = 0.0;
      for (int kk = 0; kk < 3; kk++) s += V[kk * 3 + r] * lam[kk] * Vinv[c * 3 + kk];
      out[c * 3 + r] = (float)s;
    }
}

// Heavy queries first (round 4).  A query whose own cell holds fewer than k points cannot settle in shell 0: it walks shell 1 at least -- 26 more cells, dense ones
// scanned in full until its list is full, or shell after shell of empty space in the far field -- and a wave of such queries runs 300-500 us against a mean of ~90
// (profiles/r04_c5_wavelog.txt).  In cell-sorted order those waves are scattered over the launch, and the ones that start late ARE its tail (99 % of the waves done
// at 420 us, the last at 710).  The launch therefore takes its queries through an order array: the positions of the queries with own-cell population < k first
// (ascending, so that neighbours in the list are still neighbours in space), all others behind them -- longest-processing-time-first with a per-query predictor,
// and waves whose lanes have alike work.  Same queries, same per-query search: identical results.
// (the predictor depends on the query's CELL only, so the prefix sums run over the cells -- a few 10^5 entries -- not over the points: per cell the number of its
// points when that is below k, else 0; the points of a heavy cell c go to heavy_before[c] + their rank inside the cell, the others behind all heavy ones in order)
struct HeavyCellCount {
  const int* cell_start;
  int k;
  __device__ __forceinline__ int operator()(long long c) const {
    const int size = cell_start[c + 1] - cell_start[c];
    return size < k ? size : 0;
  }
};
__global__ void __launch_bounds__(256) heavy_first_order_kernel(const int* __restrict__ cell_start, const unsigned* __restrict__ cell_of, const int* __restrict__ heavy_before,
                                                                const int* __restrict__ num_heavy, int k, int nq, int* __restrict__ order) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nq) return;
  if (t == 0) order[nq] = nq;  // (the list's length, where covariance_kernel's todo protocol reads it)
  const int c = (int)cell_of[t];
  const int b = cell_start[c], size = cell_start[c + 1] - b, hb = heavy_before[c];
  order[size < k ? hb + (t - b) : *num_heavy + (t - hb)] = t;
}

// estimate_covariances, per-lane search (every query walks its own shells; see knn_query_bins / knn_query): the general path, and
// the second pass of the tiled kernel below for the queries it left over (todo_list != nullptr: the *todo_count positions listed)
// MIN_WAVES = 4 (k <= 10): registers capped at 128 for four waves per SIMD instead of three: 1.41 -> 1.28 ms per 1 M points (round 2).
// FULL: k == KMAX, the list is always full: straight-line insertion (TopK<KMAX, true>)
__device__ __forceinline__ void todo_append(bool flag, int pos, int* __restrict__ todo_list, int* __restrict__ todo_count) {
  const unsigned long long m = __ballot(flag);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(todo_count, __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1, 64);
  if (flag) todo_list[base + __popcll(m & ((1ull << lane) - 1ull))] = pos;
}

// ---- sparse neighbourhoods: SIXTEEN LANES PER QUERY (round 5) ----------------------------------------------------------------------
// A query the fine shells do not settle (the far field of a LiDAR scan: ring spacing of a metre, one point per 0.25 m cell) used to go on lane by lane: blocks as cells,
// then superblocks -- a chain of dependent loads on ONE lane while 63 wait: 400-600 us per wave against a mean of 110 (profiles/r05_c5_wavelog.txt), the tail of the
// launch.  Here kFarLanes lanes share one query (four queries per wave).  Such queries are few (0.1-1.5 % of a cloud), so the kernel runs at a fraction of a wave per
// SIMD and nothing hides a load's latency: what counts is the NUMBER OF DEPENDENT ROUND TRIPS, and every stage asks for everything it needs at once.
//   blocks -> candidates (far_process): a list of <= 128 blocks (4 x 4 x 4 cells: 1 m for the covariance structure), eight per lane: the eight block entries in one
//            trip, the sixteen cell_start words in the next, the (first point, length) ranges into LDS; then the CANDIDATES -- not the blocks -- are dealt to the lanes
//            (candidate o of the concatenated ranges to lane o % 16: one dense block does not leave fifteen lanes idle), eight loads in flight per lane;
//   phase A  the 5 x 5 x 5 blocks around the query's block: one list;
//   phase B  cube shells of SUPERBLOCKS (4 x 4 x 4 blocks, one 64-bit occupancy mask each) around the query's superblock, the masks dealt to the lanes four at a time:
//            empty space costs one 8-byte load per 64 m^3; occupied blocks outside phase A's cube and not farther than the group's best k-th distance so far are
//            appended to the list (LDS counter), which is processed whenever it is full and at the end of the shell.
// Every lane keeps the k best of what it scanned (exact f64 distances, strict '<' like KnnResult::push).  After phase A / a shell the group is done when k of its
// candidates lie within the safe radius (R units + the distance to the nearest face of the query's own unit: every unvisited point is farther) -- the stopping rule of the
// per-lane search, counted with ballots instead of read off a merged list -- or when the grid is exhausted.  Then the group merges its lists once: k rounds of "smallest
// head" (a group-wide minimum each; ties: smaller original index), which yields the neighbours in ascending order, the order covariance_from_neighbours sums them in.
// Exact like the per-lane search: the neighbour set is the k smallest distances either way.
constexpr int kFarBlockShells = 2, kFarLanes = 16, kFarList = 128, kFarPer = kFarList / kFarLanes, kFarMasks = 4;
struct FarGroupLds {
  int2 range[kFarList];  // (first point, points) per listed block
  int blk[kFarList];     // linear block index (phase B)
  int count;
  int pad_[3];
};
template <int KMAX>
// (four waves per SIMD = 128 registers, 560 B of scratch per lane: uncapped -- 256 registers, no scratch -- the kernel is 20 % faster ALONE (190 vs 240 us), but beside the
// other launch, whose waves hold a quarter of a SIMD's registers each, a 256-register wave waits until two of them on one SIMD have retired: profiles/r05_c5_summary.txt)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) covariance_far_kernel(BinGridView g, const float* __restrict__ points, int k, float* __restrict__ covs, int* __restrict__ num_short,
                                                             const int* __restrict__ far_list, const float* __restrict__ far_bound, const int* __restrict__ far_count) {
  constexpr int kGroups = 64 / kFarLanes;
  // these few waves are chains of dependent round trips with short bursts of arithmetic in between, and they run beside the other launch's waves (four per SIMD, busy
  // with list insertions): at equal issue priority every burst takes four times as long -- the kernel measured 350 us beside the light queries' launch, 190 alone
  __builtin_amdgcn_s_setprio(3);
  __shared__ FarGroupLds lds_all[kGroups];  // (one wave per workgroup: a 256-register wave finds a place where a four-wave workgroup waits for four at once)
  const int lane = threadIdx.x & 63, sub = lane % kFarLanes, grp = lane / kFarLanes;
  FarGroupLds& L = lds_all[grp];
  const unsigned long long gmask = ((1ull << kFarLanes) - 1ull) << (grp * kFarLanes);
  const int groups = (int)gridDim.x * kGroups, count = *far_count;
  constexpr double kInf = 1.7976931348623157e308;
  auto wave_sync = [] {  // LDS traffic of a wave is ordered; this keeps the compiler from moving LDS accesses across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int w0 = (int)blockIdx.x * kGroups; w0 < count; w0 += groups) {  // (wave-uniform trip count; a group without a query idles through it)
    const int w = w0 + grp;
    const bool active = w < count;
    const int t = far_list[active ? w : w0];
    // what the per-lane search knew when it gave up: its k-th distance (rounded up; +inf when it had found fewer than k) -- nothing farther can be a neighbour, so
    // blocks beyond it are not even listed (a noise point a metre above a dense surface: without it phase A scanned 6000-8000 candidates, profiles/r05_c5_farlog.txt)
    const double known = (double)far_bound[active ? w : w0];
    double bound = known;  // nothing farther than this can be a neighbour: what the per-lane search knew, then the group's exact k-th distance after every stage
    const float4 self = g.sorted[t];
    const int i = __float_as_int(self.w);
    const double qx = (double)self.x, qy = (double)self.y, qz = (double)self.z;
    const double B = 4.0 * g.h, inv_B = 0.25 * g.inv_h;
    // block coordinates RELATIVE to the grid's first block (what the superblock masks are indexed by); the query is a point of the cloud: inside the grid
    const double ux = qx * inv_B - (double)g.geom.lo[0], uy = qy * inv_B - (double)g.geom.lo[1], uz = qz * inv_B - (double)g.geom.lo[2];
    const int c[3] = {fast_floor(ux), fast_floor(uy), fast_floor(uz)};
    const double fx = ux - (double)c[0], fy = uy - (double)c[1], fz = uz - (double)c[2];
    const double face = fmin(fmin(fmin(fx, 1.0 - fx), fmin(fy, 1.0 - fy)), fmin(fz, 1.0 - fz)) * B;
    const int dim[3] = {g.geom.dim[0], g.geom.dim[1], g.geom.dim[2]};
    const double ox = (double)g.geom.lo[0] * B, oy = (double)g.geom.lo[1] * B, oz = (double)g.geom.lo[2] * B;  // corner of block (0, 0, 0)
    TopK<KMAX, false> top;
    top.init(k, kInf);
#ifdef GP_KNN_WAVELOG  // rows of 8 words per far query behind the per-wave rows: start | phase A done | phase B done | merged | candidates of the group | shells of B | settled in A
    unsigned long long* fl = (g.counters && active) ? g.counters + 8 + 8 * ((size_t)(g.n + 63) / 64 + 2) + 8 * (size_t)w : nullptr;
    if (fl && sub == 0) fl[0] = __builtin_amdgcn_s_memrealtime();
    unsigned far_cands = 0, far_bshells = 0;
#endif
    // the blocks listed for this group (nb <= kFarList; block index of list position j from `block_of(j)`) -> their points through the lanes' lists.
    // Wave-convergent: every lane of the wave calls it, groups without work pass nb = 0
    int fresh = 0;  // candidates the group has scanned since its lists were last merged (group-uniform)
    auto far_process = [&](int nb, auto block_of, auto block_wanted) {
      if (__builtin_amdgcn_ballot_w64(nb > 0) == 0ull) return;  // (an empty shell -- an outlier walks several: no round trips for nothing)
      int4 raw[kFarPer];
      bool wanted[kFarPer];
#pragma unroll
      for (int q = 0; q < kFarPer; q++) {
        const int j = sub * kFarPer + q;
        wanted[q] = j < nb && block_wanted(j);
        raw[q] = *reinterpret_cast<const int4*>(g.blocks + (wanted[q] ? block_of(j) : 0));  // (unconditional: the eight loads are in flight together)
      }
      int pb[kFarPer], pe[kFarPer];
#pragma unroll
      for (int q = 0; q < kFarPer; q++) {
        const unsigned long long bits = ((unsigned long long)(unsigned)raw[q].y << 32) | (unsigned long long)(unsigned)raw[q].x;
        const bool valid = wanted[q] && bits != 0ull;
        pb[q] = g.cell_start[valid ? raw[q].z : 0];
        pe[q] = valid ? g.cell_start[raw[q].z + __popcll(bits)] : pb[q];
      }
      int mine = 0;
#pragma unroll
      for (int q = 0; q < kFarPer; q++) {
        const int len = wanted[q] ? pe[q] - pb[q] : 0;
        L.range[sub * kFarPer + q] = make_int2(pb[q], len);
        mine += len;
      }
      int total = mine;
#pragma unroll
      for (int off = kFarLanes / 2; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
      wave_sync();
      fresh += total;
#ifdef GP_KNN_WAVELOG
      far_cands += (unsigned)total;
#endif
      // candidate o of the concatenated ranges -> lane o % kFarLanes; a lane's ordinals ascend, so its cursor over the ranges only moves forward
      int rj = 0, rc = 0;  // range under the cursor, candidates in front of it
      int2 cur = L.range[0];
      constexpr int kFarCand = 4;  // candidates in flight per lane
      for (int o = sub; __builtin_amdgcn_ballot_w64(o < total) != 0ull; o += kFarLanes * kFarCand) {
        int pp[kFarCand];
        bool ok[kFarCand];
#pragma unroll
        for (int q = 0; q < kFarCand; q++) {
          const int oq = o + q * kFarLanes;
          ok[q] = oq < total;
          if (ok[q]) {
            while (oq >= rc + cur.y) {
              rc += cur.y;
              rj++;
              cur = L.range[rj];
            }
            pp[q] = cur.x + (oq - rc);
          } else {
            pp[q] = 0;
          }
        }
        float4 v[kFarCand];
#pragma unroll
        for (int q = 0; q < kFarCand; q++) v[q] = g.sorted[pp[q]];
#pragma unroll
        for (int q = 0; q < kFarCand; q++)
          if (ok[q]) {
            const double ex = (double)v[q].x - qx, ey = (double)v[q].y - qy, ez = (double)v[q].z - qz;
            const double d2 = ex * ex + ey * ey + ez * ez;
            if (d2 <= bound) top.push(__float_as_int(v[q].w), d2);
          }
      }
      // a lane that holds k candidates bounds the group's k-th distance with its own (its list is a subset of the group's): `bound` tightens after EVERY list, without a
      // merge, and prunes the rest of the shell (the kitti scan's far kernel 230 -> 192 us; profiles/r05_c5_summary.txt item 6)
      double lane_kth = top.worst();
#pragma unroll
      for (int off = kFarLanes / 2; off > 0; off >>= 1) lane_kth = fmin(lane_kth, __shfl_xor(lane_kth, off, 64));
      bound = fmin(bound, lane_kth);
      wave_sync();  // (the list may be refilled)
    };
    // k candidates of the group within `safe`?  (a lane holds its k best: one that has k within the radius settles it alone)
    auto settled_within = [&](double safe, bool live) {
      const double safe2 = safe * safe;
      int within = 0;
#pragma unroll
      for (int j = 0; j < KMAX; j++) within += __popcll(__builtin_amdgcn_ballot_w64(live && j < k && top.idx[j] >= 0 && top.d[j] <= safe2) & gmask);
      return within >= k;
    };
    // the k smallest of the group's lists, ascending (ties: smaller index): k rounds of "smallest head", a group-wide minimum each -> fin.idx (group-uniform), the
    // group's exact k-th distance in `kth` (+inf while it holds fewer than k); returns how many there are.  ~4 us: once per shell, not per block
    int fin_i[KMAX];
    double kth = kInf;
    auto merge = [&]() -> int {
#pragma unroll
      for (int r = 0; r < KMAX; r++) fin_i[r] = -1;
      kth = kInf;
      int head = 0, n_have = 0;
#pragma unroll
      for (int r = 0; r < KMAX; r++) {
        double cur = kInf;
        int cur_i = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < KMAX; j++)
          if (head == j && j < k && top.idx[j] >= 0) cur = top.d[j], cur_i = top.idx[j];
        double m = cur;
#pragma unroll
        for (int off = kFarLanes / 2; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
        int mi = cur == m ? cur_i : 0x7fffffff;
#pragma unroll
        for (int off = kFarLanes / 2; off > 0; off >>= 1) mi = min(mi, __shfl_xor(mi, off, 64));
        if (r < k && mi != 0x7fffffff) {
          fin_i[r] = mi;
          n_have = r + 1;
          if (r == k - 1) kth = m;
          if (cur == m && cur_i == mi) head++;  // (every candidate was scanned by exactly one lane: one winner)
        }
      }
      return n_have;
    };
    int have = 0;
    bool done = !active;
    auto box2 = [&](int bx, int by, int bz) {  // squared distance of a block's box from the query
      const double bx0 = ox + (double)bx * B, by0 = oy + (double)by * B, bz0 = oz + (double)bz * B;
      const double ddx = fmax(fmax(bx0 - qx, qx - (bx0 + B)), 0.0), ddy = fmax(fmax(by0 - qy, qy - (by0 + B)), 0.0), ddz = fmax(fmax(bz0 - qz, qz - (bz0 + B)), 0.0);
      return ddx * ddx + ddy * ddy + ddz * ddz;
    };
    // ---- phase A: the blocks around the query's block: the 3 x 3 x 3 cube, then -- with the bound that brought -- the shell around it.  One list each ----
    {
      int ra_max = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) ra_max = max(ra_max, max(c[a], dim[a] - 1 - c[a]));  // the shell that covers the grid
      static_assert((2 * kFarBlockShells + 1) * (2 * kFarBlockShells + 1) * (2 * kFarBlockShells + 1) <= kFarList, "a cube of phase A is one list");
      for (int R = 1; R <= kFarBlockShells; R++) {
        const int x0 = max(c[0] - R, 0), x1 = min(c[0] + R, dim[0] - 1), y0 = max(c[1] - R, 0), y1 = min(c[1] + R, dim[1] - 1), z0 = max(c[2] - R, 0), z1 = min(c[2] + R, dim[2] - 1);
        const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
        far_process(
          done ? 0 : nx * ny * nz, [&](int j) { return ((size_t)(z0 + j / (nx * ny)) * (size_t)dim[1] + (size_t)(y0 + (j / nx) % ny)) * (size_t)dim[0] + (size_t)(x0 + j % nx); },
          [&](int j) {
            const int bx = x0 + j % nx, by = y0 + (j / nx) % ny, bz = z0 + j / (nx * ny);
            return (R == 1 || max(max(abs(bx - c[0]), abs(by - c[1])), abs(bz - c[2])) == R) && box2(bx, by, bz) <= bound;  // (not the cube of the step before)
          });
        const bool ok = settled_within((double)R * B + face, !done);
        if (!done && (ok || R >= ra_max)) done = true;
        if (__builtin_amdgcn_ballot_w64(!done && fresh > 0) != 0ull) {  // somebody goes on with new candidates: the exact k-th distance so far prunes what follows
          have = merge();
          bound = fmin(bound, kth);
          fresh = 0;
        }
      }
    }
#ifdef GP_KNN_WAVELOG
    if (fl && sub == 0) fl[1] = __builtin_amdgcn_s_memrealtime(), fl[6] = done ? 1 : 0;
#endif
    // ---- phase B: shells of superblocks (blocks outside phase A's cube) ----
    if (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
      const double S = 4.0 * B;
      const int cs[3] = {c[0] >> 2, c[1] >> 2, c[2] >> 2};
      const double fsx = (ux - 4.0 * (double)cs[0]) * 0.25, fsy = (uy - 4.0 * (double)cs[1]) * 0.25, fsz = (uz - 4.0 * (double)cs[2]) * 0.25;
      const double sface = fmin(fmin(fmin(fsx, 1.0 - fsx), fmin(fsy, 1.0 - fsy)), fmin(fsz, 1.0 - fsz)) * S;
      int rs_max = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) rs_max = max(rs_max, max(cs[a], g.sdim[a] - 1 - cs[a]));
      for (int R = 0; __builtin_amdgcn_ballot_w64(!done) != 0ull; R++) {  // (the wave goes on while any of its groups does; a group ends at rs_max at the latest)
        // the SURFACE of the cube of radius R, enumerated directly (an outlier tens of metres from everything walks six shells: the cube's 2197 positions at R = 6 are
        // 866 on the surface): the two z-faces, (2R + 1)^2 positions each, then 2R - 1 slabs with the 8R positions of their rim
        const int side = 2 * R + 1, face_n = side * side, rim_n = 8 * R;
        const int total = done ? 0 : (R == 0 ? 1 : 2 * face_n + (side - 2) * rim_n);
        auto shell_pos = [&](int e, int& dx, int& dy, int& dz) {
          if (R == 0) {
            dx = dy = dz = 0;
          } else if (e < 2 * face_n) {
            const int f = e / face_n, r = e % face_n;
            dz = f ? R : -R;
            dx = r % side - R;
            dy = r / side - R;
          } else {
            const int r = e - 2 * face_n, slab = r / rim_n, pos = r % rim_n, edge = pos / (2 * R), off = pos % (2 * R);
            dz = slab - R + 1;
            dx = edge == 0 ? -R + off : (edge == 1 ? R : (edge == 2 ? R - off : -R));
            dy = edge == 0 ? -R : (edge == 1 ? -R + off : (edge == 2 ? R : R - off));
          }
        };
        int e = sub;                        // next position of the shell this lane looks at (stride kFarLanes)
        unsigned long long rest[kFarMasks];  // occupied blocks of the lane's current masks that are still to be listed
        unsigned long long spos[kFarMasks];  // their superblocks, packed (x | y << 21 | z << 42: a grid has < 2^24 blocks)
#pragma unroll
        for (int q = 0; q < kFarMasks; q++) rest[q] = 0ull, spos[q] = 0;
        bool lane_more = e < total;
        while (__builtin_amdgcn_ballot_w64(lane_more) != 0ull) {  // rounds of: fill the group's list (<= kFarList blocks), process it
          // (`bound`: the group's exact k-th distance as of the last stage -- a block farther away than that holds nothing of interest)
          if (sub == 0) L.count = 0;
          wave_sync();
          bool full = false;
          while (lane_more && !full) {
            bool any_rest = false;
#pragma unroll
            for (int q = 0; q < kFarMasks; q++) any_rest = any_rest || rest[q] != 0ull;
            if (!any_rest) {  // the next kFarMasks masks of this lane's positions, requested together
              if (e >= total) {
                lane_more = false;
                break;
              }
#pragma unroll
              for (int q = 0; q < kFarMasks; q++) {
                const int eq = e + q * kFarLanes;
                int dx = 0, dy = 0, dz = 0;
                shell_pos(eq < total ? eq : 0, dx, dy, dz);
                const int x = cs[0] + dx, y = cs[1] + dy, z = cs[2] + dz;
                const bool in = eq < total && x >= 0 && x < g.sdim[0] && y >= 0 && y < g.sdim[1] && z >= 0 && z < g.sdim[2];
                const unsigned long long mask = g.super[in ? ((size_t)z * (size_t)g.sdim[1] + (size_t)y) * (size_t)g.sdim[0] + (size_t)x : 0];
                rest[q] = in ? mask : 0ull;
                spos[q] = (unsigned long long)(unsigned)x | ((unsigned long long)(unsigned)y << 21) | ((unsigned long long)(unsigned)z << 42);
              }
              e += kFarMasks * kFarLanes;
            }
#pragma unroll
            for (int q = 0; q < kFarMasks; q++) {
              while (rest[q] != 0ull && !full) {
                const int bit = __ffsll((long long)rest[q]) - 1;
                const int bx = 4 * (int)(spos[q] & 0x1fffffull) + (bit & 3), by = 4 * (int)((spos[q] >> 21) & 0x1fffffull) + ((bit >> 2) & 3), bz = 4 * (int)(spos[q] >> 42) + (bit >> 4);
                bool want = max(max(abs(bx - c[0]), abs(by - c[1])), abs(bz - c[2])) > kFarBlockShells;  // (else: phase A scanned it)
                if (want) want = box2(bx, by, bz) <= bound;
                if (want) {
                  const int slot = atomicAdd(&L.count, 1);
                  if (slot >= kFarList) {  // the list is full: this block waits for the next round
                    full = true;
                    break;
                  }
                  L.blk[slot] = (bz * dim[1] + by) * dim[0] + bx;
                }
                rest[q] &= rest[q] - 1ull;
              }
            }
          }
          wave_sync();
          const int nb = done ? 0 : min(L.count, kFarList);
          wave_sync();
          // (Scanning an unbounded group's nearest blocks first -- eight at a time until it holds k candidates, the rest against that bound -- cuts an outlier's candidates
          // from 1500 to 350 per lane and not its time: the 160-260 us of such a query are the ~35 dependent mask trips of five empty shells, not the surface behind them.
          // Measured and removed: profiles/r05_c5_summary.txt item 7.)
          far_process(nb, [&](int j) { return (size_t)L.blk[j]; }, [](int) { return true; });
        }
#ifdef GP_KNN_WAVELOG
        if (!done) far_bshells++;
#endif
        const bool ok = settled_within((double)R * S + sface, !done);
        if (!done && (ok || R >= rs_max)) done = true;
        if (__builtin_amdgcn_ballot_w64(!done && fresh > 0) != 0ull) {  // (a shell that brought nothing leaves the bound as it is: no merge)
          have = merge();
          bound = fmin(bound, kth);
          fresh = 0;
        }
      }
    }
#ifdef GP_KNN_WAVELOG
    if (fl && sub == 0) fl[2] = __builtin_amdgcn_s_memrealtime(), fl[5] = far_bshells;
#endif
    // ---- the k smallest of the group's lists, ascending (ties: smaller index) ----
    have = merge();
#ifdef GP_KNN_WAVELOG
    if (fl && sub == 0) fl[3] = __builtin_amdgcn_s_memrealtime(), fl[4] = far_cands;
#endif
    if (active && sub == 0) {
      float* out = covs + 9 * (size_t)i;
      if (have < k) {
        atomicAdd(num_short, 1);
        for (int j = 0; j < 9; j++) out[j] = (j % 4 == 0) ? 1.0f : 0.0f;
      } else {
        TopK<KMAX, false> fin;
        fin.init(k, kInf);
#pragma unroll
        for (int r = 0; r < KMAX; r++) fin.idx[r] = fin_i[r];
        covariance_from_neighbours<KMAX, false>(fin, points, k, out);
      }
    }
  }
}

// far_list / far_count (optional): queries whose neighbourhood is sparse (knn_query_bins, `sparse`) are not searched lane by lane -- one lane walking hundreds of
// empty blocks holds its 63 neighbours for 400-600 us, the launch's tail (profiles/r05_c5_wavelog_near_first.txt) -- but appended here for covariance_far_kernel
template <int KMAX, int MIN_WAVES = 1, bool FULL = false>
__global__ void __launch_bounds__(128, MIN_WAVES) covariance_kernel(SearchView g, const float* __restrict__ points, int n, int k, float* __restrict__ covs,
                                                         int* __restrict__ num_short, const int* __restrict__ todo_list, const int* __restrict__ todo_count,
                                                         int* __restrict__ far_list = nullptr, int* __restrict__ far_count = nullptr, float* __restrict__ far_bound = nullptr,
                                                         const int* __restrict__ todo_begin = nullptr) {
  int t = blockIdx.x * 128 + threadIdx.x;
  if (todo_list) {  // positions [*todo_begin, *todo_count) of the list (round 5: the heavy part and the rest are two launches on two streams)
    if (todo_begin) t += *todo_begin;
    if (t >= *todo_count) return;
    t = todo_list[t];
  }
  if (t >= n) return;
  // queries are taken in the finest grid's cell-sorted order: the lanes of a wave then sit in the same or adjacent cells,
  // walk the same shells and read the same cell ranges (coherent loads, little divergence); results go to the original index
  const float4 self = g.binned ? g.bins[0].sorted[t] : g.hashed.lv[0].sorted[t];
  const int i = __float_as_int(self.w);
  const double qx = (double)self.x, qy = (double)self.y, qz = (double)self.z;
  TopK<KMAX, FULL> top;
  top.init(k, 1.7976931348623157e308);
  int2* rl = nullptr;
  if constexpr (FULL) {  // (the k = KMAX build of estimate_covariances: flat scan, see knn_query_bins)
    __shared__ int2 range_lists[kRangeCap * kRangeStride];
    static_assert(kRangeStride == 128, "one list per thread of this kernel's workgroups");
    rl = range_lists + threadIdx.x;
  }
  bool sparse = false;
  knn_query_any<KMAX, FULL, FULL>(g, qx, qy, qz, 2 * k, top, todo_list != nullptr, rl, far_list ? &sparse : nullptr);
  if (far_list) {
    // (position in the cell-sorted array and the k-th distance found so far, rounded up; one atomic per wave)
    const unsigned long long m = __ballot(sparse);
    if (m != 0ull) {
      const int lane = threadIdx.x & 63, first = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == first) base = atomicAdd(far_count, __popcll(m));
      base = __shfl(base, first, 64);
      if (sparse) {
        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
        far_list[slot] = t;
        far_bound[slot] = __double2float_ru(top.worst());
      }
    }
    if (sparse) return;
  }
  float* out = covs + 9 * (size_t)i;
  if (top.count() < k) {
    atomicAdd(num_short, 1);
    for (int j = 0; j < 9; j++) out[j] = (j % 4 == 0) ? 1.0f : 0.0f;
    return;
  }
  covariance_from_neighbours<KMAX, FULL>(top, points, k, out);
#ifdef GP_KNN_WAVELOG
  if (g.binned && g.bins[0].counters && (threadIdx.x & 63) == 0) {
    unsigned long long* wl = g.bins[0].counters + 8 + 8 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    wl[4] = __builtin_amdgcn_s_memrealtime();
  }
  if (g.binned && g.bins[0].counters) {  // cells the wave's 64 queries span (ordinal of the last lane's cell - ordinal of the first one's + 1), and the own cell's population
    const BinGridView& b = g.bins[0];
    const int cx = fast_floor(qx * b.inv_h), cy = fast_floor(qy * b.inv_h), cz = fast_floor(qz * b.inv_h);
    const size_t bi = ((size_t)((cz >> 2) - b.geom.lo[2]) * (size_t)b.geom.dim[1] + (size_t)((cy >> 2) - b.geom.lo[1])) * (size_t)b.geom.dim[0] + (size_t)((cx >> 2) - b.geom.lo[0]);
    const int4 raw = *reinterpret_cast<const int4*>(b.blocks + bi);
    const unsigned long long bits = ((unsigned long long)(unsigned)raw.y << 32) | (unsigned long long)(unsigned)raw.x;
    const int bit = (cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4);
    const int ord = raw.z + __popcll(bits & ((1ull << bit) - 1ull));
    const int pop = b.cell_start[ord + 1] - b.cell_start[ord];
    const int o0 = __builtin_amdgcn_readlane(ord, 0), o63 = __builtin_amdgcn_readlane(ord, 63), p0 = __builtin_amdgcn_readlane(pop, 0);
    const int blkpop = b.cell_start[raw.z + __popcll(bits)] - b.cell_start[raw.z];
    const int bp0 = __builtin_amdgcn_readlane(blkpop, 0);
    if ((threadIdx.x & 63) == 0) {
      unsigned long long* wl = g.bins[0].counters + 8 + 8 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
      wl[7] = (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) | ((unsigned long long)(o63 - o0 + 1) << 32);
      wl[6] = (unsigned long long)p0 | ((unsigned long long)bp0 << 32);
    }
  }
#endif
}

// estimate_covariances, tiled: ONE WAVE PER OCCUPIED CELL ROW (the <= 4 x-adjacent cells of one (y, z) row of a block).  The queries
// are the row's own points -- one contiguous range of the cell-sorted array, ~20-70 of them -- and the candidates are the points of the
// cells x_min-1 .. x_max+1 of the 3 x 3 rows around it: 27 (block, row, x-mask) pieces, each again ONE contiguous range because
// occupied cells of a row have consecutive ordinals.  The 27 lookups run on 27 lanes at once (one latency chain for the whole row
// instead of one per lane and cell), the candidates are staged through LDS with coalesced loads and scanned by every query lane with
// broadcast reads: no per-lane pointer chasing, no divergence in the scan loop, and only ~1.5x the candidates a single query needs
// (the block-sized tiles tried first scanned 15-30x: DESIGN.md section 4.8).  A query is settled when its k-th distance is no
// larger than its distance to the border of that region (>= one cell edge): every point outside is farther.  Anything else -- sparse
// neighbourhoods, rows too dense for one wave -- is appended to `todo_list` and goes through the per-lane search, so the result is
// exact either way.
constexpr int kRowThreads = 64;       // one wave per workgroup: __syncthreads() is free and rows finish independently
constexpr int kRowCand = 256;         // candidates per LDS chunk (4 KB)
constexpr int kRowMaxCand = 8192;     // denser neighbourhoods (near field) are left to the per-lane search
constexpr int kRowMaxQueries = 512;
constexpr int kTileQueue = 32;        // per-lane queue of candidates that passed the f32 filter (2 B each)
constexpr int kTileKeep = 12;         // f32 top list: k (<= 10) + 2 entries of slack for the exactness check

// f32 top list of the tiled kernel: same insertion rule as TopK, floats, compile-time indices only
struct TopF {
  float d[kTileKeep];
  int idx[kTileKeep];
  __device__ void init() {
#pragma unroll
    for (int j = 0; j < kTileKeep; j++) {
      d[j] = __builtin_inff();
      idx[j] = -1;
    }
  }
  __device__ float bound() const { return d[kTileKeep - 1]; }
  __device__ void push(int index, float dist) {
    if (!(dist < d[kTileKeep - 1])) return;
    bool placed = false;
#pragma unroll
    for (int j = kTileKeep - 1; j >= 0; j--) {
      if (!placed) {
        if (j > 0 && dist < d[j - 1]) {
          d[j] = d[j - 1];
          idx[j] = idx[j - 1];
        } else {
          d[j] = dist;
          idx[j] = index;
          placed = true;
        }
      }
    }
  }
};

// appends the sorted positions of the lanes with `flag` to todo_list (one atomic per wave; the order of the list does not matter:
// every leftover query writes its own output slot)

// Scan kernel.  The scan loop is an LDS broadcast read, an f32 distance, a compare and a 2-byte LDS append for the lanes whose
// candidate passes.  What passes is pushed into the lane's f32 top list only when a queue is full or the chunk ends -- then every lane
// is busy with its OWN candidates, instead of the whole wave executing an insertion whenever any one lane has a hit.  The pieces are
// scanned own row first, so the acceptance threshold is tight after the first few dozen candidates.  Per query the kernel leaves the
// kTileKeep nearest candidates by f32 distance (original indices), the kTileKeep-th f32 distance and the query's distance to the
// border of the scanned region; the exact decision is taken by covariance_settle_kernel below with all lanes busy (a row fills a
// quarter of a wave on average, and the f64 work is the expensive part).
struct RowScanOut {
  int* kept;     // [kTileKeep][nq] original indices (-1: none), by sorted position
  float* bound;  // [nq] kTileKeep-th f32 squared distance (inf: fewer candidates than that), < 0: row not scanned
  float* safe;   // [nq] distance to the border of the scanned region, rounded down
  int nq;
};

__global__ void __launch_bounds__(kRowThreads) covariance_rows_kernel(BinGridView g, const int* __restrict__ occ_blocks, RowScanOut out, int knock) {
  __shared__ float4 cand[kRowCand];
  __shared__ unsigned short queue[kTileQueue][kRowThreads];
  __shared__ int rstart[27], rpref[28];
  const int lane = threadIdx.x;
  const int row = blockIdx.x & 15;                  // y + 4 z inside the block
  const long long b = occ_blocks[blockIdx.x >> 4];  // work list: the occupied blocks only (a LiDAR box is >99 % empty blocks)
  const GridBlock me = g.blocks[b];
  const unsigned rowbits = (unsigned)(me.bits >> (4 * row)) & 0xFu;
  if (rowbits == 0u) return;
  const int ord0 = me.base + __popcll(me.bits & ((1ull << (4 * row)) - 1ull));
  const int q0 = g.cell_start[ord0];
  const int Q = g.cell_start[ord0 + __popc(rowbits)] - q0;
  const int dim0 = g.geom.dim[0], dim1 = g.geom.dim[1], dim2 = g.geom.dim[2];
  const int bx = (int)(b % dim0), by = (int)((b / dim0) % dim1), bz = (int)(b / ((long long)dim0 * dim1));
  // cell coordinates relative to the grid's first cell; the candidate region is x in [cx_lo, cx_hi], y in cy +- 1, z in cz +- 1
  const int cy = 4 * by + (row & 3), cz = 4 * bz + (row >> 2);
  const int cx_lo = 4 * bx + (__ffs((int)rowbits) - 1) - 1, cx_hi = 4 * bx + (31 - __clz((int)rowbits)) + 1;
  int len = 0;
  if (lane < 27) {
    // piece order: own row first, then the rows sharing a face with it, then the diagonal ones; own block column first in each
    const int t = lane / 3, u = lane % 3;
    const int dy = (int)((0x22161u >> (2 * t)) & 3u) - 1;  // two bits per entry: t = 0..8 -> dy = 0,-1,1, 0,0, -1,1,-1,1
    const int dz = (int)((0x28215u >> (2 * t)) & 3u) - 1;  //                                    dz = 0, 0,0,-1,1, -1,-1,1,1
    const int nbx = bx + (u == 0 ? 0 : (u == 1 ? -1 : 1)), ny = cy + dy, nz = cz + dz;
    int start = 0;
    const int lo = max(cx_lo - 4 * nbx, 0), hi = min(cx_hi - 4 * nbx, 3);  // cells of block column nbx inside the x-range
    if (lo <= hi && nbx >= 0 && nbx < dim0 && ny >= 0 && ny < 4 * dim1 && nz >= 0 && nz < 4 * dim2) {
      const GridBlock nb = g.blocks[((long long)(nz >> 2) * dim1 + (ny >> 2)) * dim0 + nbx];
      const int sh = 4 * ((ny & 3) + 4 * (nz & 3)) + lo;
      const unsigned m = (unsigned)(nb.bits >> sh) & ((2u << (hi - lo)) - 1u);
      if (m) {
        const int o = nb.base + __popcll(nb.bits & ((1ull << sh) - 1ull));
        start = g.cell_start[o];
        len = g.cell_start[o + __popc(m)] - start;
      }
    }
    rstart[lane] = start;
  }
  int incl = len;  // inclusive prefix of the 27 piece lengths across the lanes
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane < 27) rpref[lane + 1] = incl;
  if (lane == 0) rpref[0] = 0;
  __syncthreads();
  const int C = rpref[27];
  if (knock == 1) return;
  if (Q > kRowMaxQueries || C > kRowMaxCand) {
    for (int t = lane; t < Q; t += kRowThreads) out.bound[q0 + t] = -1.0f;
    return;
  }
  // the region's faces (metres)
  const double rlo[3] = {(double)(4 * g.geom.lo[0] + cx_lo) * g.h, (double)(4 * g.geom.lo[1] + cy - 1) * g.h, (double)(4 * g.geom.lo[2] + cz - 1) * g.h};
  const double rhi[3] = {(double)(4 * g.geom.lo[0] + cx_hi + 1) * g.h, (double)(4 * g.geom.lo[1] + cy + 2) * g.h, (double)(4 * g.geom.lo[2] + cz + 2) * g.h};
  for (int pass = 0; pass * kRowThreads < Q; pass++) {
    const int qi = pass * kRowThreads + lane;
    const bool active = qi < Q;
    const float4 self = g.sorted[q0 + (active ? qi : 0)];
    TopF top;
    top.init();
    int queued = 0;
    auto drain = [&]() {  // every lane inserts its own queued candidates (f32 distance recomputed from LDS)
      for (int i = 0; __any(i < queued); i++) {
        if (i < queued) {
          const float4 v = cand[queue[i][lane]];
          const float dxf = v.x - self.x, dyf = v.y - self.y, dzf = v.z - self.z;
          top.push(__float_as_int(v.w), dxf * dxf + dyf * dyf + dzf * dzf);
        }
      }
      queued = 0;
    };
    for (int c0 = 0; c0 < C; c0 += kRowCand) {
      __syncthreads();  // the previous chunk has been consumed
      const int cnt = min(kRowCand, C - c0);
      for (int i = lane; i < cnt; i += kRowThreads) {
        const int gi = c0 + i;
        int r = 0;
#pragma unroll
        for (int t = 1; t < 27; t++) r += (rpref[t] <= gi) ? 1 : 0;  // piece holding candidate gi (prefix sums are non-decreasing)
        cand[i] = g.sorted[rstart[r] + (gi - rpref[r])];
      }
      __syncthreads();
      if (knock == 2) continue;
      float thr = top.bound();
      // four candidates per step: the four broadcast reads are in flight together, one queue-full test per step
      const int cnt4 = cnt & ~3;
      for (int j = 0; j < cnt4; j += 4) {
        const float4 v0 = cand[j], v1 = cand[j + 1], v2 = cand[j + 2], v3 = cand[j + 3];
        const float ax = v0.x - self.x, ay = v0.y - self.y, az = v0.z - self.z;
        const float bx_ = v1.x - self.x, by_ = v1.y - self.y, bz_ = v1.z - self.z;
        const float cx_ = v2.x - self.x, cy_ = v2.y - self.y, cz_ = v2.z - self.z;
        const float dx_ = v3.x - self.x, dy_ = v3.y - self.y, dz_ = v3.z - self.z;
        const float d0 = ax * ax + ay * ay + az * az, d1 = bx_ * bx_ + by_ * by_ + bz_ * bz_;
        const float d2 = cx_ * cx_ + cy_ * cy_ + cz_ * cz_, d3 = dx_ * dx_ + dy_ * dy_ + dz_ * dz_;
        if (active) {
          if (d0 < thr) queue[queued++][lane] = (unsigned short)j;
          if (d1 < thr) queue[queued++][lane] = (unsigned short)(j + 1);
          if (d2 < thr) queue[queued++][lane] = (unsigned short)(j + 2);
          if (d3 < thr) queue[queued++][lane] = (unsigned short)(j + 3);
        }
        if (__any(queued > kTileQueue - 4)) {
          if (knock == 3) queued = 0;
          drain();
          thr = top.bound();
        }
      }
      for (int j = cnt4; j < cnt; j++) {
        const float4 v = cand[j];
        const float dxf = v.x - self.x, dyf = v.y - self.y, dzf = v.z - self.z;
        if (active && dxf * dxf + dyf * dyf + dzf * dzf < thr) queue[queued++][lane] = (unsigned short)j;
      }
      drain();  // the chunk is about to be replaced (at most kTileQueue - 4 + 3 entries are queued)
    }
    if (active) {
      const size_t pos = (size_t)q0 + qi;
#pragma unroll
      for (int j = 0; j < kTileKeep; j++) out.kept[(size_t)j * out.nq + pos] = top.idx[j];
      out.bound[pos] = top.bound();
      double safe = 1.0e300;
      const double q[3] = {(double)self.x, (double)self.y, (double)self.z};
#pragma unroll
      for (int a = 0; a < 3; a++) safe = fmin(safe, fmin(q[a] - rlo[a], rhi[a] - q[a]));
      out.safe[pos] = (float)fmax(safe, 0.0) * 0.999999f;
    }
  }
}

// Decision kernel, one query per lane in sorted order: exact re-score of the kept candidates in f64 (the reference compares doubles),
// in f32 rank order.  A query is settled only if (i) the k-th exact distance is below the kTileKeep-th f32 distance by more than f32
// rounding -- everything that was filtered out has an f32 distance >= that, i.e. a true distance >= bound * (1 - 1e-5), so it cannot
// belong to the k nearest -- and (ii) it is no larger than the distance to the region's border, so nothing outside the region can
// either.  The rest is listed for the per-lane search.
template <int KMAX>
__global__ void __launch_bounds__(128) covariance_settle_kernel(const float4* __restrict__ sorted, RowScanOut in, const float* __restrict__ points, int k,
                                                                float* __restrict__ covs, int* __restrict__ todo_list, int* __restrict__ todo_count) {
  static_assert(KMAX + 2 <= kTileKeep, "two entries of slack");
  const int pos = blockIdx.x * 128 + threadIdx.x;
  const bool active = pos < in.nq;
  bool leftover = false;
  if (active) {
    const float bound = in.bound[pos];
    leftover = true;
    if (bound >= 0.0f) {
      const float4 self = sorted[pos];
      const double q[3] = {(double)self.x, (double)self.y, (double)self.z};
      TopK<KMAX> exact;
      exact.init(k, 1.7976931348623157e308);
      int idx[kTileKeep];
#pragma unroll
      for (int j = 0; j < kTileKeep; j++) idx[j] = in.kept[(size_t)j * in.nq + pos];
#pragma unroll
      for (int j = 0; j < kTileKeep; j++) {
        if (idx[j] >= 0) {
          const size_t nb = (size_t)idx[j];
          const double ddx = (double)points[3 * nb] - q[0], ddy = (double)points[3 * nb + 1] - q[1], ddz = (double)points[3 * nb + 2] - q[2];
          exact.push(idx[j], ddx * ddx + ddy * ddy + ddz * ddz);
        }
      }
      const double safe = (double)in.safe[pos];
      const bool separated = exact.worst() <= (double)bound * (1.0 - 1.0e-5);
      if (exact.found >= k && separated && exact.worst() <= safe * safe) {
        covariance_from_neighbours<KMAX>(exact, points, k, covs + 9 * (size_t)__float_as_int(self.w));
        leftover = false;
      }
    }
  }
  todo_append(leftover, pos, todo_list, todo_count);
}

// ---- GICP: 1-NN correspondence within max distance + the same H/b algebra as VGICP ------------------------------------
struct GicpDesc {
  const float* points;
  const float* covs;
  const float* target_points;
  const float* target_covs;
  SearchView grid;
  int n;
  double max_sq_dist;
};

// the poses ride in the kernel arguments (no H2D copy in front of the launch)
struct GicpPoses {
  double lin[16], eval[16];
};

// correspondence pass of the GICP factor (IntegratedGICPFactor_::update_correspondences, integrated_gicp_factor_impl.hpp:132-172):
// corr[i] = index of the nearest target point of T_lin p_i with squared distance < max, or -1.  ONE query per lane and nothing else in
// the kernel: the fused search + algebra kernel below holds 32 f64 accumulators and the algebra's temporaries next to the search state
// (157 VGPRs: three waves per SIMD, and a 1 M-point cloud only brings 3.8), and the search is a chain of dependent round trips that
// only occupancy hides.  The stored correspondences are also what the reference's error() evaluates on (it does not search again).
// (92 VGPRs, five waves per SIMD.  Capped at 80 VGPRs -- six waves, eleven registers spilled -- it measured 2 % faster: not worth the scratch.)
__global__ void __launch_bounds__(256) gicp_correspond_kernel(GicpDesc f, const GicpPoses poses, int* __restrict__ corr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= f.n) return;
  const Pose Tl = load_pose(poses.lin);
  const double px = (double)f.points[3 * (size_t)i], py = (double)f.points[3 * (size_t)i + 1], pz = (double)f.points[3 * (size_t)i + 2];
  const double lx = Tl.r00 * px + Tl.r01 * py + Tl.r02 * pz + Tl.tx;
  const double ly = Tl.r10 * px + Tl.r11 * py + Tl.r12 * pz + Tl.ty;
  const double lz = Tl.r20 * px + Tl.r21 * py + Tl.r22 * pz + Tl.tz;
  TopK<1> top;
  top.init(1, f.max_sq_dist);
  knn_query_any<1>(f.grid, lx, ly, lz, 1, top);
  corr[i] = top.found ? top.idx[0] : -1;
}

// CORR: the correspondences come from gicp_correspond_kernel (corr[]) instead of a search of this kernel's own
template <int MODE, bool CORR = false>  // MODE_LIN (rigid pose: 29 sums + adjoint finalize), MODE_ERR, MODE_LIN_GENERAL (any 3x3 block: 92 explicit sums)
__global__ void __launch_bounds__(256) gicp_tile_kernel(GicpDesc f, const GicpPoses poses,
                                                        int tile_points, double* __restrict__ partials, const int* __restrict__ corr = nullptr) {
  constexpr int NACC = MODE == MODE_ERR ? 2 : (MODE == MODE_LIN ? ACC_SIZE : ACCG_SIZE);
  constexpr int STRIDE = MODE == MODE_LIN_GENERAL ? ACCG_STRIDE : ACC_STRIDE;
  constexpr int NREG = MODE == MODE_LIN_GENERAL ? ACCG_SIZE : 32;
  const Pose Tl = load_pose(poses.lin);
  const Pose Te = MODE == MODE_ERR ? load_pose(poses.eval) : Tl;
  double acc[NREG];
#pragma unroll
  for (int k = 0; k < NREG; k++) acc[k] = 0.0;
  const int begin = blockIdx.x * tile_points;
  const int end = min(begin + tile_points, f.n);
  for (int i = begin + threadIdx.x; i < end; i += 256) {
    const double px = (double)f.points[3 * (size_t)i], py = (double)f.points[3 * (size_t)i + 1], pz = (double)f.points[3 * (size_t)i + 2];
    const double lx = Tl.r00 * px + Tl.r01 * py + Tl.r02 * pz + Tl.tx;
    const double ly = Tl.r10 * px + Tl.r11 * py + Tl.r12 * pz + Tl.ty;
    const double lz = Tl.r20 * px + Tl.r21 * py + Tl.r22 * pz + Tl.tz;
    // correspondence: nearest target point with sq_dist < max (integrated_gicp_factor_impl.hpp:166-170)
    size_t j;
    if constexpr (CORR) {
      const int c = corr[i];
      if (c < 0) continue;
      j = (size_t)c;
    } else {
      TopK<1> top;
      top.init(1, f.max_sq_dist);
      knn_query_any<1>(f.grid, lx, ly, lz, 1, top);
      if (top.found == 0) continue;
      j = (size_t)top.idx[0];
    }
    const float* cp = f.covs + 9 * (size_t)i;
    const float* cq = f.target_covs + 9 * j;
    // reuse the VGICP per-point algebra: the "voxel" is the matched target point (mu_B, C_B)
    const double mux = (double)f.target_points[3 * j], muy = (double)f.target_points[3 * j + 1], muz = (double)f.target_points[3 * j + 2];
    // symmetric parts of both column-major 3x3 covariances (exactly the inputs when they are symmetric)
    const double cb[6] = {(double)cq[0], 0.5 * ((double)cq[3] + (double)cq[1]), 0.5 * ((double)cq[6] + (double)cq[2]),
                          (double)cq[4], 0.5 * ((double)cq[7] + (double)cq[5]), (double)cq[8]};
    if constexpr (MODE == MODE_LIN_GENERAL) {
      const double ca[6] = {(double)cp[0], 0.5 * ((double)cp[3] + (double)cp[1]), 0.5 * ((double)cp[6] + (double)cp[2]),
                            (double)cp[4], 0.5 * ((double)cp[7] + (double)cp[5]), (double)cp[8]};
      double m[6];
      fused_mahalanobis(Tl, ca, cb, m);
      accumulate_sums<MODE_LIN_GENERAL>(Tl, m, px, py, pz, lx, ly, lz, mux - lx, muy - ly, muz - lz, acc);
    } else {
      const v2d c01 = {cb[0], cb[1]}, c23 = {cb[2], cb[3]}, c45 = {cb[4], cb[5]};
      accumulate_terms_mu<MODE, double>(Tl, Te, (float)px, (float)py, (float)pz, cp, mux, muy, muz, c01, c23, c45, acc);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double lds[4][STRIDE];
  if constexpr (MODE == MODE_LIN) {
    const double s = butterfly_reduce32(acc, lane);
    if ((lane & 1) == 0) lds[wave][butterfly_component(lane)] = s;
  } else {
#pragma unroll
    for (int k = 0; k < NACC; k++) {
      double v = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) lds[wave][k] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < STRIDE) {
    double s = 0.0;
    if (threadIdx.x < NACC) s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    partials[(size_t)blockIdx.x * STRIDE + threadIdx.x] = s;
  }
}

}  // namespace gp

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

struct gp_grid_level {
  gp::DeviceArray arena;  // one allocation: keys | start | sorted (device allocations cost far more than the build kernels)
  void *keys_p = nullptr, *start_p = nullptr, *sorted_p = nullptr;
  uint32_t mask = 0;
  int n = 0;
  double h = 0.0;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  gp::GridView view() const {
    gp::GridView g;
    g.keys = static_cast<const unsigned long long*>(keys_p);
    g.start = static_cast<const int*>(start_p);
    g.sorted = static_cast<const float4*>(sorted_p);
    g.mask = mask;
    g.n = n;
    g.h = h;
    g.inv_h = 1.0 / h;
    for (int a = 0; a < 3; a++) {
      g.lo[a] = lo[a];
      g.hi[a] = hi[a];
    }
    return g;
  }
};


struct gp_point_grid {
  // default: the binned structure (gp_binning.hpp) + the cell-sorted copy of the points, in up to kMaxLevels levels (cell x4 each)
  struct BinLevel {
    gp::PointBins bins;
    gp::DeviceArray sorted;  // float4[num_binned]
    gp::DeviceArray super;   // unsigned long long[sdim product]
    int sdim[3] = {0, 0, 0};
    double h = 0.0;
  };
  std::vector<std::unique_ptr<BinLevel>> bin_levels;
  bool binned = false;
  int num_binned = 0;
  // fallback for clouds whose bounding box is too large for the block grid: hashed multi-level grid
  std::vector<std::unique_ptr<gp_grid_level>> levels;
  hipStream_t stream = nullptr;
  int structure = 0;                       // GP_TUNE_KNN_STRUCTURE value the grid was created with (per structure, nothing process-global)
  unsigned long long* counters = nullptr;  // caller's device buffer of 8 work counters, or null (gp_point_grid_create_ex; measurement only)
  gp::SearchView view() const {
    gp::SearchView v{};
    v.binned = binned ? (int)bin_levels.size() : 0;
    // round 4 measured both knobs on the 1 M-point cloud (scripts/r04_c5.py, profiles/r04_c5_staging.jsonl): fine shells 0 .. 4 with and without two or three shells
    // of blocks in front of the superblocks -- 1.02-1.05 ms per call whatever the staging: 1434 of 10^6 queries get past the fine shells at all
    // (profiles/r04_c5_wavelog.txt).  The defaults stay round 3's.
    v.block_stage = 0;
    v.fine_shells = 4;
    if (structure >= 16) {  // experiment encoding (scripts/r04_c5.py): 16 | fine shells << 4 | block shells << 8
      v.fine_shells = (structure >> 4) & 7;
      v.block_stage = (structure >> 8) & 7;
    }
    if (binned) {
      for (size_t l = 0; l < bin_levels.size(); l++) {
        const BinLevel& b = *bin_levels[l];
        v.bins[l].blocks = b.bins.blocks.as<gp::GridBlock>();
        v.bins[l].cell_start = b.bins.cell_start.as<int>();
        v.bins[l].sorted = b.sorted.as<float4>();
        v.bins[l].geom = b.bins.geom;
        v.bins[l].inv_h = 1.0 / b.h;
        v.bins[l].h = b.h;
        v.bins[l].n = b.bins.num_binned;
        v.bins[l].super = b.super.as<unsigned long long>();
        for (int a = 0; a < 3; a++) v.bins[l].sdim[a] = b.sdim[a];
        v.bins[l].counters = counters;
      }
    } else {
      v.hashed.num_levels = (int)levels.size();
      for (int l = 0; l < v.hashed.num_levels; l++) v.hashed.lv[l] = levels[l]->view();
    }
    return v;
  }
};


struct gp_gicp_factor {
  gp_point_grid* grid = nullptr;
  gp::GicpDesc desc{};
  hipStream_t stream = nullptr;
  int tile_points = 1024;
  int num_tiles = 0;
  gp::DeviceArray partials, d_poses, d_out;
  gp::PinnedArray h_out;
  void* h_out_dev = nullptr;
  gp::PinnedArray h_done;  // completion word of the synchronous calls (gp_vgicp_shared.hpp: DoneFlags)
  void* h_done_dev = nullptr;
  unsigned long long seq = 0;
  gp::DeviceArray corr;          // [n] correspondences of the last correspondence pass (gicp_correspond_kernel)
  double corr_pose[16] = {0};    // ... and the linearisation pose they belong to
  bool corr_valid = false;
};

// correspondences at `pose_lin`.  A linearise always searches (IntegratedGICPFactor_::linearize calls update_correspondences every time,
// the default update tolerances being zero); an error evaluation re-uses the stored correspondences when they belong to its
// linearisation pose -- the reference's error() evaluates on the correspondences of the last linearise (impl.hpp:183-185).
static int gicp_correspond(gp_gicp_factor* f, const gp::GicpPoses& P, bool reuse) {
  constexpr bool split = true;  // the correspondence pass is its own kernel (the fused kernel: 0.233 vs 0.160 ms per linearise, profiles/r02_gicp_split_ab.jsonl)
  if (!split || f->desc.n <= 0) return 0;
  if (!f->corr.ptr) {
    if (f->corr.alloc(sizeof(int) * (size_t)f->desc.n) != GP_OK) return 0;  // no memory for the index array: the fused kernel still works
    f->corr_valid = false;
  }
  if (!reuse || !f->corr_valid || memcmp(f->corr_pose, P.lin, sizeof(double) * 16) != 0) {
    hipLaunchKernelGGL(gp::gicp_correspond_kernel, dim3((f->desc.n + 255) / 256), dim3(256), 0, f->stream, f->desc, P, f->corr.as<int>());
    memcpy(f->corr_pose, P.lin, sizeof(double) * 16);
    f->corr_valid = true;
  }
  return 1;
}

extern "C" {

static inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

static uint32_t level_slots(int n) {
  uint32_t slots = 1024;
  while (slots < 2u * (uint32_t)std::max(n, 1)) slots <<= 1;  // at most n distinct cells -> load factor <= 0.5
  return slots;
}

// scratch of one level build (counts | cursor | point_slot | block_sums | total | bbox), reused by every level of a grid
static size_t level_scratch_bytes(int n) {
  const uint32_t slots = level_slots(n);
  const size_t nb = (slots + gp::kScanBlock - 1) / gp::kScanBlock;
  return 2 * align256(sizeof(int) * slots) + align256(sizeof(int) * (size_t)std::max(n, 1)) + align256(sizeof(int) * nb) + align256(64) +
         align256(sizeof(int) * 6 * (((size_t)std::max(n, 1) + 255) / 256));
}

static int build_level(const float* points_dev, int n, double cell_size, hipStream_t s, char* scratch, int* bbox, gp_grid_level** out) {
  auto* g = new gp_grid_level;
  g->n = n;
  g->h = cell_size;
  const uint32_t slots = level_slots(n);
  g->mask = slots - 1;
  const int nb = (int)((slots + gp::kScanBlock - 1) / gp::kScanBlock);
  const size_t keys_b = align256(sizeof(unsigned long long) * slots), start_b = align256(sizeof(int) * ((size_t)slots + 1)),
               sorted_b = align256(sizeof(float4) * (size_t)std::max(n, 1));
  const int rc = g->arena.alloc_async(keys_b + start_b + sorted_b, s);
  if (rc != GP_OK) {
    delete g;
    return rc;
  }
  g->keys_p = g->arena.as<char>();
  g->start_p = g->arena.as<char>() + keys_b;
  g->sorted_p = g->arena.as<char>() + keys_b + start_b;
  unsigned long long* keys = static_cast<unsigned long long*>(g->keys_p);
  int* start = static_cast<int*>(g->start_p);
  float4* sorted = static_cast<float4*>(g->sorted_p);
  char* cur = scratch;
  int* counts = reinterpret_cast<int*>(cur);
  cur += align256(sizeof(int) * slots);
  int* cursor = reinterpret_cast<int*>(cur);
  cur += align256(sizeof(int) * slots);
  int* point_slot = reinterpret_cast<int*>(cur);
  cur += align256(sizeof(int) * (size_t)std::max(n, 1));
  int* block_sums = reinterpret_cast<int*>(cur);
  cur += align256(sizeof(int) * (size_t)nb);
  int* total = reinterpret_cast<int*>(cur);
  cur += align256(64);
  int* block_boxes = reinterpret_cast<int*>(cur);
  GP_HIP(hipMemsetAsync(keys, 0xff, sizeof(unsigned long long) * slots, s));
  GP_HIP(hipMemsetAsync(counts, 0, 2 * align256(sizeof(int) * slots), s));  // counts and cursor are adjacent
  if (n > 0) {
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(gp::grid_insert_kernel, dim3(blocks), dim3(256), 0, s, points_dev, n, 1.0 / cell_size, keys, counts, point_slot, g->mask, block_boxes);
    hipLaunchKernelGGL(gp::bbox_reduce_kernel, dim3(1), dim3(256), 0, s, block_boxes, blocks, bbox);
  }
  hipLaunchKernelGGL(gp::scan_block_kernel, dim3(nb), dim3(gp::kScanBlock), 0, s, counts, start, block_sums, (int)slots);
  hipLaunchKernelGGL(gp::scan_sums_kernel, dim3(1), dim3(gp::kScanBlock), 0, s, block_sums, nb, total);
  hipLaunchKernelGGL(gp::scan_add_kernel, dim3(nb), dim3(gp::kScanBlock), 0, s, start, block_sums, (int)slots, total);
  if (n > 0) {
    hipLaunchKernelGGL(gp::grid_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, points_dev, n, point_slot, start, cursor, sorted);
  }
  GP_HIP(hipGetLastError());
  // no synchronisation here: the next level reuses the scratch in stream order; the caller fetches all bounding boxes at once
  *out = g;
  return GP_OK;
}

}  // extern "C" (re-opened below)

extern "C" {

// levels: cell_size, 4 cell_size, 16 cell_size (coarser levels only when the cloud is large enough to need them)
int gp_point_grid_create(const float* points_dev, int n, double cell_size, gp_stream_t stream, gp_point_grid_t** out) {
  return gp_point_grid_create_ex(points_dev, n, cell_size, 0, nullptr, stream, out);
}

// structure: GP_TUNE_KNN_STRUCTURE value (0 binned + per-lane search, 1 hashed multi-level grid, 3 row-tiled covariance pass first, 4 two binned
// levels); counters_dev: device buffer of 8 uint64 work counters (measurement) or null
namespace gp {
// ---- the side stream of gp_estimate_covariances: a hardware queue that does NOT share a dispatch pipe with the caller's ----
// Two kernels on two HIP streams overlap only when their hardware queues sit on different pipes of the command processor: a pipe dispatches ONE grid at a time, and a
// grid with more workgroups than the device holds keeps its pipe until its last workgroup has been placed.  Measured (profiles/r05_c5_queue_pipes.txt): with the
// caller on queue 1 and the side stream on queue 3 the second covariance launch starts 8 us after the first; on queue 5 (the same process after bench.py's C4 phase had
// taken queues 2-4) it starts when the first launch's last workgroup is placed, 100 / 160 us later on the two 1 M-point clouds -- 0.07 ms per call.  HIP does not say
// which queue a stream gets, so the library asks the device: up to four low-priority streams are created (each gets a queue of its own from the runtime's pool for that
// priority), and the candidates are probed ONCE per host thread and device, beside the first caller stream that thread brings -- a grid of 6144 workgroups that holds two
// LDS-bound workgroups per CU for ~5 us each on the caller's stream, and one wave on the candidate that reports how long after the grid's first workgroup it got to run
// (~1 us on another pipe, the grid's whole dispatch on the same one).  The candidate with the shortest delay serves the thread's calls on that device from then on
// (~0.4 ms once).  Round 5 probed per caller stream (a 16-entry table keyed on the raw handle): an application that cycles streams re-paid 0.4 ms per new handle to save
// 0.07 ms per call, and a recycled handle inherited a stale choice -- a later caller stream on another pipe now simply keeps the first choice (correct either
// way: the choice only decides whether the two launches overlap).  gp_trim_device_cache() releases the thread's candidate streams, events and probe words.
__global__ void __launch_bounds__(256) pipe_probe_hog_kernel(unsigned long long* __restrict__ words, int ticks) {
  __shared__ float pad[12 * 1024];  // 48 KB: three workgroups per CU (160 KB of LDS), the wave slots stay free for the probe's wave
  pad[threadIdx.x * 48] = (float)threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(words, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (pad[(threadIdx.x * 48 + 7) % (12 * 1024)] < -1.0f) words[2] = 1;  // (keeps the array)
}
__global__ void __launch_bounds__(64) pipe_probe_stamp_kernel(unsigned long long* __restrict__ words) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long seen = 0;
  while ((seen = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < 200000ull) __builtin_amdgcn_s_sleep(4);  // (<= 2 ms)
  words[1] = __builtin_amdgcn_s_memrealtime();
}

struct SideStreamHandles {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
struct SideStream : SideStreamHandles {
  static constexpr int kCandidates = 4;
  struct PerDevice {
    SideStreamHandles cand[kCandidates];
    int num = 0;
    unsigned long long* words = nullptr;  // device: [0] first workgroup of the hog started, [1] the probe's wave ran, [2] unused
    struct Choice {
      hipStream_t caller;
      int index;
      float delay_us[kCandidates];
    } chosen{nullptr, 0, {-1.f, -1.f, -1.f, -1.f}};
    bool probed = false;
    bool created = false;
  };
  // releases what this thread holds for the current device (gp_trim_device_cache): the next covariance call creates and probes again
  static void release() {
    PerDevice* c = device_state();
    if (!c || !c->created) return;
    for (int i = 0; i < c->num; i++) {
      if (c->cand[i].stream) (void)hipStreamSynchronize(c->cand[i].stream), (void)hipStreamDestroy(c->cand[i].stream);
      if (c->cand[i].fork) (void)hipEventDestroy(c->cand[i].fork);
      if (c->cand[i].join) (void)hipEventDestroy(c->cand[i].join);
    }
    if (c->words) (void)hipFree(c->words);
    (void)hipGetLastError();
    *c = PerDevice{};
  }
  static PerDevice* device_state() {
    static thread_local PerDevice cache[16];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) return nullptr;
    return &cache[d];
  }
  static void create(PerDevice& c) {
    c.created = true;
    // the LOWEST priority the device offers: what runs on the side stream fills the slots the caller's stream leaves free, it does not compete for them
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0, (void)hipGetLastError();
    for (int i = 0; i < kCandidates; i++) {
      SideStreamHandles n;
      if (hipStreamCreateWithPriority(&n.stream, hipStreamNonBlocking, least) != hipSuccess || hipEventCreateWithFlags(&n.fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&n.join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      c.cand[c.num++] = n;
    }
    if (c.num > 1 && hipMalloc(&c.words, 4 * sizeof(unsigned long long)) != hipSuccess) (void)hipGetLastError(), c.words = nullptr;
  }
  // delay (us) between the first workgroup of a pipe-filling grid on `caller` and a wave on the candidate; < 0: the probe did not run
  static float probe(PerDevice& c, hipStream_t caller, const SideStreamHandles& n) {
    unsigned long long h[2] = {0, 0};
    if (hipMemsetAsync(c.words, 0, 4 * sizeof(unsigned long long), caller) != hipSuccess || hipEventRecord(n.fork, caller) != hipSuccess ||
        hipStreamWaitEvent(n.stream, n.fork, 0) != hipSuccess) {
      (void)hipGetLastError();
      return -1.f;
    }
    hipLaunchKernelGGL(pipe_probe_hog_kernel, dim3(6144), dim3(256), 0, caller, c.words, 500);
    hipLaunchKernelGGL(pipe_probe_stamp_kernel, dim3(1), dim3(64), 0, n.stream, c.words);
    if (hipEventRecord(n.join, n.stream) != hipSuccess || hipStreamWaitEvent(caller, n.join, 0) != hipSuccess) (void)hipStreamSynchronize(n.stream);
    if (hipStreamSynchronize(caller) != hipSuccess || hipMemcpy(h, c.words, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || h[0] == 0 || h[1] == 0) {
      (void)hipGetLastError();
      return -1.f;
    }
    return h[1] >= h[0] ? (float)(h[1] - h[0]) / 100.f : 0.f;  // 100 MHz
  }
  static int get(hipStream_t caller, SideStream* out, const PerDevice::Choice** report = nullptr) {
    PerDevice* c = device_state();
    if (!c) return fail(GP_ERROR_HIP, "SideStream: no current device");
    if (!c->created) create(*c);
    if (c->num == 0) return fail(GP_ERROR_HIP, "SideStream: cannot create the side stream");
    if (c->probed) {
      static_cast<SideStreamHandles&>(*out) = c->cand[c->chosen.index];
      if (report) *report = &c->chosen;
      return GP_OK;
    }
    PerDevice::Choice pick{caller, 0, {-1.f, -1.f, -1.f, -1.f}};
    c->probed = true;  // (once per thread and device, whatever comes of it)
    if (c->words) {
      for (int i = 0; i < c->num; i++) {
        // (two probes, the smaller delay: a first launch on a new queue pays for the queue)
        const float a = probe(*c, caller, c->cand[i]), b = probe(*c, caller, c->cand[i]);
        pick.delay_us[i] = a < 0.f ? b : b < 0.f ? a : std::min(a, b);
        if (pick.delay_us[i] >= 0.f && (pick.delay_us[pick.index] < 0.f || pick.delay_us[i] < pick.delay_us[pick.index] - 2.f)) pick.index = i;  // (ties within 2 us: the earlier one)
      }
    }
    c->chosen = pick;
    if (report) *report = &c->chosen;
    static_cast<SideStreamHandles&>(*out) = c->cand[pick.index];
    return GP_OK;
  }
};
extern "C++" void release_side_streams() { SideStream::release(); }  // (gp_trim_device_cache, gp_runtime.hip; declared in gp_host.hpp)
}  // namespace gp

// measurement / tests: the side stream gp_estimate_covariances uses beside `caller` on the current device -- the delays (us) the pipe probe measured for the (up to four)
// candidate streams (< 0: not probed) and the index of the one in use
int gp_debug_side_stream_probe(gp_stream_t caller, float delays_us[4], int* chosen) {
  gp::SideStream side;
  const gp::SideStream::PerDevice::Choice* report = nullptr;
  GP_TRY(gp::SideStream::get((hipStream_t)caller, &side, &report));
  for (int i = 0; i < 4; i++)
    if (delays_us) delays_us[i] = report ? report->delay_us[i] : -1.f;
  if (chosen) *chosen = report ? report->index : 0;
  return GP_OK;
}

static int point_grid_create_impl(const float* points_dev, int n, double cell_size, int structure, unsigned long long* counters_dev, gp_stream_t stream, bool keep_cell_of,
                                  bool synchronise, gp_point_grid_t** out, const gp::FillJob caller_zero = gp::FillJob{}, bool* caller_zero_applied = nullptr);
int gp_point_grid_create_ex(const float* points_dev, int n, double cell_size, int structure, unsigned long long* counters_dev, gp_stream_t stream, gp_point_grid_t** out) {
  // (synchronised: gp_knn_search takes a stream of its own, which need not be the one the structure was built on)
  return point_grid_create_impl(points_dev, n, cell_size, structure, counters_dev, stream, false, true, out);
}
// keep_cell_of: the cell ordinals of the sorted positions stay with the first level (gp_estimate_covariances orders its queries by them);
// synchronise = false: the caller searches on `stream` itself, the last kernels of the build need not be waited for
// caller_zero: a fill the caller wants done on the stream before it searches; it rides in one of the build's kernels when the binned build runs (*caller_zero_applied)
static int point_grid_create_impl(const float* points_dev, int n, double cell_size, int structure, unsigned long long* counters_dev, gp_stream_t stream, bool keep_cell_of,
                                  bool synchronise, gp_point_grid_t** out, const gp::FillJob caller_zero, bool* caller_zero_applied) {
  bool caller_zero_done = false;
  struct Report {
    bool* out;
    const bool* done;
    ~Report() {
      if (out) *out = *done;
    }
  } report{caller_zero_applied, &caller_zero_done};
  if (!points_dev || n < 0 || !(cell_size > 0.0) || !out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_point_grid_create: bad arguments");
  if (structure != 0 && structure != 1 && structure != 3 && structure != 4 && structure != 6 && structure != 7 && structure < 16)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_point_grid_create_ex: structure in {0, 1, 3, 4, 6, 7} (>= 16: staging experiment)");
  auto* g = new gp_point_grid;
  g->stream = (hipStream_t)stream;
  g->structure = structure;
  g->counters = counters_dev;
  const bool force_hashed_grid = structure == 1;
  const int knn_levels = structure == 4 ? 2 : 1;  // binned levels (cell size x4 each); the blocks of the last one serve as the coarse level
  if (n > 0 && !force_hashed_grid) {
    // binned levels h, 4h, 16h (one level for small clouds and for radius-bounded searches, which never leave the first shells)
    const int want_levels = (n > 4096 && knn_levels > 1) ? std::min(knn_levels, gp::kMaxLevels) : 1;
    int rc = GP_OK;
    bool ok = true;
    double h = cell_size;
    for (int l = 0; l < want_levels && ok && rc == GP_OK; l++, h *= 4.0) {
      auto lv = std::make_unique<gp_point_grid::BinLevel>();
      bool too_large = false;
      rc = gp::bin_points(points_dev, n, 1.0 / h, g->stream, &lv->bins, &too_large);
      if (rc != GP_OK) break;
      if (too_large || lv->bins.num_cells <= 0) {
        ok = false;
        break;
      }
      rc = lv->sorted.alloc_pooled(sizeof(float4) * (size_t)std::max(lv->bins.num_binned, 1), g->stream);
      if (rc != GP_OK) break;
      // superblock occupancy (coarse stage of the search): zeroed by the gather kernel on its way, marked by the kernel behind it
      size_t sn = 1;
      for (int a = 0; a < 3; a++) {
        lv->sdim[a] = (lv->bins.geom.dim[a] + 3) / 4;
        sn *= (size_t)lv->sdim[a];
      }
      const size_t super_bytes = (sizeof(unsigned long long) * sn + 255) & ~size_t(255);
      rc = lv->super.alloc_pooled(super_bytes, g->stream);
      if (rc != GP_OK) break;
      hipLaunchKernelGGL(gp::gather_sorted_kernel, dim3((std::max(lv->bins.num_binned, 1) + 255) / 256), dim3(256), 0, g->stream, points_dev, (const int*)lv->bins.order.as<int>(),
                         lv->bins.num_binned, lv->sorted.as<float4>(), gp::fill_job(lv->super.ptr, super_bytes, 0u));
      // (launched also without occupied blocks when it carries the caller's fill)
      if (lv->bins.num_occ_blocks > 0 || (l == 0 && caller_zero.count > 0))
        hipLaunchKernelGGL(gp::super_mark_kernel, dim3((std::max(lv->bins.num_occ_blocks, 1) + 255) / 256), dim3(256), 0, g->stream, (const int*)lv->bins.occ_blocks.as<int>(),
                           lv->bins.num_occ_blocks, lv->bins.geom, lv->sdim[0], lv->sdim[1], lv->super.as<unsigned long long>(), l == 0 ? caller_zero : gp::FillJob{});
      if (l == 0) caller_zero_done = true;
      {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = gp::hip_fail(e, "gather_sorted_kernel", __FILE__, __LINE__);
      }
      if (synchronise && rc == GP_OK) {
        const hipError_t e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) rc = gp::hip_fail(e, "gather_sorted_kernel", __FILE__, __LINE__);
      }
      // (the arrays below go back to the pool in stream order)
      lv->h = h;
      lv->bins.order.release_on(g->stream);  // only the sorted copy is searched
      if (!(keep_cell_of && l == 0)) lv->bins.cell_of.release_on(g->stream);
      lv->bins.cell_block.release_on(g->stream);
      g->bin_levels.push_back(std::move(lv));
    }
    if (rc != GP_OK) {
      delete g;
      return rc;
    }
    if (ok && !g->bin_levels.empty()) {
      g->binned = true;
      g->num_binned = g->bin_levels[0]->bins.num_binned;
      *out = g;
      return GP_OK;
    }
    g->bin_levels.clear();  // bounding box too large for the block grid: hashed fallback below
  }
  const int num_levels = n > 4096 ? gp::kMaxLevels : 1;
  gp::DeviceArray scratch;
  {
    const int rc = scratch.alloc_async(level_scratch_bytes(n), g->stream);
    if (rc != GP_OK) {
      delete g;
      return rc;
    }
  }
  gp::DeviceArray d_bbox;
  int h_bbox[6 * gp::kMaxLevels];
  {
    const int rc = d_bbox.alloc_async(sizeof(int) * 6 * gp::kMaxLevels, g->stream);
    if (rc != GP_OK) {
      delete g;
      return rc;
    }
  }
  double h = cell_size;
  for (int l = 0; l < num_levels; l++, h *= 4.0) {
    gp_grid_level* lv = nullptr;
    const int rc = build_level(points_dev, n, h, g->stream, scratch.as<char>(), d_bbox.as<int>() + 6 * l, &lv);
    if (rc != GP_OK) {
      delete g;
      return rc;
    }
    g->levels.emplace_back(lv);
  }
  // one copy + one synchronisation for the whole structure (the scratch and d_bbox go back to the pool on return)
  hipError_t e = hipMemcpyAsync(h_bbox, d_bbox.ptr, sizeof(int) * 6 * (size_t)num_levels, hipMemcpyDeviceToHost, g->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  if (e != hipSuccess) {
    delete g;
    return gp::hip_fail(e, "gp_point_grid_create", __FILE__, __LINE__);
  }
  if (n > 0) {
    const int* hb = h_bbox;
    for (int l = 0; l < num_levels; l++)
      for (int a = 0; a < 3; a++) {
        g->levels[l]->lo[a] = hb[6 * l + a];
        g->levels[l]->hi[a] = hb[6 * l + 3 + a];
      }
  }
  *out = g;
  return GP_OK;
}

int gp_point_grid_destroy(gp_point_grid_t* g) {
  if (!g) return GP_OK;
  // the arenas come from the stream-ordered pool and are returned to it in the order of the creation stream: searches issued
  // on other streams must have finished first (hipFree used to imply this)
  (void)hipDeviceSynchronize();
  delete g;
  return GP_OK;
}

int gp_knn_search(const gp_point_grid_t* g, const float* queries_dev, int nq, int k, double max_sq_dist, int* indices_dev, double* sq_dists_dev, int* num_found_dev,
                  gp_stream_t stream) {
  if (!g || !queries_dev || nq < 0 || k <= 0 || k > 32 || !indices_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_knn_search: bad arguments (1 <= k <= 32)");
  if (nq == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  const gp::SearchView v = g->view();
  const dim3 grid((nq + 127) / 128), block(128);
  if (k == 1)
    hipLaunchKernelGGL(gp::knn_kernel<1>, grid, block, 0, s, v, queries_dev, nq, k, max_sq_dist, indices_dev, sq_dists_dev, num_found_dev);
  else if (k <= 10)
    hipLaunchKernelGGL(gp::knn_kernel<10>, grid, block, 0, s, v, queries_dev, nq, k, max_sq_dist, indices_dev, sq_dists_dev, num_found_dev);
  else
    hipLaunchKernelGGL(gp::knn_kernel<32>, grid, block, 0, s, v, queries_dev, nq, k, max_sq_dist, indices_dev, sq_dists_dev, num_found_dev);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

int gp_estimate_covariances(const float* points_dev, int n, int k, double cell_size, float* covs_dev, int* num_short, gp_stream_t stream) {
  return gp_estimate_covariances_ex(points_dev, n, k, cell_size, covs_dev, num_short, 0, nullptr, stream);
}

int gp_estimate_covariances_ex(const float* points_dev, int n, int k, double cell_size, float* covs_dev, int* num_short, int structure, unsigned long long* counters_dev,
                               gp_stream_t stream) {
  if (!points_dev || n < 0 || k <= 0 || k > 32 || !covs_dev) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_estimate_covariances: bad arguments (1 <= k <= 32)");
  if (num_short) *num_short = 0;
  if (n == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  gp_point_grid_t* g = nullptr;
  // one zeroed block: [0] the count of queries with fewer than k neighbours, [256 B ..) the look-back state of the heavy-first scan (sized for one cell per point).
  // It is zeroed by one of the structure build's kernels on its way (gp_host.hpp, FillJob); only the hashed fallback build leaves it to a fill here.
  gp::DeviceArray d_short;
  const size_t zero_bytes = 256 + ((sizeof(unsigned long long) * gp::onepass_state_words(n) + 255) & ~size_t(255));
  GP_TRY(d_short.alloc_async(zero_bytes, s));
  bool zeroed = false;
  GP_TRY(point_grid_create_impl(points_dev, n, cell_size > 0.0 ? cell_size : 0.25, structure, counters_dev, stream, true, false, &g, gp::fill_job(d_short.ptr, zero_bytes, 0u), &zeroed));
  const bool heavy_first = g->binned && g->structure != 6 && g->structure != 3 && !g->bin_levels.empty() && g->bin_levels[0]->bins.cell_of.ptr;
  // round 5: sparse neighbourhoods are handed to covariance_far_kernel (one wave per query); structure 7 = round 4's search (every query lane by lane) for the A/B
  const bool coop_far = heavy_first && g->structure == 0 && g->bin_levels.size() == 1 && k <= 10;
  int rc = GP_OK;
  if (rc == GP_OK) {
    if (!zeroed) (void)hipMemsetAsync(d_short.ptr, 0, zero_bytes, s);
    const gp::SearchView v = g->view();
    const int nq = g->binned ? g->num_binned : n;  // queries = the cell-sorted points; non-finite points are not among them
    if (nq < n) hipLaunchKernelGGL(gp::nonfinite_identity_kernel, dim3((n + 255) / 256), dim3(256), 0, s, points_dev, n, covs_dev, d_short.as<int>());
    const dim3 grid((nq + 127) / 128), block(128);
    gp::DeviceArray todo;  // [nq] positions + the count behind them
    gp::DeviceArray scan_buf;  // RowScanOut: kept [kTileKeep][nq] | bound [nq] | safe [nq]
    const int* d_todo = nullptr;
    if (nq > 0 && g->binned && g->structure == 3 && k <= 10) {
      // tiled pass over the occupied cell rows of the finest level; what it cannot settle is listed for the per-lane pass
      rc = todo.alloc_async(sizeof(int) * ((size_t)nq + 1), s);
      if (rc == GP_OK) rc = scan_buf.alloc_async(sizeof(int) * (size_t)(gp::kTileKeep + 2) * nq, s);
      if (rc == GP_OK) {
        (void)hipMemsetAsync(todo.as<int>() + nq, 0, sizeof(int), s);
        const gp::RowScanOut scan{scan_buf.as<int>(), reinterpret_cast<float*>(scan_buf.as<int>() + (size_t)gp::kTileKeep * nq),
                                  reinterpret_cast<float*>(scan_buf.as<int>() + (size_t)(gp::kTileKeep + 1) * nq), nq};
        hipLaunchKernelGGL(gp::covariance_rows_kernel, dim3(16u * (unsigned)g->bin_levels[0]->bins.num_occ_blocks), dim3(gp::kRowThreads), 0, s, v.bins[0],
                           (const int*)g->bin_levels[0]->bins.occ_blocks.as<int>(), scan, 0);
        hipLaunchKernelGGL(gp::covariance_settle_kernel<10>, grid, block, 0, s, v.bins[0].sorted, scan, points_dev, k, covs_dev, todo.as<int>(), todo.as<int>() + nq);
        d_todo = todo.as<int>();
      }
    }
    gp::DeviceArray heavy_before;
    if (nq > 0 && rc == GP_OK && heavy_first && !d_todo) {
      // heavy queries first (HeavyCellCount above): order[] = positions with own-cell population < k, then the rest; one scan over the cells + one scatter
      const gp::PointBins& bins = g->bin_levels[0]->bins;
      const int nc = bins.num_cells;
      rc = todo.alloc_async(sizeof(int) * ((size_t)nq + 2), s);
      if (rc == GP_OK) rc = heavy_before.alloc_async(sizeof(int) * (size_t)nc, s);
      if (rc == GP_OK) {
        int* d_heavy = todo.as<int>() + nq + 1;
        rc = gp::exclusive_scan_of(gp::HeavyCellCount{bins.cell_start.as<int>(), k}, heavy_before.as<int>(), nc, d_heavy, s,
                                   reinterpret_cast<unsigned long long*>(d_short.as<char>() + 256));
        if (rc == GP_OK) {
          hipLaunchKernelGGL(gp::heavy_first_order_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, (const int*)bins.cell_start.as<int>(), (const unsigned*)bins.cell_of.as<unsigned>(),
                             (const int*)heavy_before.as<int>(), (const int*)d_heavy, k, nq, todo.as<int>());
          d_todo = todo.as<int>();
        }
      }
    }
    gp::DeviceArray far;  // positions (cell-sorted array) of the queries left to the cooperative kernel + their k-th distances so far; the count is word 1 of the zeroed block
    int *d_far = nullptr, *d_far_count = nullptr;
    float* d_far_bound = nullptr;
    if (nq > 0 && rc == GP_OK && coop_far) {
      rc = far.alloc_async((sizeof(int) + sizeof(float)) * (size_t)nq, s);
      if (rc == GP_OK) d_far = far.as<int>(), d_far_bound = reinterpret_cast<float*>(far.as<int>() + nq), d_far_count = d_short.as<int>() + 1;
    }
    if (nq > 0 && rc == GP_OK) {
      constexpr int cov_waves = 4;      // registers capped at 128 for four waves per SIMD (uncapped, three waves: 1.41 vs 1.28 ms, round 2)
      constexpr bool cov_full = true;   // k = 10: the straight-line insertion of full lists (profiles/r03_c5_straightline.txt)
      const int* d_count = d_todo ? d_todo + nq : nullptr;
      gp::SideStream side;
      if (d_far && k == 10 && gp::SideStream::get(s, &side) == GP_OK) {
        // Round 5: TWO launches of the same kernel on two streams.  The heavy part of the order (own-cell population < k: the only queries that can turn out sparse)
        // runs on `s` with the deferral, covariance_far_kernel behind it; the rest runs beside it on a side stream.  The far kernel is a few hundred waves bound by
        // the latency of its round trips (150-200 us for 0.15 % of the queries): behind ONE launch it would be added to the call, here it runs under the other
        // launch's waves.  Both cover the whole order with their grids and leave by the counts on the device (the split is not known to the host).
        const int* d_heavy = d_todo + nq + 1;
        bool forked = hipEventRecord(side.fork, s) == hipSuccess && hipStreamWaitEvent(side.stream, side.fork, 0) == hipSuccess;
        hipLaunchKernelGGL((gp::covariance_kernel<10, 4, true>), grid, block, 0, s, v, points_dev, nq, k, covs_dev, d_short.as<int>(), d_todo, d_heavy, d_far, d_far_count, d_far_bound,
                           (const int*)nullptr);
        hipLaunchKernelGGL((gp::covariance_kernel<10, 4, true>), grid, block, 0, forked ? side.stream : s, v, points_dev, nq, k, covs_dev, d_short.as<int>(), d_todo, d_count,
                           (int*)nullptr, (int*)nullptr, (float*)nullptr, d_heavy);
        const unsigned far_wgs = (unsigned)std::min<long long>(((long long)nq + 3) / 4, 8192);  // one wave = four queries per workgroup
        hipLaunchKernelGGL(gp::covariance_far_kernel<10>, dim3(far_wgs), dim3(64), 0, s, v.bins[0], points_dev, k, covs_dev, d_short.as<int>(), (const int*)d_far, (const float*)d_far_bound,
                           (const int*)d_far_count);
        if (forked && (hipEventRecord(side.join, side.stream) != hipSuccess || hipStreamWaitEvent(s, side.join, 0) != hipSuccess)) {
          (void)hipStreamSynchronize(side.stream);  // (the join could not be queued: wait for the side stream here, the call is synchronous anyway)
        }
      } else {
        if (k == 10 && cov_waves == 4 && cov_full)  // (capped at 96 registers for five waves per SIMD: 41 spilled, 1.08 vs 1.07 ms -- no gain)
          hipLaunchKernelGGL((gp::covariance_kernel<10, 4, true>), grid, block, 0, s, v, points_dev, nq, k, covs_dev, d_short.as<int>(), d_todo, d_count, d_far, d_far_count, d_far_bound,
                             (const int*)nullptr);
        else if (k <= 10 && cov_waves == 4)
          hipLaunchKernelGGL((gp::covariance_kernel<10, 4>), grid, block, 0, s, v, points_dev, nq, k, covs_dev, d_short.as<int>(), d_todo, d_count, d_far, d_far_count, d_far_bound,
                             (const int*)nullptr);
        else if (k <= 10)
          hipLaunchKernelGGL((gp::covariance_kernel<10, 1>), grid, block, 0, s, v, points_dev, nq, k, covs_dev, d_short.as<int>(), d_todo, d_count, d_far, d_far_count, d_far_bound,
                             (const int*)nullptr);
        else
          hipLaunchKernelGGL(gp::covariance_kernel<32>, grid, block, 0, s, v, points_dev, nq, k, covs_dev, d_short.as<int>(), d_todo, d_count, (int*)nullptr, (int*)nullptr, (float*)nullptr,
                             (const int*)nullptr);
        if (d_far) {
          // a fixed grid (the count stays on the device): 4 waves per workgroup, 4 queries per wave, every group of 16 lanes takes queries w, w + groups, ... of the list
          const unsigned far_wgs = (unsigned)std::min<long long>(((long long)nq + 3) / 4, 8192);
          hipLaunchKernelGGL(gp::covariance_far_kernel<10>, dim3(far_wgs), dim3(64), 0, s, v.bins[0], points_dev, k, covs_dev, d_short.as<int>(), (const int*)d_far, (const float*)d_far_bound,
                             (const int*)d_far_count);
        }
      }
    }
    // the count of short queries comes back through a host-mapped word, behind a one-thread kernel whose flag the host polls (a D2H copy is a copy kernel + the
    // stream synchronisation's wake-up: ~10 us more)
    int h_short = 0;
    {
      gp::HostWords hw;
      int frc = gp::HostWords::get(&hw);
      if (frc == GP_OK) frc = hw.finish(s, d_short.as<int>(), 13);
      if (frc == GP_OK) {
        h_short = reinterpret_cast<volatile int*>(hw.host)[13];
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) frc = gp::hip_fail(e, "covariance_kernel", __FILE__, __LINE__);
      }
      if (frc != GP_OK && rc == GP_OK) rc = frc;
    }
    if (num_short) *num_short = h_short;
    if (h_short > 0) fprintf(stderr, "warning: fewer than k neighbors found for %d points\n", h_short);  // covariance_estimation.cpp:28
  }
  // the structure was built and searched on `s` only, and `s` has been synchronised: its arrays go back to the pool in stream order
  // (gp_point_grid_destroy has to assume searches on other streams and synchronises the device: 1.4 ms in a process with many streams)
  for (auto& lv : g->bin_levels) {
    for (gp::DeviceArray* a : {&lv->bins.blocks, &lv->bins.cell_start, &lv->bins.order, &lv->bins.cell_of, &lv->bins.cell_block, &lv->bins.occ_blocks, &lv->sorted, &lv->super})
      a->release_on(s);
  }
  delete g;
  return rc;
}

// ---- GICP factor ----------------------------------------------------------------------------------------------------

int gp_gicp_factor_create(const float* target_points_dev, const float* target_covs_dev, int n_target, const float* points_dev, const float* covs_dev, int n,
                          double max_correspondence_distance_sq, gp_stream_t stream, gp_gicp_factor_t** out) {
  return gp_gicp_factor_create_ex(target_points_dev, target_covs_dev, n_target, points_dev, covs_dev, n, max_correspondence_distance_sq, 0, nullptr, stream, out);
}

int gp_gicp_factor_create_ex(const float* target_points_dev, const float* target_covs_dev, int n_target, const float* points_dev, const float* covs_dev, int n,
                             double max_correspondence_distance_sq, int structure, unsigned long long* counters_dev, gp_stream_t stream, gp_gicp_factor_t** out) {
  if (!target_points_dev || !target_covs_dev || !points_dev || !covs_dev || n < 0 || n_target < 0 || !(max_correspondence_distance_sq > 0.0) || !out)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_gicp_factor_create: bad arguments");
  auto* f = new gp_gicp_factor;
  f->stream = (hipStream_t)stream;
  // finest cell = 1/4 of the correspondence radius (coarser levels x4, x16); the max-distance bound ends every search
  // cell = 1/4 of the correspondence radius: the fine shells 0 and 1 settle the well-matched points, and one block edge = the
  // radius, so the block walk behind them ends at the first block shell at the latest
  int rc = gp_point_grid_create_ex(target_points_dev, n_target, std::sqrt(max_correspondence_distance_sq) / 4.0, structure, counters_dev, stream, &f->grid);
  if (rc != GP_OK) {
    delete f;
    return rc;
  }
  f->desc.points = points_dev;
  f->desc.covs = covs_dev;
  f->desc.target_points = target_points_dev;
  f->desc.target_covs = target_covs_dev;
  f->desc.grid = f->grid->view();  // the max-distance bound terminates the search early (worst() starts at max_sq_dist)
  f->desc.n = n;
  f->desc.max_sq_dist = max_correspondence_distance_sq;
  f->num_tiles = (n + f->tile_points - 1) / f->tile_points;
  if ((rc = f->partials.alloc(sizeof(double) * gp::ACCG_STRIDE * (size_t)std::max(f->num_tiles, 1))) || (rc = f->d_poses.alloc(sizeof(double) * 32)) ||
      (rc = f->d_out.alloc(sizeof(gp_linearized6))) || (rc = f->h_out.ensure(sizeof(gp_linearized6)))) {
    gp_point_grid_destroy(f->grid);
    delete f;
    return rc;
  }
  GP_HIP(hipHostGetDevicePointer(&f->h_out_dev, f->h_out.ptr, 0));
  if ((rc = f->h_done.ensure(sizeof(unsigned long long))) != GP_OK) {
    gp_point_grid_destroy(f->grid);
    delete f;
    return rc;
  }
  memset(f->h_done.ptr, 0, f->h_done.bytes);
  GP_HIP(hipHostGetDevicePointer(&f->h_done_dev, f->h_done.ptr, 0));
  *out = f;
  return GP_OK;
}

int gp_gicp_factor_destroy(gp_gicp_factor_t* f) {
  if (!f) return GP_OK;
  (void)hipStreamSynchronize(f->stream);
  gp_point_grid_destroy(f->grid);
  delete f;
  return GP_OK;
}

int gp_gicp_factor_linearize(gp_gicp_factor_t* f, const double pose[16], gp_linearized6* out_host) {
  if (!f || !pose || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_gicp_factor_linearize: null");
  gp::GicpPoses P;
  memcpy(P.lin, pose, sizeof(double) * 16);
  memcpy(P.eval, pose, sizeof(double) * 16);
  // the 29-sum kernel + adjoint finalize is exact only for an orthonormal 3x3 block; any other pose (e.g. built from 6-digit
  // quaternions, src/test/test_matching_cost_factors.cpp:50-55) takes the 92-sum path with the explicit J_s, like the VGICP factor
  const bool rigid = gp::pose_is_rigid(pose);
  if (f->num_tiles > 0) {
    const int* corr = gicp_correspond(f, P, false) ? f->corr.as<int>() : nullptr;
    if (rigid && corr)
      hipLaunchKernelGGL((gp::gicp_tile_kernel<gp::MODE_LIN, true>), dim3(f->num_tiles), dim3(256), 0, f->stream, f->desc, P, f->tile_points, f->partials.as<double>(), corr);
    else if (rigid)
      hipLaunchKernelGGL((gp::gicp_tile_kernel<gp::MODE_LIN, false>), dim3(f->num_tiles), dim3(256), 0, f->stream, f->desc, P, f->tile_points, f->partials.as<double>(), corr);
    else if (corr)
      hipLaunchKernelGGL((gp::gicp_tile_kernel<gp::MODE_LIN_GENERAL, true>), dim3(f->num_tiles), dim3(256), 0, f->stream, f->desc, P, f->tile_points, f->partials.as<double>(), corr);
    else
      hipLaunchKernelGGL((gp::gicp_tile_kernel<gp::MODE_LIN_GENERAL, false>), dim3(f->num_tiles), dim3(256), 0, f->stream, f->desc, P, f->tile_points, f->partials.as<double>(), corr);
    GP_HIP(hipGetLastError());
  }
  const gp::DoneFlags done{static_cast<unsigned long long*>(f->h_done_dev), ++f->seq};
  GP_TRY(gp::launch_finalize_single(f->stream, nullptr, pose, f->partials.as<double>(), f->num_tiles, reinterpret_cast<gp_linearized6*>(f->h_out_dev), !rigid, done));
  GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(f->h_done.ptr), 1, done.seq, f->stream, 100 + (long)f->num_tiles * (long)f->tile_points / 1000));  // spin budget ~4x the kernel (0.2 ns per point)
  memcpy(out_host, f->h_out.ptr, sizeof(gp_linearized6));
  return GP_OK;
}

int gp_gicp_factor_compute_error(gp_gicp_factor_t* f, const double pose_lin[16], const double pose_eval[16], double* out_host) {
  if (!f || !pose_lin || !pose_eval || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_gicp_factor_compute_error: null");
  gp::GicpPoses P;
  memcpy(P.lin, pose_lin, sizeof(double) * 16);
  memcpy(P.eval, pose_eval, sizeof(double) * 16);
  if (f->num_tiles > 0) {
    const int* corr = gicp_correspond(f, P, true) ? f->corr.as<int>() : nullptr;
    if (corr)
      hipLaunchKernelGGL((gp::gicp_tile_kernel<gp::MODE_ERR, true>), dim3(f->num_tiles), dim3(256), 0, f->stream, f->desc, P, f->tile_points, f->partials.as<double>(), corr);
    else
      hipLaunchKernelGGL((gp::gicp_tile_kernel<gp::MODE_ERR, false>), dim3(f->num_tiles), dim3(256), 0, f->stream, f->desc, P, f->tile_points, f->partials.as<double>(), corr);
    GP_HIP(hipGetLastError());
  }
  const gp::DoneFlags done{static_cast<unsigned long long*>(f->h_done_dev), ++f->seq};
  GP_TRY(gp::launch_finalize_error_single(f->stream, f->partials.as<double>(), f->num_tiles, reinterpret_cast<double*>(f->h_out_dev), done));
  GP_TRY(gp::wait_done(static_cast<const unsigned long long*>(f->h_done.ptr), 1, done.seq, f->stream, 100 + (long)f->num_tiles * (long)f->tile_points / 1000));  // spin budget ~4x the kernel (0.2 ns per point)
  memcpy(out_host, f->h_out.ptr, sizeof(double));
  return GP_OK;
}

}  // extern "C"
