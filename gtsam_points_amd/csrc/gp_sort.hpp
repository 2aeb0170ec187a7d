// gp_sort.hpp -- stable LSD radix sort of (key, value) pairs of 32-bit integers on the device, 8 bits per pass.
//
// Used to bin points by voxel / cell ordinal (gp_binning.hpp): the sort is STABLE, so points of one cell end up in ascending
// point-index order -- which makes everything derived from the bins (voxel statistics, k-NN tie order) independent of the
// order atomics happen to land in, i.e. bit-reproducible from run to run.  No atomics on global memory anywhere in here.
//
// One pass = histogram kernel (256 bins per 4096-element tile, LDS) -> exclusive scan of the [bin][tile] table (gp_scan.hpp)
// -> scatter kernel.  Inside a tile every wave owns 1024 consecutive elements and ranks them in 16 rounds of 64: the lanes
// holding the same digit find each other with 8 ballots, popcount gives the rank inside the round, a per-wave running count in LDS
// the rank inside the wave's range; ranges of the four waves and the tile's global offsets are added at the end.
#pragma once

#include "gp_scan.hpp"

namespace gp {

constexpr int kSortTile = 4096;  // elements per workgroup (256 threads x 16)

template <int UNUSED = 0>
__global__ void __launch_bounds__(256) radix_hist_kernel(const unsigned* __restrict__ keys, int n, int shift, int* __restrict__ hist, int num_tiles) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const size_t i = base + (size_t)r * 256 + threadIdx.x;
    if (i < (size_t)n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = h[threadIdx.x];  // [bin][tile]: the scan of this table is the global offset
}

// vals_in == nullptr: the values are the element indices (first pass of an argsort)
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) radix_scatter_kernel(const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, int n, int shift,
                                                            const int* __restrict__ offsets /*scanned [bin][tile]*/, int num_tiles, unsigned* __restrict__ keys_out,
                                                            int* __restrict__ vals_out) {
  __shared__ int wave_count[4][256];
  __shared__ int wave_base[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 4 * 256; k += 256) (&wave_count[0][0])[k] = 0;
  __syncthreads();
  const size_t sub = (size_t)blockIdx.x * kSortTile + (size_t)wave * 1024;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned key[16];
  int val[16], rank[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const size_t i = sub + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n;
    key[r] = valid ? keys_in[i] : 0u;
    val[r] = valid ? (vals_in ? vals_in[i] : (int)i) : 0;
    const unsigned d = (key[r] >> shift) & 255u;
    // lanes of this round that hold the same digit (invalid lanes match nobody)
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    int before = 0;
    if (valid) {
      before = wave_count[wave][d];  // same-wave LDS traffic is ordered: every peer reads before the leader's update below
      rank[r] = before + __popcll(peers & below);
      if ((peers & below) == 0ull) wave_count[wave][d] = before + __popcll(peers);
    } else {
      rank[r] = -1;
    }
  }
  __syncthreads();
  {
    // digit d: exclusive prefix of the four waves' counts on top of the tile's global offset
    const int d = threadIdx.x;
    int run = offsets[(size_t)d * num_tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      wave_base[w][d] = run;
      run += wave_count[w][d];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (rank[r] >= 0) {
      const int pos = wave_base[wave][(key[r] >> shift) & 255u] + rank[r];
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
  }
}

// scratch ints needed by radix_sort_pairs for n elements
inline size_t radix_sort_scratch_ints(int n) {
  const size_t tiles = ((size_t)n + kSortTile - 1) / kSortTile;
  return 256 * tiles + scan_scratch_ints(256ll * (long long)tiles);
}

// Sorts n pairs by the low `key_bits` bits of the key, stable.  (keys_a, vals_a) hold the input -- with vals_iota the values are
// taken to be 0..n-1 and vals_a is only storage; the passes ping-pong between the a and b buffers; *result_in_b tells where the
// sorted pairs ended up.  n <= 2^30.
// zeroed_states: radix_sort_state_words(n, key_bits) words the caller has zeroed on `s` (one look-back state per pass), or null
inline size_t radix_sort_state_words(int n, int key_bits) {
  const size_t tiles = ((size_t)n + kSortTile - 1) / kSortTile;
  return (size_t)((key_bits + 7) / 8) * onepass_state_words(256ll * (long long)tiles);
}
inline int radix_sort_pairs(unsigned* keys_a, int* vals_a, unsigned* keys_b, int* vals_b, int n, int key_bits, bool vals_iota, int* scratch, hipStream_t s,
                            bool* result_in_b, unsigned long long* zeroed_states = nullptr) {
  *result_in_b = false;
  if (n <= 0) return GP_OK;
  const int tiles = (n + kSortTile - 1) / kSortTile;
  int* hist = scratch;
  int* scan_scratch = scratch + 256 * (size_t)tiles;
  bool in_a = true, first = true;
  for (int shift = 0; shift < key_bits; shift += 8) {
    const unsigned* kin = in_a ? keys_a : keys_b;
    const int* vin = (first && vals_iota) ? nullptr : (in_a ? vals_a : vals_b);
    unsigned* kout = in_a ? keys_b : keys_a;
    int* vout = in_a ? vals_b : vals_a;
    hipLaunchKernelGGL(radix_hist_kernel<0>, dim3(tiles), dim3(256), 0, s, kin, n, shift, hist, tiles);
    GP_HIP(hipGetLastError());
    GP_TRY(exclusive_scan_strided(hist, 1, hist, 1, 256ll * tiles, scan_scratch, s, zeroed_states ? zeroed_states + (size_t)(shift / 8) * onepass_state_words(256ll * tiles) : nullptr));
    hipLaunchKernelGGL(radix_scatter_kernel<0>, dim3(tiles), dim3(256), 0, s, kin, vin, n, shift, (const int*)hist, tiles, kout, vout);
    GP_HIP(hipGetLastError());
    in_a = !in_a;
    first = false;
  }
  *result_in_b = !in_a;
  return GP_OK;
}

}  // namespace gp
