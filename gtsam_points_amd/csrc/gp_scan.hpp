// gp_scan.hpp -- exclusive prefix sum of a strided int array on the device: one launch (decoupled look-back, round 4); the three-launch form (per-block scan,
// scan of the block sums by one workgroup, add) remains for arrays beyond 2^31 - 4096 entries.  Used by the occupancy-grid build (gp_voxelmap.hip); the arrays are a few
// 10^4 .. 10^7 entries long and the scan runs once per map build, so it is written for clarity, not tuned.
#pragma once

#include "gp_host.hpp"

namespace gp {

constexpr int kScanThreads = 1024;

// out[i * out_stride] = sum_{j < i} in[j * in_stride]; block_sums[b] = total of block b
template <int UNUSED = 0>
__global__ void __launch_bounds__(kScanThreads) strided_scan_block_kernel(const int* __restrict__ in, int in_stride, int* __restrict__ out, int out_stride,
                                                                          int* __restrict__ block_sums, long long m) {
  __shared__ int lds[kScanThreads];
  const long long i = (long long)blockIdx.x * kScanThreads + threadIdx.x;
  const int v = i < m ? in[i * in_stride] : 0;
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < kScanThreads; off <<= 1) {
    const int t = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
    __syncthreads();
    lds[threadIdx.x] += t;
    __syncthreads();
  }
  if (i < m) out[i * out_stride] = lds[threadIdx.x] - v;
  if (threadIdx.x == kScanThreads - 1) block_sums[blockIdx.x] = lds[threadIdx.x];
}

// in-place exclusive scan of block_sums[0..nb) by ONE workgroup; *total = grand total
template <int UNUSED = 0>
__global__ void __launch_bounds__(kScanThreads) strided_scan_sums_kernel(int* __restrict__ block_sums, int nb, int* __restrict__ total) {
  __shared__ int lds[kScanThreads];
  int carry = 0;
  for (int base = 0; base < nb; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {
      const int t = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
      __syncthreads();
      lds[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) block_sums[i] = carry + lds[threadIdx.x] - v;
    const int last = lds[kScanThreads - 1];
    __syncthreads();
    carry += last;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

template <int UNUSED = 0>
__global__ void __launch_bounds__(kScanThreads) strided_scan_add_kernel(int* __restrict__ out, int out_stride, const int* __restrict__ block_sums, long long m) {
  const long long i = (long long)blockIdx.x * kScanThreads + threadIdx.x;
  if (i < m) out[i * out_stride] += block_sums[blockIdx.x];
}

// ---- one-pass scan (round 4): decoupled look-back, ONE launch instead of three -------------------------------------------------------------------------------
// A map build runs five scans and a k-NN structure build as many; each was three launches (block scan, scan of the block sums by one workgroup, add), 14-19 us of
// kernels and two launch gaps apiece (profiles/r03_bench_kernel_stats.csv: 72 us of a 370 us build).  Here a workgroup scans a tile of 4096 elements, publishes the
// tile's total, and finds its exclusive prefix by looking back over its predecessors' words: total-only words are added up 64 at a time by one wave until a word that
// already carries an inclusive prefix ends the walk.  Tiles are handed out by a ticket counter, so a tile's predecessors have always been started.
// (the words are self-contained -- flag and value in one 64-bit access -- so relaxed agent-scope accesses suffice)
// state: onepass_state_words(m) 64-bit words, ZERO when the kernel starts: word 0 = ticket counter, word 1 + t = (flag << 32 | value) of tile t, flag 1 = total, 2 = inclusive prefix.
constexpr int kOnePassTile = 16384;  // 1024 threads x 16: a 2 M-entry scan is 122 tiles -- two look-back windows at most (4096-element tiles: 488 tiles, 19 us per scan)
constexpr int kOnePassPerThread = kOnePassTile / 1024;
inline size_t onepass_state_words(long long m) { return 2 + (size_t)((m + kOnePassTile - 1) / kOnePassTile); }

// the scan's input as a function of the position (a flag computed from other arrays need not be stored first), or an array
struct ScanArrayInput {
  const int* in;
  int stride;
  __device__ __forceinline__ int operator()(long long i) const { return in[i * stride]; }
};
template <typename Input>
__global__ void __launch_bounds__(1024) scan_onepass_kernel(const Input input, const int* __restrict__ in, int in_stride, int* __restrict__ out, int out_stride, long long m,
                                                            unsigned long long* __restrict__ state, int* __restrict__ total) {
  __shared__ int wave_sum[16];
  __shared__ int tile_id, tile_prefix;
  if (threadIdx.x == 0) tile_id = (int)atomicAdd(state, 1ull);
  __syncthreads();
  const int tile = tile_id;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)tile * kOnePassTile + (long long)threadIdx.x * kOnePassPerThread;
  int v[kOnePassPerThread];
  int mine = 0;
  // a thread owns 16 consecutive elements (64 B): as four 16-byte accesses when the array is dense -- sixteen 4-byte accesses at a 64-byte lane stride cost a
  // cache-line operation per lane and instruction, 19 us for a 2 M-entry scan
  const bool vec_out = out_stride == 1 && base + kOnePassPerThread <= m && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const bool vec = in != nullptr && in_stride == 1 && vec_out && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  if (vec) {
    const int4* p = reinterpret_cast<const int4*>(in + base);
#pragma unroll
    for (int q = 0; q < kOnePassPerThread / 4; q++) {
      const int4 x = p[q];
      v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
    }
#pragma unroll
    for (int k = 0; k < kOnePassPerThread; k++) mine += v[k];
  } else {
#pragma unroll
    for (int k = 0; k < kOnePassPerThread; k++) {
      v[k] = base + k < m ? input(base + k) : 0;
      mine += v[k];
    }
  }
  int incl = mine;  // inclusive scan over the wave's 64 threads
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  int wave_excl = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    if (w < wave) wave_excl += wave_sum[w];
    tile_total += wave_sum[w];
  }
  unsigned long long* words = state + 1;
  if (wave == 0) {
    if (lane == 0) __hip_atomic_store(words + tile, tile == 0 ? (2ull << 32 | (unsigned)tile_total) : (1ull << 32 | (unsigned)tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int prefix = 0;
    // look back: lanes inspect the 64 predecessors tile - 1 - lane ...; the walk ends at the nearest word that carries an inclusive prefix
    for (int hi = tile - 1; hi >= 0;) {
      const int t = hi - lane;
      unsigned long long w = t >= 0 ? __hip_atomic_load(words + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 32);
      while (__builtin_amdgcn_ballot_w64((w >> 32) == 0ull) != 0ull) {  // some predecessor in the window has not published yet
        __builtin_amdgcn_s_sleep(1);
        w = t >= 0 ? __hip_atomic_load(words + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 32);
      }
      const unsigned long long incl_mask = __builtin_amdgcn_ballot_w64((w >> 32) == 2ull);
      const int first_incl = incl_mask ? __ffsll((long long)incl_mask) - 1 : 64;  // nearest predecessor with an inclusive prefix (lane index = distance - 1)
      int part = lane <= first_incl ? (int)(unsigned)w : 0;                          // totals up to it, and its prefix
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      prefix += part;
      if (incl_mask) break;
      hi -= 64;
    }
    if (lane == 0) {
      tile_prefix = prefix;
      if (tile > 0) __hip_atomic_store(words + tile, 2ull << 32 | (unsigned)(prefix + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (total && (long long)(tile + 1) * kOnePassTile >= m) *total = prefix + tile_total;
    }
  }
  __syncthreads();
  int run = tile_prefix + wave_excl + incl - mine;
  if (vec_out) {
    int4* p = reinterpret_cast<int4*>(out + base);
#pragma unroll
    for (int q = 0; q < kOnePassPerThread / 4; q++) {
      int4 x;
      x.x = run, run += v[4 * q];
      x.y = run, run += v[4 * q + 1];
      x.z = run, run += v[4 * q + 2];
      x.w = run, run += v[4 * q + 3];
      p[q] = x;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kOnePassPerThread; k++) {
      if (base + k < m) out[(base + k) * out_stride] = run;
      run += v[k];
    }
  }
}

// scratch: at least ceil(m / 1024) + 1 ints + (one-pass form) 2 * onepass_state_words(m) + 2 ints.  `in` and `out` may alias only when they are the same array with the same
// stride.  The grand total is left at scratch[ceil(m / 1024)] (both forms).
inline size_t scan_scratch_ints(long long m) { return (size_t)(m / kScanThreads) + 8 + 2 * onepass_state_words(m) + 2; }

// exclusive scan of input(0 .. m - 1) into out[] (dense), the grand total into *total: one launch.  zeroed_state as below (required).  m <= 2^31 - 16384.
template <typename Input>
inline int exclusive_scan_of(const Input& input, int* out, long long m, int* total, hipStream_t s, unsigned long long* zeroed_state) {
  if (m <= 0) return GP_OK;
  hipLaunchKernelGGL(scan_onepass_kernel<Input>, dim3((unsigned)((m + kOnePassTile - 1) / kOnePassTile)), dim3(1024), 0, s, input, (const int*)nullptr, 1, out, 1, m, zeroed_state, total);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

// zeroed_state: onepass_state_words(m) words the CALLER has zeroed on `s` (a build that runs several scans zeroes all their states with one fill), or null
inline int exclusive_scan_strided(const int* in, int in_stride, int* out, int out_stride, long long m, int* scratch, hipStream_t s, unsigned long long* zeroed_state = nullptr) {
  if (m <= 0) return GP_OK;
  const int nb = (int)((m + kScanThreads - 1) / kScanThreads);
  if (m <= (1ll << 31) - kOnePassTile) {
    // the state words live behind the total's slot, 8-byte aligned
    unsigned long long* state = zeroed_state ? zeroed_state : reinterpret_cast<unsigned long long*>(reinterpret_cast<uintptr_t>(scratch + nb + 2 + 1) & ~uintptr_t(7));
    if (!zeroed_state) GP_HIP(hipMemsetAsync(state, 0, sizeof(unsigned long long) * onepass_state_words(m), s));
    hipLaunchKernelGGL(scan_onepass_kernel<ScanArrayInput>, dim3((unsigned)((m + kOnePassTile - 1) / kOnePassTile)), dim3(1024), 0, s, ScanArrayInput{in, in_stride}, in,
                       in_stride, out, out_stride, m, state, scratch + nb);
    GP_HIP(hipGetLastError());
    return GP_OK;
  }
  hipLaunchKernelGGL(strided_scan_block_kernel<0>, dim3(nb), dim3(kScanThreads), 0, s, in, in_stride, out, out_stride, scratch, m);
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(strided_scan_sums_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, scratch, nb, scratch + nb);
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(strided_scan_add_kernel<0>, dim3(nb), dim3(kScanThreads), 0, s, out, out_stride, (const int*)scratch, m);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

}  // namespace gp
