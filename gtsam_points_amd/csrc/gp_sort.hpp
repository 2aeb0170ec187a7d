// gp_sort.hpp -- stable LSD radix sort of (key, value) pairs of 32-bit integers on the device, 8 bits per pass.
//
// Used to bin points by voxel / cell ordinal (gp_binning.hpp): the sort is STABLE, so points of one cell end up in ascending
// point-index order -- which makes everything derived from the bins (voxel statistics, k-NN tie order) independent of the
// order atomics happen to land in, i.e. bit-reproducible from run to run.  The only atomics on global memory are integer counters
// (digit histograms, tile tickets): their sums do not depend on the order they land in.
//
// Round 4: ONE histogram kernel for all passes (a digit's count depends on the multiset of keys, not on their order) and ONE kernel per pass: a workgroup ranks its
// tile of 4096 elements, publishes the tile's 256 digit counts and finds its offsets from its predecessors' published counts (two levels, below)
// -- instead of histogram + scan of the [digit][tile] table + scatter per pass (29 us per pass and 2 M keys, three launches; a map build sorts in three
// passes, a k-NN grid in four: profiles/r04_map_build_kernel_stats.csv).
// Inside a tile every wave owns 1024 consecutive elements and ranks them in 16 rounds of 64: the lanes holding the same digit find each other with 8 ballots,
// popcount gives the rank inside the round, a per-wave running count in LDS the rank inside the wave's range; ranges of the four waves, the tile's offset among the
// tiles and the digit's offset among the digits are added at the end.
#pragma once

#include "gp_scan.hpp"

namespace gp {

// measurement build only (scripts/probe/sort_probe.hip): phase stamps of every tile on the 100 MHz clock
#ifdef GP_SORT_TRACE
__device__ unsigned long long* g_sort_trace = nullptr;
#define GP_SORT_STAMP(tile, k)                                                                  \
  do {                                                                                          \
    if (threadIdx.x == 0 && g_sort_trace) g_sort_trace[(size_t)(tile) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define GP_SORT_STAMP(tile, k) \
  do {                         \
  } while (0)
#endif

constexpr int kSortTile = 4096;  // elements per workgroup (256 threads x 16)
constexpr int kSortMaxPasses = 4;
constexpr int kSortGroup = 32;  // tiles per group of the two-level offsets (below)
// Atomics of many workgroups on ONE address are served one after the other at the device's coherence point, ~33 ns apiece on MI355X: a ticket counter drawn by
// 488 workgroups, or 488 histogram flushes into the same 256 words, stretch a 3 us kernel to 16-18 us (profiles/r04_build_pmc.txt: wave lifetimes of 3.6 us in
// kernels 16 us long).  So the counters come in classes -- a workgroup uses the words of class blockIdx % classes -- and whoever needs the sum adds the classes up.
constexpr int kSortTicketClasses = 32;
constexpr int kSortFaultWord = 64;      // word of a pass's ticket line ([0 .. 31] tickets) that a tile whose wait expired sets to 1
constexpr int kSortSpinBound = 200000;  // polls (>= ~1 us each: a sleep + a round trip to the coherence point) before a tile gives up: ~0.2 - 0.5 s
// radix_onesweep_kernel's `ticket_classes` argument: kSortTicketClasses (fast path), 1 (deadlock-free), or a NEGATIVE class count = test hook: tile 0 raises the
// fault word as if its wait had expired (tests/test_sort_gpu.py drives the fallback with it); beyond -kSortCorruptHook (the class count is the remainder) tile 0 ALSO
// leaves what a real expired wait leaves -- keys that are not the input's (all-ones patterns far outside any cell range) in its part of the output -- so that the
// consumers of a voided sort are tested against real garbage, not only against the fault word (gp_binning.hip indexed its block grid with such keys)
constexpr int kSortCorruptHook = 1000;
constexpr int kSortHistClasses = 16;

// LDS histograms of all passes' digits for a key the caller has in a register (the kernel that PRODUCES the keys counts them: no pass over the keys for it)
struct SortHistLds {
  unsigned h[kSortMaxPasses][256];
};
__device__ __forceinline__ void sort_hist_clear(SortHistLds& l) {  // 256-thread workgroups; barrier behind it is the caller's
  for (int p = 0; p < kSortMaxPasses; p++) l.h[p][threadIdx.x] = 0;
}
// (counting a wave's equal digits by their first lane alone -- ballot + readlane, up to three groups, the rest by atomics -- was measured against these plain LDS
// atomics: a pass's load + count phase 1.0 -> 6.0 us, the key kernel 19 -> 26 us per 2 M points: the LDS unit serves same-address lanes faster than that loop)
__device__ __forceinline__ void sort_hist_count(SortHistLds& l, unsigned key, int passes) {
#pragma unroll
  for (int p = 0; p < kSortMaxPasses; p++)
    if (p < passes) atomicAdd(&l.h[p][(key >> (8 * p)) & 255u], 1u);
}
__device__ __forceinline__ void sort_hist_flush(SortHistLds& l, int passes, unsigned* __restrict__ hist /*[class][pass][256]*/) {  // behind a barrier
  unsigned* mine = hist + (size_t)(blockIdx.x % kSortHistClasses) * kSortMaxPasses * 256;
#pragma unroll
  for (int p = 0; p < kSortMaxPasses; p++)
    if (p < passes && l.h[p][threadIdx.x]) atomicAdd(mine + p * 256 + threadIdx.x, l.h[p][threadIdx.x]);
}
// A tile index for this workgroup without one hot counter: class c = blockIdx % classes draws from its own counter, tile = ticket * classes + c (a class's share of
// the grid is exactly the tiles of that form, so the indices are a permutation of the grid).  tickets: classes words, zeroed.
// What a tile may wait for are tiles with SMALLER indices.  One counter for all would guarantee that those have been drawn by workgroups that are running or done,
// whatever order the hardware starts workgroups in; the classes guarantee it within a class only.  Across classes it rests on the dispatcher starting workgroups in
// blockIdx order (as it does): the workgroups started so far then are a prefix of the grid, every class has handed out the same number of tickets (+- 1), and the
// drawn tiles are a prefix too -- also when the grid exceeds what is resident at once (tests/test_sort_gpu.py sorts 1026 tiles against ~768 resident workgroups).
// The price of the hot counter was 16 us per kernel (488 draws at ~33 ns apiece).
// HIP does not promise that start order (CU masks, partition modes, other streams' kernels holding the CUs, a later ROCm: ADVICE r04), so the classes are a fast
// path with a way out, not an assumption: every wait below is BOUNDED (kSortSpinBound polls); a tile whose wait expires raises the pass's fault word and stops
// waiting (everybody has published before waiting, so every workgroup still ends), the host sees the word behind the sort (radix_sort_fault) and runs the sort
// again with ONE class -- the deadlock-free form above, whatever the start order.
template <typename Word>
__device__ __forceinline__ int draw_tile(Word* tickets, int classes) {
  const int c = (int)(blockIdx.x % (unsigned)classes);
  return (int)atomicAdd(tickets + c, (Word)1) * classes + c;
}

// hist[class][pass][digit] += occurrences (hist zeroed by the caller): for callers whose keys come from elsewhere
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) radix_hist_all_kernel(const unsigned* __restrict__ keys, int n, int passes, unsigned* __restrict__ hist) {
  __shared__ SortHistLds l;
  sort_hist_clear(l);
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortTile / 256; r++) {
    const size_t i = base + (size_t)r * 256 + threadIdx.x;
    if (i < (size_t)n) sort_hist_count(l, keys[i], passes);
  }
  __syncthreads();
  sort_hist_flush(l, passes, hist);
}

// One pass.  Offsets without a chain (a look-back over 488 tiles costs ~10 us per pass: every tile finishes its ranking at the same moment, and the inclusive
// prefixes then travel tile by tile): every tile publishes its 256 digit counts (word = 1 << 31 | count) AND adds them to its group's words (32 tiles per group:
// word += 1 << 26 | count, so the top bits count the contributors; a group's 32 x 4096 elements fit 18 bits).  A tile's offset for a digit = the complete groups in
// front + the tiles in front inside its own group: <= tiles / 32 + 31 independent loads, all of them published before anybody's ranking is done.
// state (32-bit words, ZERO at the start): [0 .. 31] ticket counters, [256 ..) group words [group][256], behind them tile words [tile][256].
// vals_in == nullptr: the values are the element indices.
inline size_t radix_sort_groups(int n) { return (((size_t)n + kSortTile - 1) / kSortTile + kSortGroup - 1) / kSortGroup; }
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) radix_onesweep_kernel(const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, int n, int shift,
                                                             const unsigned* __restrict__ digit_hist /*[class][.][256], at this pass*/, unsigned* __restrict__ state, int num_groups,
                                                             unsigned* __restrict__ keys_out, int* __restrict__ vals_out, int ticket_classes) {
  __shared__ int wave_count[4][256];   // elements of the digit in the wave's range; later: where the wave's elements of the digit start inside the tile
  __shared__ unsigned tile_count[256];
  __shared__ int out_delta[256];       // global position of the digit's first element of this tile - its position inside the tile
  __shared__ unsigned scan_sums[4];
  __shared__ unsigned skeys[kSortTile];
  __shared__ int svals[kSortTile];
  __shared__ int tile_id;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    tile_id = draw_tile(state, ticket_classes < 0 ? (-ticket_classes) % kSortCorruptHook : ticket_classes);  // a tile's predecessors have been started (one class: always; more: see draw_tile)
    if (ticket_classes < 0 && tile_id == 0) __hip_atomic_store(state + kSortFaultWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // test hook
  }
  for (int k = threadIdx.x; k < 4 * 256; k += 256) (&wave_count[0][0])[k] = 0;
  tile_count[threadIdx.x] = 0;
  __syncthreads();
  const int tile = tile_id;
  GP_SORT_STAMP(tile, 0);  // ticket drawn
  unsigned* group_words = state + 256;
  unsigned* tile_words = group_words + (size_t)num_groups * 256;
  const size_t sub = (size_t)tile * kSortTile + (size_t)wave * 1024;
  unsigned key[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const size_t i = sub + (size_t)r * 64 + lane;
    key[r] = keys_in[i < (size_t)n ? i : (size_t)n - 1];  // (unconditional: the sixteen loads are in flight together)
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const size_t i = sub + (size_t)r * 64 + lane;
    if (i < (size_t)n) atomicAdd(&tile_count[(key[r] >> shift) & 255u], 1u);
  }
  __syncthreads();
  GP_SORT_STAMP(tile, 1);  // keys loaded and counted
  // the tile's counts go out before the ranking: by the time the ranks are known, everybody's are there
  const unsigned my_count = tile_count[threadIdx.x];
  const int group = tile / kSortGroup, in_group = tile % kSortGroup;
  __hip_atomic_store(tile_words + (size_t)tile * 256 + threadIdx.x, 1u << 31 | my_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(group_words + (size_t)group * 256 + threadIdx.x, 1u << 26 | my_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // what the offsets need from the others is asked for NOW and used behind the ranking (each of these is a ~2.5 us round trip to the device's coherence point; one
  // behind the other they were 10 us of a 20 us kernel): the classes' histograms of this pass, and the first 32 entries of (b) below -- the complete groups in
  // front, then the tiles in front inside the own group; they published at about the time this tile did, so most answers are final and the rest are asked again
  constexpr int kEarly = 32;
  const int entries = group + in_group;
  auto entry_word = [&](int e) -> const unsigned* {
    return e < group ? group_words + (size_t)e * 256 + threadIdx.x : tile_words + ((size_t)group * kSortGroup + (e - group)) * 256 + threadIdx.x;
  };
  unsigned hpart[kSortHistClasses], early[kEarly];
#pragma unroll
  for (int c = 0; c < kSortHistClasses; c++) hpart[c] = digit_hist[(size_t)c * kSortMaxPasses * 256 + threadIdx.x];
#pragma unroll
  for (int q = 0; q < kEarly; q++) early[q] = q < entries ? __hip_atomic_load(entry_word(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const unsigned long long below = (1ull << lane) - 1ull;
  int val[16], rank[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const size_t i = sub + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n;
    val[r] = vals_in ? vals_in[valid ? i : (size_t)n - 1] : (int)i;
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
  GP_SORT_STAMP(tile, 2);  // this wave's ranking done
  {
    // digit d = threadIdx.x.  (a) elements with a smaller digit in the whole array / in this tile: exclusive scans over the digits, both in one 64-bit scan
    unsigned hcount = 0;
#pragma unroll
    for (int c = 0; c < kSortHistClasses; c++) hcount += hpart[c];
    const unsigned long long both = (unsigned long long)hcount << 32 | my_count;
    unsigned long long incl = both;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) {
      scan_sums[wave] = (unsigned)(incl >> 32);
      out_delta[wave] = (int)(unsigned)incl;  // (borrowed until the barrier below: the waves' tile-local totals)
    }
    __syncthreads();  // (also: every wave's ranking is done, wave_count is final)
    GP_SORT_STAMP(tile, 3);  // everybody's ranking done
    unsigned digit_base = (unsigned)((incl - both) >> 32);
    int local_start = (int)(unsigned)(incl - both);
    for (int w = 0; w < wave; w++) {
      digit_base += scan_sums[w];
      local_start += out_delta[w];
    }
    // (b) elements with this digit in the tiles in front
    unsigned prefix = 0;
    int polls = 0;  // (per thread, over all its waits)
    auto take = [&](int e, unsigned w) {
      const unsigned* src = entry_word(e);
      if (e < group) {
        while ((w >> 26) != (unsigned)kSortGroup) {  // a tile of that group has not added its counts yet
          if (++polls > kSortSpinBound) {            // ... and may never: its workgroup cannot start while we hold the CU (see draw_tile). Give up, say so.
            __hip_atomic_store(state + kSortFaultWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        prefix += w & 0x3ffffffu;
      } else {
        while ((w >> 31) == 0u) {
          if (++polls > kSortSpinBound) {
            __hip_atomic_store(state + kSortFaultWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        prefix += w & 0x7fffffffu;
      }
    };
#pragma unroll
    for (int q = 0; q < kEarly; q++)
      if (q < entries) take(q, early[q]);
    for (int e0 = kEarly; e0 < entries; e0 += 16) {  // (more than 1056 tiles in front: sixteen independent loads at a time)
      unsigned w[16];
#pragma unroll
      for (int q = 0; q < 16; q++) w[q] = e0 + q < entries ? __hip_atomic_load(entry_word(e0 + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
      for (int q = 0; q < 16; q++)
        if (e0 + q < entries) take(e0 + q, w[q]);
    }
    GP_SORT_STAMP(tile, 4);  // this wave's offsets known
    __syncthreads();  // (out_delta's borrowed entries have been read)
    GP_SORT_STAMP(tile, 5);
    out_delta[threadIdx.x] = (int)(digit_base + prefix) - local_start;
    int run = local_start;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int c = wave_count[w][threadIdx.x];
      wave_count[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  // through LDS in tile-sorted order, so that neighbouring lanes store to neighbouring addresses (a digit's elements of this tile are one run: 64 B on average;
  // straight from the registers every lane of a store would touch its own cache line)
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (rank[r] >= 0) {
      const int pos = wave_count[wave][(key[r] >> shift) & 255u] + rank[r];
      skeys[pos] = key[r];
      svals[pos] = val[r];
    }
  }
  __syncthreads();
  GP_SORT_STAMP(tile, 6);  // staged in LDS
  const int tile_n = min(kSortTile, n - tile * kSortTile);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int pos = r * 256 + (int)threadIdx.x;
    if (pos < tile_n) {
      const unsigned k = skeys[pos];
      const int out = out_delta[(k >> shift) & 255u] + pos;
      if ((unsigned)out < (unsigned)n) {  // (always, unless a wait expired above and the prefix is incomplete: the pass is void then, but it must not store outside the arrays)
        keys_out[out] = (ticket_classes <= -kSortCorruptHook && tile == 0) ? (0xfffffff0u | (unsigned)(pos & 15)) : k;  // (test hook: garbage where a voided pass leaves holes)
        vals_out[out] = svals[pos];
      }
    }
  }
  GP_SORT_STAMP(tile, 7);  // stores issued
}

// 32-bit state words radix_sort_pairs needs for n elements and key_bits bits (zeroed by the caller on the sort's stream, or by the sort itself): per pass the ticket
// line + 256 words per group and per tile, and the passes' histograms behind them
inline size_t radix_sort_pass_words(int n) { return 256 + 256 * (radix_sort_groups(n) + ((size_t)n + kSortTile - 1) / kSortTile); }
inline size_t radix_sort_state_words32(int n, int key_bits) { return (size_t)((key_bits + 7) / 8) * radix_sort_pass_words(n) + 256 * kSortMaxPasses * kSortHistClasses; }
// where the histograms [class][pass][256] live inside the state (a caller that counts the digits itself adds them there, behind its fill, before the sort is issued)
inline unsigned* radix_sort_hist(unsigned* state, int n, int key_bits) { return state + (size_t)((key_bits + 7) / 8) * radix_sort_pass_words(n); }

// Sorts n pairs by the low `key_bits` (<= 32) bits of the key, stable.  (keys_a, vals_a) hold the input -- with vals_iota the values are
// taken to be 0..n-1 and vals_a is only storage; the passes ping-pong between the a and b buffers; *result_in_b tells where the
// sorted pairs ended up.  n < 2^30.
// state: radix_sort_state_words32(n, key_bits) 32-bit words; zeroed = the caller has zeroed them on `s` (a build zeroes all its states with one fill);
// hist_ready = the caller has also counted the digits into radix_sort_hist(state, n, key_bits) (needs zeroed)
// ticket_classes: kSortTicketClasses, or 1 for the form that cannot deadlock whatever order the device starts workgroups in (negative: test hook, see above).
// The caller checks radix_sort_fault_words behind the sort (the builds let their last kernel carry the words to the host) and sorts again with one class if set.
inline int radix_sort_pairs(unsigned* keys_a, int* vals_a, unsigned* keys_b, int* vals_b, int n, int key_bits, bool vals_iota, unsigned* state, bool zeroed, bool hist_ready,
                            hipStream_t s, bool* result_in_b, int ticket_classes = kSortTicketClasses) {
  *result_in_b = false;
  if (n <= 0) return GP_OK;
  if (n >= (1 << 30) || key_bits > 8 * kSortMaxPasses) return fail(GP_ERROR_INVALID_ARGUMENT, "radix_sort_pairs: n must be below 2^30 and the key at most 32 bits");
  const int tiles = (n + kSortTile - 1) / kSortTile, passes = (key_bits + 7) / 8, groups = (int)radix_sort_groups(n);
  if (!zeroed) GP_HIP(hipMemsetAsync(state, 0, sizeof(unsigned) * radix_sort_state_words32(n, key_bits), s));
  unsigned* hist = radix_sort_hist(state, n, key_bits);
  if (!(zeroed && hist_ready)) {
    hipLaunchKernelGGL(radix_hist_all_kernel<0>, dim3(tiles), dim3(256), 0, s, (const unsigned*)keys_a, n, passes, hist);
    GP_HIP(hipGetLastError());
  }
  bool in_a = true, first = true;
  for (int p = 0; p < passes; p++) {
    const unsigned* kin = in_a ? keys_a : keys_b;
    const int* vin = (first && vals_iota) ? nullptr : (in_a ? vals_a : vals_b);
    unsigned* kout = in_a ? keys_b : keys_a;
    int* vout = in_a ? vals_b : vals_a;
    hipLaunchKernelGGL(radix_onesweep_kernel<0>, dim3(tiles), dim3(256), 0, s, kin, vin, n, 8 * p, (const unsigned*)(hist + 256 * p), state + (size_t)p * radix_sort_pass_words(n),
                       groups, kout, vout, ticket_classes);
    GP_HIP(hipGetLastError());
    in_a = !in_a;
    first = false;
  }
  *result_in_b = !in_a;
  return GP_OK;
}

// device side: OR of the passes' fault words (for the kernel that reports a build's results to the host)
__device__ __forceinline__ unsigned radix_sort_faults(const unsigned* __restrict__ state, unsigned pass_words, int passes) {
  unsigned f = 0;
  for (int p = 0; p < passes; p++) f |= __hip_atomic_load(state + (size_t)p * pass_words + kSortFaultWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f;
}
// host side, synchronous (callers without such a kernel): 1 when some pass of the sort gave up a wait
inline int radix_sort_fault(const unsigned* state, int n, int key_bits, hipStream_t s, bool* fault) {
  *fault = false;
  const int passes = (key_bits + 7) / 8;
  for (int p = 0; p < passes; p++) {
    unsigned w = 0;
    GP_HIP(hipMemcpyAsync(&w, state + (size_t)p * radix_sort_pass_words(n) + kSortFaultWord, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    if (w) *fault = true;
  }
  return GP_OK;
}

}  // namespace gp
