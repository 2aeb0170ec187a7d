// Probe (round 4): where does a radix pass of gp_sort.hpp spend its time?  Every tile stamps its phases on the 100 MHz clock (GP_SORT_TRACE build of the header); the
// probe sorts 1 M / 2 M random 22-bit keys and prints, per pass, the median over tiles of every phase's duration and the spread of the tiles' start and end times.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGP_SORT_TRACE -I gtsam_points_amd/csrc -I include -o /tmp/sort_probe scripts/probe/sort_probe.hip \
//              && /tmp/sort_probe     (NOT linked with the library: its copy of the kernel template would be the one launched)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "gp_sort.hpp"

#define CHECK(x)                                                       \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

namespace gp {
int fail(int code, const std::string& msg) {
  printf("%s\n", msg.c_str());
  return code;
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  printf("%s:%d %s: %s\n", file, line, what, hipGetErrorString(e));
  return 1;
}
}  // namespace gp

static double med(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int run(int n, int key_bits) {
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<unsigned> h(n);
  std::mt19937 rng(1);
  for (auto& k : h) k = rng() & ((1u << key_bits) - 1u);
  unsigned *ka, *kb, *state;
  int *va, *vb;
  CHECK(hipMalloc(&ka, 4 * (size_t)n));
  CHECK(hipMalloc(&kb, 4 * (size_t)n));
  CHECK(hipMalloc(&va, 4 * (size_t)n));
  CHECK(hipMalloc(&vb, 4 * (size_t)n));
  const size_t words = gp::radix_sort_state_words32(n, key_bits);
  CHECK(hipMalloc(&state, 4 * words));
  const int tiles = (n + gp::kSortTile - 1) / gp::kSortTile, passes = (key_bits + 7) / 8;
  unsigned long long* trace;
  CHECK(hipMalloc(&trace, 8 * 16 * (size_t)tiles));
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(gp::g_sort_trace), &trace, sizeof(trace)));
  std::vector<unsigned long long> t(16 * (size_t)tiles);
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipMemcpy(ka, h.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
    bool in_b;
    // one pass at a time so that the stamps of a pass can be read: the sort itself is run whole for the check below
    CHECK(hipMemsetAsync(state, 0, 4 * words, s));
    unsigned* hist = gp::radix_sort_hist(state, n, key_bits);
    hipLaunchKernelGGL(gp::radix_hist_all_kernel<0>, dim3(tiles), dim3(256), 0, s, (const unsigned*)ka, n, passes, hist);
    unsigned *kin = ka, *kout = kb;
    int *vin = nullptr, *vout = vb;
    for (int p = 0; p < passes; p++) {
      CHECK(hipMemsetAsync(trace, 0, 8 * 16 * (size_t)tiles, s));
      hipLaunchKernelGGL(gp::radix_onesweep_kernel<0>, dim3(tiles), dim3(256), 0, s, (const unsigned*)kin, (const int*)vin, n, 8 * p, (const unsigned*)(hist + 256 * p),
                         state + (size_t)p * gp::radix_sort_pass_words(n), (int)gp::radix_sort_groups(n), kout, vout);
      CHECK(hipStreamSynchronize(s));
      CHECK(hipMemcpy(t.data(), trace, 8 * t.size(), hipMemcpyDeviceToHost));
      if (rep == 2) {
        unsigned long long first = ~0ull, last = 0;
        for (int i = 0; i < tiles; i++) first = std::min(first, t[16 * (size_t)i]), last = std::max(last, t[16 * (size_t)i + 7]);
        printf("n %d pass %d: first ticket -> last tile's stores issued %.2f us;", n, p, (last - first) / 100.0);
        std::vector<double> start;
        for (int i = 0; i < tiles; i++) start.push_back((t[16 * (size_t)i] - first) / 100.0);
        printf(" ticket time median %.2f max %.2f us; phases (median over tiles, us):", med(start), *std::max_element(start.begin(), start.end()));
        const char* names[7] = {"load+count", "rank", "rank barrier", "offsets", "barrier", "stage", "store"};
        for (int k = 0; k < 7; k++) {
          std::vector<double> d;
          for (int i = 0; i < tiles; i++) d.push_back((double)(long long)(t[16 * (size_t)i + k + 1] - t[16 * (size_t)i + k]) / 100.0);
          printf(" %s %.2f (max %.2f)", names[k], med(d), *std::max_element(d.begin(), d.end()));
        }
        printf("\n");
      }
      std::swap(kin, kout);
      vin = vout;
      vout = (vout == vb) ? va : vb;
    }
    (void)in_b;
  }
  // check: whole sort against std::stable_sort order of the keys
  CHECK(hipMemcpy(ka, h.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
  bool in_b = false;
  unsigned long long* null_trace = nullptr;
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(gp::g_sort_trace), &null_trace, sizeof(null_trace)));
  if (gp::radix_sort_pairs(ka, va, kb, vb, n, key_bits, true, state, false, false, s, &in_b) != 0) return 1;
  CHECK(hipStreamSynchronize(s));
  std::vector<unsigned> out(n);
  std::vector<int> ov(n);
  CHECK(hipMemcpy(out.data(), in_b ? kb : ka, 4 * (size_t)n, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ov.data(), in_b ? vb : va, 4 * (size_t)n, hipMemcpyDeviceToHost));
  std::vector<int> idx(n);
  for (int i = 0; i < n; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return h[a] < h[b]; });
  int bad = 0;
  for (int i = 0; i < n; i++) bad += (ov[i] != idx[i]) || (out[i] != h[idx[i]]);
  printf("n %d: sorted order %s (%d mismatches)\n", n, bad ? "WRONG" : "matches std::stable_sort", bad);
  // timing of the whole sort, events
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 10; rep++) {
    CHECK(hipMemcpy(ka, h.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0, s));
    if (gp::radix_sort_pairs(ka, va, kb, vb, n, key_bits, true, state, false, false, s, &in_b) != 0) return 1;
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  printf("n %d: whole sort (fill + histogram + %d passes) best of 10: %.1f us\n", n, passes, best * 1e3);
  return 0;
}

int main() {
  if (run(1000000, 22)) return 1;
  if (run(2000000, 22)) return 1;
  return 0;
}
