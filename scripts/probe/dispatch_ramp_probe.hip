// Probe (round 4): what sets the dispatch ramp of the 1024-workgroup tile launch behind an idle queue -- the number of WORKGROUPS or the number of
// WAVES?  The C2 step's kernel spends 1.6-1.9 us placing its 1024 x 256-thread workgroups when the queue stood empty before it, of an
// ~12 us kernel.  Here: empty-bodied kernels that occupy the chip like the tile kernel (16 waves per CU, one resident round) in three shapes -- 1024 x 256,
// 512 x 512, 256 x 1024 threads -- with the tile kernel's LDS footprint or none, with its ~1.2 KB of kernel arguments or none; every workgroup stamps its
// start on the 100 MHz clock and then holds its slot for `hold` ticks.  Launch pattern = the step's: launch, wait for completion on the host, idle ~10 us.
// Reported per shape: median over launches of (last start - first start) and of (last start of each XCD - first start), in us.
// Build + run: hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_ramp_probe scripts/probe/dispatch_ramp_probe.hip && /tmp/dispatch_ramp_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                       \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

struct BigArgs {
  double pad[150];  // ~1.2 KB, like InlinePoses
};

template <int THREADS, int LDS_BYTES>
__global__ void __launch_bounds__(THREADS) ramp_kernel(unsigned long long* stamps, unsigned long long hold, BigArgs big, int use_big) {
  __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 16];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x] = t0;
    lds[0] = (char)(use_big ? (int)big.pad[blockIdx.x % 150] : 0);
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < hold) __builtin_amdgcn_s_sleep(2);
  if (threadIdx.x == 1 && lds[0] == 77) stamps[blockIdx.x] = 0;  // keeps the LDS array alive
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int THREADS, int LDS_BYTES>
int run(const char* name, int wgs, int use_big, hipStream_t s, unsigned long long* d_stamps, int back_to_back) {
  const int ITERS = 200;
  std::vector<unsigned long long> h(wgs);
  std::vector<double> total, wall;
  BigArgs big{};
  for (int it = 0; it < ITERS + 20; it++) {
    const double t0 = now_us();
    hipLaunchKernelGGL((ramp_kernel<THREADS, LDS_BYTES>), dim3(wgs), dim3(THREADS), 0, s, d_stamps, 500ull /* 5 us */, big, use_big);
    if (back_to_back) hipLaunchKernelGGL((ramp_kernel<THREADS, LDS_BYTES>), dim3(wgs), dim3(THREADS), 0, s, d_stamps, 500ull, big, use_big);
    CHECK(hipStreamSynchronize(s));
    const double t1 = now_us();
    CHECK(hipMemcpy(h.data(), d_stamps, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost));
    while (now_us() - t1 < 12.0) {
    }
    if (it < 20) continue;
    const unsigned long long lo = *std::min_element(h.begin(), h.end()), hi = *std::max_element(h.begin(), h.end());
    total.push_back((hi - lo) / 100.0);
    wall.push_back(t1 - t0);
  }
  std::sort(total.begin(), total.end());
  std::sort(wall.begin(), wall.end());
  printf("%-44s wgs %4d threads %4d lds %6d args %s %s: ramp (first start -> last start) median %.2f us  p10 %.2f  p90 %.2f ; host launch->sync median %.1f us\n", name, wgs, THREADS,
         LDS_BYTES, use_big ? "1.2KB" : "small", back_to_back ? "second of two back to back" : "behind an idle queue", total[total.size() / 2], total[total.size() / 10],
         total[total.size() * 9 / 10], wall[wall.size() / 2]);
  return 0;
}

int main() {
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* d_stamps;
  CHECK(hipMalloc(&d_stamps, sizeof(unsigned long long) * 4096));
  for (int rep = 0; rep < 2; rep++) {
    if (run<256, 34816>("tile-kernel shape", 1024, 1, s, d_stamps, 0)) return 1;
    if (run<256, 34816>("tile-kernel shape, small args", 1024, 0, s, d_stamps, 0)) return 1;
    if (run<256, 0>("tile-kernel shape, no LDS", 1024, 1, s, d_stamps, 0)) return 1;
    if (run<512, 69632>("8-wave workgroups", 512, 1, s, d_stamps, 0)) return 1;
    if (run<1024, 139264>("16-wave workgroups", 256, 1, s, d_stamps, 0)) return 1;
    if (run<256, 34816>("three quarters of the round", 768, 1, s, d_stamps, 0)) return 1;
    if (run<256, 34816>("half the round", 512, 1, s, d_stamps, 0)) return 1;
    if (run<128, 17408>("2-wave workgroups", 2048, 1, s, d_stamps, 0)) return 1;
    if (run<64, 8704>("1-wave workgroups", 4096, 1, s, d_stamps, 0)) return 1;
  }
  return 0;
}
