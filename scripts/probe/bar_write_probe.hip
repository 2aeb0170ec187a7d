// bar_write_probe.hip -- can the host write straight into device memory (fine-grained allocation, large BAR), how fast, and does a kernel launched right behind the
// writes see them?  (Candidate home for the poses of a synchronous batched call: today the tile kernel's workgroups read them zero-copy from host memory,
// one PCIe round trip per workgroup.)
// Build: hipcc --offload-arch=gfx950 -O2 -o bar_write_probe bar_write_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e__ = (x);                                                            \
    if (e__ != hipSuccess) {                                                         \
      fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e__), __LINE__); \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

__global__ void check_kernel(const double* p, int n, double expect, int* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && p[i] != expect + i) atomicAdd(bad, 1);
}

int main() {
  CK(hipSetDevice(0));
  int direct = 0;
  (void)hipDeviceGetAttribute(&direct, hipDeviceAttributeDirectManagedMemAccessFromHost, 0);
  printf("hipDeviceAttributeDirectManagedMemAccessFromHost = %d\n", direct);
  const int n = 8192;  // 64 KB = the poses of 512 factors
  double* dev = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&dev, sizeof(double) * n, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(fine-grained): %s, ptr %p\n", hipGetErrorString(e), (void*)dev);
  if (e != hipSuccess) return 1;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, dev) == hipSuccess) printf("memory type %d, device pointer %p, host pointer %p\n", (int)attr.type, attr.devicePointer, attr.hostPointer);
  int* bad = nullptr;
  CK(hipMalloc(&bad, sizeof(int)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<double> src(n);
  int total_bad = 0;
  double best_us = 1e9;
  for (int rep = 0; rep < 200; rep++) {
    const double expect = 1000.0 * rep;
    for (int i = 0; i < n; i++) src[i] = expect + i;
    CK(hipMemsetAsync(bad, 0, sizeof(int), s));
    CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    memcpy(dev, src.data(), sizeof(double) * n);  // host stores into device memory through the BAR
    __builtin_ia32_sfence();
    const auto t1 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, s, (const double*)dev, n, expect, bad);
    int h = -1;
    CK(hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    total_bad += h;
    best_us = std::min(best_us, std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  printf("200 rounds of 64 KB host -> device stores + kernel right behind: %d stale values in total; fastest 64 KB store %.2f us\n", total_bad, best_us);
  return total_bad ? 3 : 0;
}
