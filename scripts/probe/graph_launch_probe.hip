// Probe: would a hipGraph take anything off the host-visible latency of the synchronous two-kernel pass (tile kernel -> finalize -> completion
// word in host-mapped memory, polled by the host)?  Times, host to host, (a) two direct launches, (b) one hipGraphLaunch of the same two
// kernel nodes with fixed parameters (the step's inputs read through pinned memory), (c) the graph with its first node's parameters
// re-set before every launch (what the in-argument pose of the single-factor path would need).
// Build + run: hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_launch_probe scripts/probe/graph_launch_probe.hip && /tmp/graph_launch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));            \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

// stands in for the tile kernel: 1024 workgroups, each busy for `ticks` of the 100 MHz device clock, one partial per workgroup
__global__ void __launch_bounds__(256) body_kernel(double* partials, unsigned long long ticks, double bias) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
  if (threadIdx.x == 0) partials[blockIdx.x] = bias + (double)blockIdx.x;
}

// stands in for the split finalize: 8 workgroups add their share of the partials, store a sum and a completion word into host memory
__global__ void __launch_bounds__(256) tail_kernel(const double* partials, int n, double* out_host, unsigned long long* done_host, const unsigned long long* seq_host) {
  __shared__ double s[256];
  double a = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) a += partials[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out_host[blockIdx.x] = s[0];
    __threadfence_system();
    __hip_atomic_store(done_host + blockIdx.x, *seq_host, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int WG = 1024, PARTS = 8, ITERS = 1500;
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double* partials;
  CHECK(hipMalloc(&partials, WG * sizeof(double)));
  double* out_host;
  unsigned long long *done_host, *seq_host;
  CHECK(hipHostMalloc(&out_host, PARTS * sizeof(double), hipHostMallocMapped));
  CHECK(hipHostMalloc(&done_host, PARTS * sizeof(unsigned long long), hipHostMallocMapped));
  CHECK(hipHostMalloc(&seq_host, sizeof(unsigned long long), hipHostMallocMapped));
  for (int i = 0; i < PARTS; i++) done_host[i] = 0;
  unsigned long long seq = 0;
  auto wait = [&](unsigned long long want) {
    for (int p = 0; p < PARTS; p++)
      while (__atomic_load_n(done_host + p, __ATOMIC_ACQUIRE) != want) {
      }
  };
  for (unsigned long long ticks : {0ull, 300ull, 600ull, 1200ull, 5000ull, 22000ull}) {  // 0 ... 220 us of "tile kernel" (C1 3.5, C2 12, C3 56, C4 shard 215)
    double bias = 1.0;
    auto direct = [&]() {
      *seq_host = ++seq;
      hipLaunchKernelGGL(body_kernel, dim3(WG), dim3(256), 0, s, partials, ticks, bias);
      hipLaunchKernelGGL(tail_kernel, dim3(PARTS), dim3(256), 0, s, partials, WG, out_host, done_host, seq_host);
      wait(seq);
    };
    // the graph: captured once
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(body_kernel, dim3(WG), dim3(256), 0, s, partials, ticks, bias);
    hipLaunchKernelGGL(tail_kernel, dim3(PARTS), dim3(256), 0, s, partials, WG, out_host, done_host, seq_host);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0;
    CHECK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CHECK(hipGraphGetNodes(g, nodes.data(), &nn));
    hipGraphNode_t body_node = nullptr;
    for (auto n : nodes) {
      hipKernelNodeParams kp;
      if (hipGraphKernelNodeGetParams(n, &kp) == hipSuccess && kp.func == (void*)body_kernel) body_node = n;
    }
    auto graph_fixed = [&]() {
      *seq_host = ++seq;
      hipGraphLaunch(ge, s);
      wait(seq);
    };
    auto graph_setparams = [&]() {
      *seq_host = ++seq;
      bias += 1.0;
      void* args[3] = {&partials, (void*)&ticks, &bias};
      hipKernelNodeParams kp{};
      kp.func = (void*)body_kernel;
      kp.gridDim = dim3(WG);
      kp.blockDim = dim3(256);
      kp.sharedMemBytes = 0;
      kp.kernelParams = args;
      kp.extra = nullptr;
      hipGraphExecKernelNodeSetParams(ge, body_node, &kp);
      hipGraphLaunch(ge, s);
      wait(seq);
    };
    auto measure = [&](const char* name, auto&& f) {
      for (int i = 0; i < 200; i++) f();
      std::vector<double> t(ITERS);
      for (int i = 0; i < ITERS; i++) {
        const double t0 = now_us();
        f();
        t[i] = now_us() - t0;
      }
      std::sort(t.begin(), t.end());
      printf("body %5.1f us  %-34s median %6.2f us  p10 %6.2f  p90 %6.2f\n", ticks / 100.0, name, t[ITERS / 2], t[ITERS / 10], t[ITERS * 9 / 10]);
    };
    // alternate the orders so that a drift of the box does not favour one form
    measure("two direct launches", direct);
    measure("hipGraphLaunch, fixed params", graph_fixed);
    if (body_node) measure("hipGraphLaunch + node SetParams", graph_setparams);
    measure("two direct launches (again)", direct);
    measure("hipGraphLaunch, fixed (again)", graph_fixed);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
  }
  printf("sum check %.1f\n", out_host[0]);
  return 0;
}
