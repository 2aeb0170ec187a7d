// Probe (round 5): the FLOOR of a VGICP error evaluation from cached linearisation state, and what writing that cache would add to the linearise.
// The cache the reference CPU factor keeps per point (integrated_vgicp_factor_impl.hpp:138-140: the fused-covariance inverse M; GPU analogue vgicp_derivatives.cuh:85-139)
// would be, here: M as six f32 (24 B) + the residual at the linearisation pose as three f32 (12 B; r(eval) = r(lin) + (T_lin - T_eval) p needs no voxel lookup) = 36 B per
// point, read beside the 12-B point: 48 B per point against the 36 B of the packed mirror that today's evaluation streams (plus its gathers, which hit L2).
//   kernel A  cached evaluation: streams 12 + 36 B per point (three 12-B rows + the point row per 64-point chunk, plain coalesced loads), r = r_lin + dR p + dt in f64,
//             e += r^T M r in f32, block reduction, one partial per workgroup -- everything a real kernel would do except the fused finalize
//   kernel B  the cache's stores alone: 36 B per point written (what the linearise would add)
// Launch pattern = the step's (launch, host waits, ~10 us idle) and back to back.  1 M points.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/error_cache_probe scripts/probe/error_cache_probe.hip && /tmp/error_cache_probe
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

struct Delta {
  double r[9], t[3];
};

__global__ void __launch_bounds__(256) cached_error_kernel(const float* __restrict__ pts, const float* __restrict__ cache, int n, Delta d, double* __restrict__ partials) {
  __shared__ double wsum[4];
  float acc = 0.f;
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int chunk = i >> 6, l = i & 63;
    const float* c = cache + (size_t)chunk * 576;  // [64 x r_lin | 64 x (m0 m1 m2) | 64 x (m3 m4 m5)], 12 B each
    const float px = __builtin_nontemporal_load(pts + 3 * (size_t)i), py = __builtin_nontemporal_load(pts + 3 * (size_t)i + 1), pz = __builtin_nontemporal_load(pts + 3 * (size_t)i + 2);
    const float r0 = __builtin_nontemporal_load(c + 3 * l), r1 = __builtin_nontemporal_load(c + 3 * l + 1), r2 = __builtin_nontemporal_load(c + 3 * l + 2);
    const float m0 = __builtin_nontemporal_load(c + 192 + 3 * l), m1 = __builtin_nontemporal_load(c + 193 + 3 * l), m2 = __builtin_nontemporal_load(c + 194 + 3 * l);
    const float m3 = __builtin_nontemporal_load(c + 384 + 3 * l), m4 = __builtin_nontemporal_load(c + 385 + 3 * l), m5 = __builtin_nontemporal_load(c + 386 + 3 * l);
    const double x = px, y = py, z = pz;
    const float rx = r0 + (float)(d.r[0] * x + d.r[1] * y + d.r[2] * z + d.t[0]), ry = r1 + (float)(d.r[3] * x + d.r[4] * y + d.r[5] * z + d.t[1]),
                rz = r2 + (float)(d.r[6] * x + d.r[7] * y + d.r[8] * z + d.t[2]);
    const float mrx = m0 * rx + m1 * ry + m2 * rz, mry = m1 * rx + m3 * ry + m4 * rz, mrz = m2 * rx + m4 * ry + m5 * rz;
    acc += rx * mrx + ry * mry + rz * mrz;
  }
  double v = acc;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void __launch_bounds__(256) cache_store_kernel(float* __restrict__ cache, int n, float v) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int chunk = i >> 6, l = i & 63;
    float* c = cache + (size_t)chunk * 576;
#pragma unroll
    for (int row = 0; row < 3; row++) {
      __builtin_nontemporal_store(v + (float)l, c + 192 * row + 3 * l);
      __builtin_nontemporal_store(v, c + 192 * row + 3 * l + 1);
      __builtin_nontemporal_store(v - 1.f, c + 192 * row + 3 * l + 2);
    }
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  const int n = 1000000, wgs = 1024;
  float *pts, *cache;
  double* partials;
  CHECK(hipMalloc(&pts, sizeof(float) * 3 * n));
  CHECK(hipMalloc(&cache, sizeof(float) * 9 * (size_t)((n + 63) / 64) * 64));
  CHECK(hipMalloc(&partials, sizeof(double) * wgs));
  CHECK(hipMemset(pts, 0, sizeof(float) * 3 * n));
  CHECK(hipMemset(cache, 0, sizeof(float) * 9 * (size_t)((n + 63) / 64) * 64));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  Delta d{};
  for (int k = 0; k < 9; k++) d.r[k] = 1e-3 * (k + 1);
  for (int which = 0; which < 2; which++) {
    auto launch = [&]() {
      if (which == 0) hipLaunchKernelGGL(cached_error_kernel, dim3(wgs), dim3(256), 0, s, (const float*)pts, (const float*)cache, n, d, partials);
      else hipLaunchKernelGGL(cache_store_kernel, dim3(wgs), dim3(256), 0, s, cache, n, 0.5f);
    };
    for (int i = 0; i < 200; i++) launch();
    CHECK(hipStreamSynchronize(s));
    // back to back
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; i++) launch();
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // in step: launch, wait, idle 10 us
    std::vector<double> wall;
    for (int i = 0; i < 300; i++) {
      const double t0 = now_us();
      launch();
      CHECK(hipStreamSynchronize(s));
      const double t1 = now_us();
      wall.push_back(t1 - t0);
      while (now_us() - t1 < 10.0) {
      }
    }
    const double bytes = which == 0 ? 48.0 * n : 36.0 * n;
    printf("%s: back to back %.2f us per launch (%.2f TB/s on its %.0f MB); launch + wait in the step's pattern: median %.1f us host to host\n",
           which == 0 ? "cached error evaluation (12 B point + 36 B cache per point, streamed)" : "cache stores alone (36 B per point written)         ", ms * 1e3 / 200.0,
           bytes / (ms * 1e-3 / 200.0) / 1e12, bytes / 1e6, med(wall));
  }
  return 0;
}
