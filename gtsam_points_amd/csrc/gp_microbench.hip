// gp_microbench.hip -- measurement kernels behind scripts/stream_bench.py, scripts/alu_rate.py and the PMC calibration pass.
// Nothing here is on the product path: these answer "what does it cost just to read the source bytes / gather the records /
// issue the arithmetic on this GPU", the numbers DESIGN.md section 8 prices the tile kernel against.
#include <cstdlib>

#include "gp_host.hpp"
#include "gp_sort.hpp"
#include "gp_vgicp_tile.hpp"

namespace gp {

// HBM-counter calibration: streams points (12 B) and covariances (36 B) with exactly the per-lane access pattern of the
// tile kernel and nothing else, so that rocprofv3's FETCH_SIZE can be scaled on a known byte count (48 * n)
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is uncalibrated for non-16-B-per-lane patterns).
__global__ void __launch_bounds__(256) calibration_stream_kernel(const float* __restrict__ points, const float* __restrict__ covs, int n, float* __restrict__ sink) {
  float s = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float* pp = points + 3 * (size_t)i;
    const float* cp = covs + 9 * (size_t)i;
    s += pp[0] + pp[1] + pp[2] + cp[0] + cp[3] + cp[4] + cp[6] + cp[7] + cp[8];
  }
  if (s == 123.456f) sink[0] = s;  // never true for real data; keeps the loads alive
}

// stream micro-benchmarks (what does it cost just to READ the 48*n source bytes at this problem size?)
//   mode 1: coalesced 16 B per lane, grid-stride;  mode 2: LDS-DMA, 12 KB per wave like kernel5
__global__ void __launch_bounds__(256) stream_float4_kernel(const float4* __restrict__ a, size_t n16, float* __restrict__ sink) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) sink[0] = s;
}

__global__ void __launch_bounds__(256) stream_ldsdma_kernel(const char* __restrict__ a, size_t bytes, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 12288];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = ((size_t)blockIdx.x * 4 + wave) * 12288;
  if (base + 12288 > bytes) return;
  char* wbase = smem + wave * 12288;
  const GP_GLOBAL char* g = (const GP_GLOBAL char*)a + base + lane * 16;
#pragma unroll
  for (int k = 0; k < 12; k++) __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)(g + k * 1024), (GP_LDS void*)(wbase + k * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float v = reinterpret_cast<float*>(wbase)[lane * 3] + reinterpret_cast<float*>(wbase)[2048 + lane];
  if (v == 123.456f) sink[0] = v;
}


// memory-pattern micro-benchmark of the tile kernel: per point 12 strided dwords of source + one 16-B key + one 64-B record
// from an L2-resident slot table.  GATHER: 0 none, 1 slot from the point index (independent of the source loads),
// 2 slot from the loaded source bits (dependent, as in the real kernel).  SOURCE: read the source or not.
template <int GATHER, bool SOURCE, int PPT>
__global__ void __launch_bounds__(256) pattern_kernel(const float* __restrict__ points, const float* __restrict__ covs, int n, const char* __restrict__ table,
                                                      uint32_t mask, float* __restrict__ sink) {
  float acc = 0.f;
  const GP_GLOBAL float* gp_ = (const GP_GLOBAL float*)points;
  const GP_GLOBAL float* gc_ = (const GP_GLOBAL float*)covs;
  const GP_GLOBAL char* keys = (const GP_GLOBAL char*)table;
  const GP_GLOBAL char* recs = keys + 16 * ((size_t)mask + 1);
  float src[PPT];
  int idx[PPT];
#pragma unroll
  for (int u = 0; u < PPT; u++) {
    const int i = (blockIdx.x * PPT + u) * 256 + threadIdx.x;
    idx[u] = i;
    src[u] = 0.f;
    if (SOURCE && i < n) {
      const GP_GLOBAL float* pp = gp_ + 3 * (size_t)i;
      const GP_GLOBAL float* cp = gc_ + 9 * (size_t)i;
      src[u] = pp[0] + pp[1] + pp[2] + cp[0] + cp[3] + cp[4] + cp[6] + cp[7] + cp[8];
    }
  }
#pragma unroll
  for (int u = 0; u < PPT; u++) {
    acc += src[u];
    if (GATHER && idx[u] < n) {
      uint32_t h = (uint32_t)(idx[u] >> 2) * 2654435761u;
      if (GATHER == 2) h ^= __float_as_uint(src[u]);  // zero-filled inputs: same slot, but the address now waits for the data
      const uint32_t slot = (h >> 7) & mask;
      const v4i key = *(const GP_GLOBAL v4i*)(keys + 16 * (size_t)slot);
      const GP_GLOBAL char* rec = recs + 64 * (size_t)slot;
      const v4f head = *(const GP_GLOBAL v4f*)rec;
      const v2d c01 = *(const GP_GLOBAL v2d*)(rec + 16), c23 = *(const GP_GLOBAL v2d*)(rec + 32), c45 = *(const GP_GLOBAL v2d*)(rec + 48);
      acc += (float)key.x + head.x + (float)(c01.x + c23.y + c45.x);
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// same bytes as pattern_kernel, but shaped for the vector-memory address pipeline: the source of a wave's 128 points is
// read as fully coalesced 16-B lanes (7 instructions instead of 24) and turned through LDS; a voxel record is read by the
// four lanes of a quad together (4 instructions touching 16 records each instead of 4 touching 64) and turned through LDS.
template <bool COOP_SOURCE, bool COOP_GATHER>
__global__ void __launch_bounds__(256) pattern_coop_kernel(const float* __restrict__ points, const float* __restrict__ covs, int n,
                                                           const char* __restrict__ table, uint32_t mask, float* __restrict__ sink) {
  constexpr int PPT = 2;
  __shared__ __attribute__((aligned(16))) float lsrc[4][128 * 12];   // 6 KB per wave
  __shared__ __attribute__((aligned(16))) float lrec[4][64 * 20];    // 64 records at an 80-B stride
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const GP_GLOBAL char* keys = (const GP_GLOBAL char*)table;
  const GP_GLOBAL char* recs = keys + 16 * ((size_t)mask + 1);
  const int first = (blockIdx.x * 4 + wave) * 128;  // this wave's 128 consecutive points
  if (first >= n) return;
  float src[PPT];
  if (COOP_SOURCE) {
    const GP_GLOBAL v4f* gp4 = (const GP_GLOBAL v4f*)((const GP_GLOBAL char*)points + (size_t)first * 12);
    const GP_GLOBAL v4f* gc4 = (const GP_GLOBAL v4f*)((const GP_GLOBAL char*)covs + (size_t)first * 36);
    v4f* lp4 = (v4f*)&lsrc[wave][0];
    v4f* lc4 = (v4f*)&lsrc[wave][128 * 3];
    const int np16 = min(128, n - first) * 12 / 16, nc16 = min(128, n - first) * 36 / 16;
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (k * 64 + lane < np16) lp4[k * 64 + lane] = gp4[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 5; k++)
      if (k * 64 + lane < nc16) lc4[k * 64 + lane] = gc4[k * 64 + lane];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const float* pp = &lsrc[wave][3 * (u * 64 + lane)];
      const float* cp = &lsrc[wave][128 * 3 + 9 * (u * 64 + lane)];
      src[u] = pp[0] + pp[1] + pp[2] + cp[0] + cp[3] + cp[4] + cp[6] + cp[7] + cp[8];
    }
  } else {
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const int i = first + u * 64 + lane;
      const GP_GLOBAL float* pp = (const GP_GLOBAL float*)points + 3 * (size_t)i;
      const GP_GLOBAL float* cp = (const GP_GLOBAL float*)covs + 9 * (size_t)i;
      src[u] = i < n ? pp[0] + pp[1] + pp[2] + cp[0] + cp[3] + cp[4] + cp[6] + cp[7] + cp[8] : 0.f;
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < PPT; u++) {
    const int i = first + u * 64 + lane;
    uint32_t h = (uint32_t)(i >> 2) * 2654435761u;
    h ^= __float_as_uint(src[u]);
    const uint32_t slot = (h >> 7) & mask;
    const v4i key = *(const GP_GLOBAL v4i*)(keys + 16 * (size_t)slot);
    v4f head;
    v2d c01, c23, c45;
    if (COOP_GATHER) {
      const int m = lane & 3;
      float* mine = &lrec[wave][0];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t sj = __shfl(slot, (lane & ~3) | j, 64);
        const v4f chunk = *(const GP_GLOBAL v4f*)(recs + 64 * (size_t)sj + 16 * m);
        *(v4f*)(mine + ((lane & ~3) | j) * 20 + 4 * m) = chunk;
      }
      const float* r = mine + lane * 20;
      head = *(const v4f*)r;
      c01 = *(const v2d*)(r + 4);
      c23 = *(const v2d*)(r + 8);
      c45 = *(const v2d*)(r + 12);
    } else {
      const GP_GLOBAL char* rec = recs + 64 * (size_t)slot;
      head = *(const GP_GLOBAL v4f*)rec;
      c01 = *(const GP_GLOBAL v2d*)(rec + 16);
      c23 = *(const GP_GLOBAL v2d*)(rec + 32);
      c45 = *(const GP_GLOBAL v2d*)(rec + 48);
    }
    acc += src[u] + (float)key.x + head.x + (float)(c01.x + c23.y + c45.x);
  }
  if (acc == 123.456f) sink[0] = acc;
}

// one wave that keeps the device "busy" for `us` microseconds (s_memrealtime: the 100 MHz constant clock): probe of whether the tile
// kernel's slower start behind an idle queue follows the device's utilisation
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
    t = __builtin_amdgcn_s_memrealtime();
  }
  if (sink && t == 1) sink[0] = t;
}

// issue-rate micro-benchmark: 8 independent chains of one VALU instruction, 512 instructions per lane per launch.
template <int OP>
__global__ void __launch_bounds__(256) alu_rate_kernel(float* __restrict__ sink, int reps) {
  double d[8];
  float f[8];
  int q[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    d[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    f[k] = 1.0f + 1e-6f * (threadIdx.x + k);
    q[k] = threadIdx.x + k;
  }
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[k]));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[k]));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[k]));
        if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));
        if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
        if (OP == 5) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[k]) : "v"(q[k]));
        if (OP == 6) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k]));
        if (OP == 7) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[k]));
        if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(q[k]) : "v"(q[(k + 1) & 7]));
        if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(q[k]));
        if (OP == 10) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
        if (OP == 11) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q[k]) : "v"(d[k]));
        if (OP == 12) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[k]));
        if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
        if (OP == 14) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
        if (OP == 15) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(d[k]));
      }
    }
  }
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) acc += d[k] + f[k] + q[k];
  if (acc == 123.456) sink[0] = (float)acc;
}

template <int OP>
static void launch_alu(hipStream_t s, float* sink, int blocks, int reps) {
  hipLaunchKernelGGL(alu_rate_kernel<OP>, dim3(blocks), dim3(256), 0, s, sink, reps);
}

}  // namespace gp

extern "C" {

int gp_debug_calibration_stream(const float* points_dev, const float* covs_dev, int n, int iters, gp_stream_t stream) {
  if (!points_dev || !covs_dev || n <= 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_calibration_stream: bad arguments");
  gp::DeviceArray sink;
  GP_TRY(sink.alloc(16));
  for (int i = 0; i < iters; i++)
    hipLaunchKernelGGL(gp::calibration_stream_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, points_dev, covs_dev, n, sink.as<float>());
  GP_HIP(hipGetLastError());
  GP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return GP_OK;
}

// mode 0: per-lane strided dword pattern (calibration kernel), 1: coalesced float4 grid-stride, 2: LDS-DMA 12 KB per wave.
// Reads the points array then the covs array (48*n bytes); returns the mean milliseconds per pass (HIP events).
int gp_debug_stream_bench(const float* points_dev, const float* covs_dev, int n, int mode, int iters, float* ms) {
  if (!points_dev || !covs_dev || n <= 0 || iters <= 0 || !ms) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_stream_bench: bad arguments");
  gp::DeviceArray sink;
  GP_TRY(sink.alloc(16));
  hipStream_t s;
  GP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  GP_HIP(hipEventCreate(&e0));
  GP_HIP(hipEventCreate(&e1));
  gp::DeviceArray table;
  const char* slots_env = getenv("GP_PATTERN_SLOTS");  // table size of the pattern micro-benchmarks (power of two)
  const uint32_t mask = (slots_env ? (uint32_t)atoi(slots_env) : 131072u) - 1;
  if (mode >= 3) {
    GP_TRY(table.alloc(80 * ((size_t)mask + 1)));
    GP_HIP(hipMemset(table.ptr, 0, 80 * ((size_t)mask + 1)));
  }
  auto launch = [&]() {
    const char* tb = table.as<char>();
    const unsigned g2 = (unsigned)((n + 511) / 512), g1 = (unsigned)((n + 255) / 256);
    if (mode == 3) {  // gather only
      hipLaunchKernelGGL((gp::pattern_kernel<1, false, 2>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode == 4) {  // source + independent gather
      hipLaunchKernelGGL((gp::pattern_kernel<1, true, 2>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode == 5) {  // source + dependent gather
      hipLaunchKernelGGL((gp::pattern_kernel<2, true, 2>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode == 6) {  // source only, same grid shape
      hipLaunchKernelGGL((gp::pattern_kernel<0, true, 2>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode == 7) {  // source + dependent gather, 1 point per thread
      hipLaunchKernelGGL((gp::pattern_kernel<2, true, 1>), dim3(g1), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode == 8) {  // source + dependent gather, 4 points per thread
      hipLaunchKernelGGL((gp::pattern_kernel<2, true, 4>), dim3((g1 + 3) / 4), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode >= 100 && mode < 116) {
      // n = workgroups, 64 reps x 64 instructions per lane
      using L = void (*)(hipStream_t, float*, int, int);
      static const L table_[16] = {gp::launch_alu<0>, gp::launch_alu<1>, gp::launch_alu<2>,  gp::launch_alu<3>,  gp::launch_alu<4>,  gp::launch_alu<5>,
                                   gp::launch_alu<6>, gp::launch_alu<7>, gp::launch_alu<8>,  gp::launch_alu<9>,  gp::launch_alu<10>, gp::launch_alu<11>,
                                   gp::launch_alu<12>, gp::launch_alu<13>, gp::launch_alu<14>, gp::launch_alu<15>};
      table_[mode - 100](s, sink.as<float>(), n, 64);
    } else if (mode >= 9 && mode <= 12) {
      if (mode == 9) hipLaunchKernelGGL((gp::pattern_coop_kernel<false, false>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
      if (mode == 10) hipLaunchKernelGGL((gp::pattern_coop_kernel<true, false>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
      if (mode == 11) hipLaunchKernelGGL((gp::pattern_coop_kernel<false, true>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
      if (mode == 12) hipLaunchKernelGGL((gp::pattern_coop_kernel<true, true>), dim3(g2), dim3(256), 0, s, points_dev, covs_dev, n, tb, mask, sink.as<float>());
    } else if (mode == 0) {
      hipLaunchKernelGGL(gp::calibration_stream_kernel, dim3(2048), dim3(256), 0, s, points_dev, covs_dev, n, sink.as<float>());
    } else if (mode == 1) {
      // one kernel over both arrays would need them contiguous; two launches back to back measure the same bytes
      hipLaunchKernelGGL(gp::stream_float4_kernel, dim3(2048), dim3(256), 0, s, (const float4*)covs_dev, (size_t)n * 36 / 16, sink.as<float>());
      hipLaunchKernelGGL(gp::stream_float4_kernel, dim3(1024), dim3(256), 0, s, (const float4*)points_dev, (size_t)n * 12 / 16, sink.as<float>());
    } else {
      const size_t bc = (size_t)n * 36, bp = (size_t)n * 12;
      hipLaunchKernelGGL(gp::stream_ldsdma_kernel, dim3((unsigned)(bc / 49152)), dim3(256), 0, s, (const char*)covs_dev, bc, sink.as<float>());
      hipLaunchKernelGGL(gp::stream_ldsdma_kernel, dim3((unsigned)(bp / 49152)), dim3(256), 0, s, (const char*)points_dev, bp, sink.as<float>());
    }
  };
  launch();
  GP_HIP(hipStreamSynchronize(s));
  GP_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; i++) launch();
  GP_HIP(hipEventRecord(e1, s));
  GP_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  GP_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s);
  return GP_OK;
}

/* measurement hook: one wave spins for `microseconds` on `stream` (asynchronous) */
// test hook for gp_sort.hpp (tests/test_sort_gpu.py): stable argsort of n keys by their low key_bits bits.  keys_dev is overwritten (ping-pong buffer);
// sorted keys / original indices are copied into keys_out_dev / vals_out_dev.  Synchronous.
int gp_debug_sort_pairs(unsigned* keys_dev, int n, int key_bits, unsigned* keys_out_dev, int* vals_out_dev, gp_stream_t stream) {
  if (n < 0 || key_bits < 1 || key_bits > 32 || (n > 0 && (!keys_dev || !keys_out_dev || !vals_out_dev))) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_sort_pairs: bad arguments");
  if (n == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  gp::DeviceArray keys_b, vals_a, vals_b, state;
  GP_TRY(keys_b.alloc(sizeof(unsigned) * (size_t)n));
  GP_TRY(vals_a.alloc(sizeof(int) * (size_t)n));
  GP_TRY(vals_b.alloc(sizeof(int) * (size_t)n));
  GP_TRY(state.alloc(sizeof(unsigned) * gp::radix_sort_state_words32(n, key_bits)));
  bool in_b = false;
  GP_TRY(gp::radix_sort_pairs(keys_dev, vals_a.as<int>(), keys_b.as<unsigned>(), vals_b.as<int>(), n, key_bits, true, state.as<unsigned>(), false, false, s, &in_b));
  GP_HIP(hipMemcpyAsync(keys_out_dev, in_b ? keys_b.ptr : (void*)keys_dev, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, s));
  GP_HIP(hipMemcpyAsync(vals_out_dev, in_b ? vals_b.ptr : vals_a.ptr, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
  GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

// the sort with a chosen number of ticket classes (gp_sort.hpp: 32 = the builds' fast path, 1 = the deadlock-free form, negative = tile 0 raises the fault word);
// *fault = some pass gave up a wait (the output is void then)
int gp_debug_sort_pairs_ex(unsigned* keys_dev, int n, int key_bits, unsigned* keys_out_dev, int* vals_out_dev, int ticket_classes, int* fault, gp_stream_t stream) {
  if (n < 0 || key_bits < 1 || key_bits > 32 || ticket_classes == 0 || !fault || (n > 0 && (!keys_dev || !keys_out_dev || !vals_out_dev)))
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_sort_pairs_ex: bad arguments");
  *fault = 0;
  if (n == 0) return GP_OK;
  hipStream_t s = (hipStream_t)stream;
  gp::DeviceArray keys_b, vals_a, vals_b, state;
  GP_TRY(keys_b.alloc(sizeof(unsigned) * (size_t)n));
  GP_TRY(vals_a.alloc(sizeof(int) * (size_t)n));
  GP_TRY(vals_b.alloc(sizeof(int) * (size_t)n));
  GP_TRY(state.alloc(sizeof(unsigned) * gp::radix_sort_state_words32(n, key_bits)));
  bool in_b = false, f = false;
  GP_TRY(gp::radix_sort_pairs(keys_dev, vals_a.as<int>(), keys_b.as<unsigned>(), vals_b.as<int>(), n, key_bits, true, state.as<unsigned>(), false, false, s, &in_b, ticket_classes));
  GP_TRY(gp::radix_sort_fault(state.as<unsigned>(), n, key_bits, s, &f));
  *fault = f ? 1 : 0;
  GP_HIP(hipMemcpyAsync(keys_out_dev, in_b ? keys_b.ptr : (void*)keys_dev, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, s));
  GP_HIP(hipMemcpyAsync(vals_out_dev, in_b ? vals_b.ptr : vals_a.ptr, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
  GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

// `workgroups` x 256 threads that each spin for `microseconds` (asynchronous): fills the CUs from another stream (the sort's forward-progress test)
int gp_debug_occupy(double microseconds, int workgroups, gp_stream_t stream) {
  if (workgroups <= 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_occupy: bad arguments");
  hipLaunchKernelGGL(gp::spin_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0), (unsigned long long*)nullptr);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

int gp_debug_spin(double microseconds, gp_stream_t stream) {
  hipLaunchKernelGGL(gp::spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0), (unsigned long long*)nullptr);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

}  // extern "C"
