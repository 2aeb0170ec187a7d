// gp_scan.hpp -- exclusive prefix sum of a strided int array on the device (three small launches: per-block scan,
// scan of the block sums by one workgroup, add).  Used by the occupancy-grid build (gp_voxelmap.hip); the arrays are a few
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

// scratch: at least ceil(m / 1024) + 1 ints.  `in` and `out` may alias only when they are the same array with the same stride.
inline int exclusive_scan_strided(const int* in, int in_stride, int* out, int out_stride, long long m, int* scratch, hipStream_t s) {
  if (m <= 0) return GP_OK;
  const int nb = (int)((m + kScanThreads - 1) / kScanThreads);
  hipLaunchKernelGGL(strided_scan_block_kernel<0>, dim3(nb), dim3(kScanThreads), 0, s, in, in_stride, out, out_stride, scratch, m);
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(strided_scan_sums_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, scratch, nb, scratch + nb);
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(strided_scan_add_kernel<0>, dim3(nb), dim3(kScanThreads), 0, s, out, out_stride, (const int*)scratch, m);
  GP_HIP(hipGetLastError());
  return GP_OK;
}

}  // namespace gp
