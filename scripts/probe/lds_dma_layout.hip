// Probe: where does global_load_lds_dwordx3 / dwordx4 put each lane's bytes in LDS, and does the instruction offset move the LDS address?
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_layout lds_dma_layout.hip ; prints, per variant, for LDS dword k the global dword it holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GP_GLOBAL __attribute__((address_space(1)))
#define GP_LDS __attribute__((address_space(3)))

template <int MODE>
__global__ void probe(const int* g, int* out) {
  __shared__ __attribute__((aligned(16))) int sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = -1;
  __syncthreads();
  const unsigned lane = threadIdx.x;
  const unsigned lds = (unsigned)(size_t)(GP_LDS int*)sm;
  const unsigned long long base = (unsigned long long)g;
  unsigned saved;
  if (MODE == 0) {  // x3, offset 0, voff = lane * 12
    unsigned voff = lane * 12;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(voff), "s"(base), "s"(lds) : "memory");
  } else if (MODE == 1) {  // x3, offset 768
    unsigned voff = lane * 12;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx3 %1, %2 offset:768\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(voff), "s"(base), "s"(lds) : "memory");
  } else if (MODE == 2) {  // x4, offset 0, voff = lane * 16
    unsigned voff = lane * 16;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(voff), "s"(base), "s"(lds) : "memory");
  } else if (MODE == 3) {  // x4, offset 1024
    unsigned voff = lane * 16;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(voff), "s"(base), "s"(lds) : "memory");
  } else if (MODE == 4) {  // builtin x3
    __builtin_amdgcn_global_load_lds((const GP_GLOBAL void*)((const GP_GLOBAL char*)g + lane * 12), (GP_LDS void*)sm, 12, 0, 0);
  } else if (MODE == 5) {  // x1 (dword), offset 0, voff = lane*4
    unsigned voff = lane * 4;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(voff), "s"(base), "s"(lds) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = sm[i];
}

int main() {
  std::vector<int> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = i;
  int *g, *o;
  hipMalloc(&g, 4096 * 4);
  hipMalloc(&o, 1024 * 4);
  hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  std::vector<int> r(1024);
  const char* names[] = {"x3 off0", "x3 off768", "x4 off0", "x4 off1024", "builtin x3", "x1 off0"};
  for (int mode = 0; mode < 6; mode++) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, g, o); break;
      case 1: hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, g, o); break;
      case 2: hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, g, o); break;
      case 3: hipLaunchKernelGGL(probe<3>, 1, 64, 0, 0, g, o); break;
      case 4: hipLaunchKernelGGL(probe<4>, 1, 64, 0, 0, g, o); break;
      case 5: hipLaunchKernelGGL(probe<5>, 1, 64, 0, 0, g, o); break;
    }
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1, count = 0;
    for (int i = 0; i < 1024; i++)
      if (r[i] >= 0) { if (first < 0) first = i; last = i; count++; }
    printf("%-12s written dwords %d, LDS dword range [%d, %d]; first 20 of range:", names[mode], count, first, last);
    for (int i = first; i >= 0 && i < first + 20 && i < 1024; i++) printf(" %d", r[i]);
    printf(" | around dword 48*4:");
    for (int i = first + 188; i >= 0 && i < first + 200 && i < 1024; i++) printf(" %d", r[i]);
    printf("\n");
  }
  return 0;
}
