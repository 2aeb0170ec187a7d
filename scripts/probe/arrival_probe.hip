// Probe (round 5): what the tail of the fused finalize is made of, and what it would be if the traffic of a part stayed inside its XCD.
// The fused linearise (gp_vgicp_stream.hpp) ends with: every workgroup stores its 32-double partial row write-through (sc0 sc1), waits for the
// acknowledgement, adds 1 to its part's arrival counter (agent scope, sc1), and the part's last arriver reads the part's 128 rows (sc1 loads), adds
// them up and hands 32 sums to the host: three dependent trips to the device's coherence point, ~1.4 us behind the last row (VERDICT r04 #1a).
// A part IS an XCD's workgroups (blockIdx % 8), and an XCD's L2 is coherent for its own CUs: stores, the atomic and the loads of a part could stop
// at that L2 (no sc1).  This probe runs the arrival protocol alone, in the step's launch pattern (launch, host waits, ~10 us idle), with every
// combination of {store, atomic, load} x {device-wide (sc1), XCD-local (no sc1)}, stamps the three phases on the 100 MHz clock, checks the sums
// (exact integers in f64) and that every workgroup really ran on XCD blockIdx % 8 (XCC_ID).
// Build + run: hipcc --offload-arch=gfx950 -O2 -o /tmp/arrival_probe scripts/probe/arrival_probe.hip && /tmp/arrival_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                       \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

constexpr int kWgs = 1024, kParts = 8, kRowsPerPart = kWgs / kParts, kStride = 32, kCtrStride = 512 /* x 8 B = 4 KB */;
#define GETREG_XCC_ID ((4 - 1) << 11 | 0 << 6 | 20)  // hwreg(HW_REG_XCC_ID, 0, 4)

template <int STORE, int ATOM, int LOAD>
__global__ void __launch_bounds__(256, 4) arrival_kernel(double* rows, unsigned long long* counters, unsigned long long target, double* out, unsigned long long* stamps,
                                                         unsigned seq) {
  __shared__ double wsum[4 * 32];
  __shared__ int last;
  const int bx = blockIdx.x % kParts, bq = blockIdx.x / kParts, row = bx * kRowsPerPart + bq;
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  // hold the slot ~3 us, end spread over ~1.5 us like the tile kernel's workgroups
  const unsigned long long hold = 300ull + ((blockIdx.x * 2654435761u + seq * 40503u) >> 8) % 150ull;
  while (__builtin_amdgcn_s_memrealtime() - t_start < hold) __builtin_amdgcn_s_sleep(2);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x < kStride) {
    double* dst = rows + (size_t)row * kStride + threadIdx.x;
    const double v = (double)(row * kStride + (int)threadIdx.x) + (double)seq;
    if (STORE == 0) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(dst), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : : "v"(dst), "v"(v) : "memory");
  }
  __syncthreads();
  unsigned long long t1 = 0, t2 = 0;
  if (threadIdx.x == 0) {
    t1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* ctr = counters + (size_t)bx * kCtrStride;
    unsigned long long seen, one = 1ull;
    if (ATOM == 0) asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(seen) : "v"(ctr), "v"(one) : "memory");
    else asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(seen) : "v"(ctr), "v"(one) : "memory");
    t2 = __builtin_amdgcn_s_memrealtime();
    last = seen + 1 == target;
  }
  __syncthreads();
  unsigned long long t3 = 0;
  if (last) {
    // finalize_part_rows' access pattern: 8 slices, a thread's 16 rows requested in one batch
    const int comp = threadIdx.x & 31, slice = threadIdx.x >> 5, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* base = rows + (size_t)bx * kRowsPerPart * kStride + comp;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const double* p = base + (size_t)(slice + 8 * k) * kStride;
      if (LOAD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(v[k]) : "v"(p) : "memory");
      else asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=&v"(v[k]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                 "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 :
                 : "memory");
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; k++) v[k] += v[k + w];
    double total = v[0];
    total += __shfl_xor(total, 32, 64);
    if (lane < 32) wsum[wave * 32 + lane] = total;
    __syncthreads();
    if (wave == 0 && lane < 32) {
      const double s = (wsum[lane] + wsum[64 + lane]) + (wsum[32 + lane] + wsum[96 + lane]);
      asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(out + bx * kStride + lane), "v"(s) : "memory");
    }
    t3 = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0) {
    unsigned long long* st = stamps + (size_t)blockIdx.x * 8;
    st[0] = t0, st[1] = t1, st[2] = t2, st[3] = t3, st[4] = __builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 0xf, st[5] = last ? 1 : 0, st[6] = t_start;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double>& v) {
  if (v.empty()) return -1;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int STORE, int ATOM, int LOAD>
int run(hipStream_t s, double* d_rows, unsigned long long* d_ctr, double* d_out, unsigned long long* d_st) {
  const int ITERS = 300;
  std::vector<unsigned long long> st(kWgs * 8);
  std::vector<double> out(kParts * kStride);
  std::vector<double> store_us, atom_us, fin_us, tail_us, store_last, atom_last;
  CHECK(hipMemset(d_ctr, 0, sizeof(unsigned long long) * kParts * kCtrStride));
  CHECK(hipDeviceSynchronize());
  int wrong = 0, moved = 0;
  for (int it = 0; it < ITERS + 20; it++) {
    const unsigned seq = (unsigned)it + 1;
    hipLaunchKernelGGL((arrival_kernel<STORE, ATOM, LOAD>), dim3(kWgs), dim3(256), 0, s, d_rows, d_ctr, (unsigned long long)kRowsPerPart * seq, d_out, d_st, seq);
    CHECK(hipStreamSynchronize(s));
    const double t1 = now_us();
    CHECK(hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(out.data(), d_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
    while (now_us() - t1 < 12.0) {
    }
    for (int p = 0; p < kParts; p++)
      for (int c = 0; c < kStride; c++) {
        double want = 0;
        for (int r = 0; r < kRowsPerPart; r++) want += (double)((p * kRowsPerPart + r) * kStride + c) + (double)seq;
        if (out[p * kStride + c] != want) wrong++;
      }
    if (it < 20) continue;
    unsigned long long last_row_in = 0, last_out = 0;
    for (int b = 0; b < kWgs; b++) {
      const unsigned long long* q = &st[(size_t)b * 8];
      if ((int)q[4] != b % kParts) moved++;
      store_us.push_back((q[1] - q[0]) / 100.0);
      atom_us.push_back((q[2] - q[1]) / 100.0);
      if (q[5]) {
        fin_us.push_back((q[3] - q[2]) / 100.0);
        store_last.push_back((q[1] - q[0]) / 100.0);
        atom_last.push_back((q[2] - q[1]) / 100.0);
        last_out = std::max(last_out, q[3]);
      }
      last_row_in = std::max(last_row_in, q[0]);
    }
    tail_us.push_back((double)(last_out - last_row_in) / 100.0);
  }
  printf("store %-9s atomic %-9s loads %-4s | row store+ack %.2f us (last arrivers %.2f) | arrival atomic %.2f (last %.2f) | last arriver: loads+tree+sums out %.2f | "
         "last workgroup's row ready -> last sums out %.2f us | wrong sums %d | workgroups off their XCD %d\n",
         STORE ? "plain" : "sc0 sc1", ATOM ? "sc0" : "sc0 sc1", LOAD ? "sc0" : "sc1", med(store_us), med(store_last), med(atom_us), med(atom_last), med(fin_us), med(tail_us), wrong,
         moved);
  return 0;
}

int main() {
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *d_rows, *d_out;
  unsigned long long *d_ctr, *d_st;
  CHECK(hipMalloc(&d_rows, sizeof(double) * kWgs * kStride));
  CHECK(hipMalloc(&d_out, sizeof(double) * kParts * kStride));
  CHECK(hipMalloc(&d_ctr, sizeof(unsigned long long) * kParts * kCtrStride));
  CHECK(hipMalloc(&d_st, sizeof(unsigned long long) * kWgs * 8));
  for (int rep = 0; rep < 2; rep++) {
    if (run<0, 0, 0>(s, d_rows, d_ctr, d_out, d_st)) return 1;  // today's protocol
    if (run<1, 0, 0>(s, d_rows, d_ctr, d_out, d_st)) return 1;  // (expected WRONG across XCDs' L2s unless a part stays on its XCD: rows in L2, loads bypass it)
    if (run<0, 1, 0>(s, d_rows, d_ctr, d_out, d_st)) return 1;
    if (run<0, 0, 1>(s, d_rows, d_ctr, d_out, d_st)) return 1;
    if (run<1, 1, 1>(s, d_rows, d_ctr, d_out, d_st)) return 1;  // everything stops at the XCD's L2
    if (run<0, 1, 1>(s, d_rows, d_ctr, d_out, d_st)) return 1;
    if (run<1, 0, 1>(s, d_rows, d_ctr, d_out, d_st)) return 1;
  }
  return 0;
}
