// aql_launch_probe.hip -- what does the HIP launch path cost in front of a kernel whose result the host polls in mapped memory?
//
// The shape of the synchronous VGICP step: ONE launch of 1024 workgroups whose last workgroups store completion words into host-mapped memory,
// the host spinning on the words.  Two ways to get the launch onto the device, host to host:
//   (a) hipLaunchKernelGGL on a HIP stream (what the library does);
//   (b) an AQL kernel-dispatch packet written by this thread into a user-mode queue of its own (hsa_queue_create), doorbell rung directly:
//       no runtime command object, no kernarg pool bookkeeping, no completion signal.
// The kernel is the same code object in both cases (loaded a second time through the HSA loader for (b)).  Body: `spin` microseconds of s_sleep per workgroup.
// Build: hipcc --offload-arch=gfx950 -O2 -o aql_launch_probe aql_launch_probe.hip -lhsa-runtime64
//        hipcc --offload-arch=gfx950 -O2 --genco -o aql_launch_probe.co aql_launch_probe.hip
// Run:   ./aql_launch_probe aql_launch_probe.co
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e__ = (x);                                                      \
    if (e__ != hipSuccess) {                                                   \
      fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e__), __LINE__); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)
#define HK(x)                                                          \
  do {                                                                 \
    hsa_status_t s__ = (x);                                            \
    if (s__ != HSA_STATUS_SUCCESS && s__ != HSA_STATUS_INFO_BREAK) {   \
      const char* m__ = nullptr;                                       \
      hsa_status_string(s__, &m__);                                    \
      fprintf(stderr, "%s -> %s (line %d)\n", #x, m__ ? m__ : "?", __LINE__); \
      exit(3);                                                         \
    }                                                                  \
  } while (0)

struct Args {
  unsigned long long* flags;  // host-mapped, 8 words
  unsigned long long* count;  // device counter (monotonic)
  unsigned long long seq;
  unsigned long long target;  // what *count reads when every workgroup of this launch has arrived
  int spin_cycles;
  int pad;
};

// every workgroup idles `spin_cycles`, arrives; the last one stores the 8 completion words
extern "C" __global__ void __launch_bounds__(256) probe_kernel(Args a) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.spin_cycles) __builtin_amdgcn_s_sleep(2);
    const unsigned long long seen = __hip_atomic_fetch_add(a.count + (blockIdx.x & 7) * 512, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen + 1 == a.target) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(a.flags + (blockIdx.x & 7)), "v"(a.seq) : "memory");
  }
}

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_kernarg_pool;
static bool g_have_gpu = false, g_have_cpu = false, g_have_pool = false;

static hsa_status_t agent_cb(hsa_agent_t agent, void*) {
  hsa_device_type_t type;
  hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type);
  if (type == HSA_DEVICE_TYPE_GPU && !g_have_gpu) {
    g_gpu = agent;
    g_have_gpu = true;
  }
  if (type == HSA_DEVICE_TYPE_CPU && !g_have_cpu) {
    g_cpu = agent;
    g_have_cpu = true;
  }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) {
    g_kernarg_pool = pool;
    g_have_pool = true;
  }
  return HSA_STATUS_SUCCESS;
}

static double median(std::vector<double>& v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <code object file>\n", argv[0]);
    return 1;
  }
  const int wgs = 1024, iters = 1000;
  const int variant = argc > 2 ? atoi(argv[2]) : 0;  // bit 0: agent-scope fences in the packet header, bit 1: HSA_QUEUE_TYPE_MULTI, bit 2: high queue priority, bit 3: a completion signal, bit 4: doorbell value = index + 1, bit 5: 16384-entry queue
  printf("variant %d: %s fences, %s queue, %s priority, %s completion signal\n", variant, variant & 1 ? "agent-scope" : "system-scope", variant & 2 ? "multi-producer" : "single-producer",
         variant & 4 ? "high" : "normal", variant & 8 ? "with" : "no");
  CK(hipSetDevice(0));
  hipStream_t stream;
  CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  unsigned long long* flags_host = nullptr;
  CK(hipHostMalloc(&flags_host, 64, hipHostMallocMapped));
  memset(flags_host, 0, 64);
  unsigned long long* flags_dev = nullptr;
  CK(hipHostGetDevicePointer((void**)&flags_dev, flags_host, 0));
  unsigned long long* count = nullptr;
  CK(hipMalloc(&count, 8 * 512 * 8));
  CK(hipMemset(count, 0, 8 * 512 * 8));
  CK(hipDeviceSynchronize());

  // ---- HSA side: a queue of our own, the code object loaded through the HSA loader ----
  HK(hsa_init());
  HK(hsa_iterate_agents(agent_cb, nullptr));
  if (!g_have_gpu || !g_have_cpu) return 4;
  HK(hsa_amd_agent_iterate_memory_pools(g_cpu, pool_cb, nullptr));
  if (!g_have_pool) return 5;
  hsa_queue_t* queue = nullptr;
  HK(hsa_queue_create(g_gpu, variant & 32 ? 16384 : 1024, variant & 2 ? HSA_QUEUE_TYPE_MULTI : HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &queue));
  if (variant & 4) HK(hsa_amd_queue_set_priority(queue, HSA_AMD_QUEUE_PRIORITY_HIGH));
  hsa_signal_t done_signal{0};
  if (variant & 8) HK(hsa_signal_create(1 << 30, 0, nullptr, &done_signal));
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 6;
  fseek(f, 0, SEEK_END);
  const long co_size = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> co(co_size);
  if (fread(co.data(), 1, co_size, f) != (size_t)co_size) return 6;
  fclose(f);
  hsa_code_object_reader_t reader;
  HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &reader));
  hsa_executable_t exe;
  HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  HK(hsa_executable_freeze(exe, nullptr));
  hsa_executable_symbol_t sym;
  HK(hsa_executable_get_symbol_by_name(exe, "probe_kernel.kd", &g_gpu, &sym));
  uint64_t kernel_object = 0;
  uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kernel_object));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kernarg_size));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group_size));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &private_size));
  printf("kernel object %#llx, kernarg %u B (explicit %zu), LDS %u, scratch %u\n", (unsigned long long)kernel_object, kernarg_size, sizeof(Args), group_size, private_size);
  // kernarg ring: one slot per in-flight launch (the host waits for each launch, so two would do)
  const size_t slot = (kernarg_size + 255) & ~size_t(255);
  char* kernarg = nullptr;
  HK(hsa_amd_memory_pool_allocate(g_kernarg_pool, slot * 16, 0, (void**)&kernarg));
  HK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernarg));
  memset(kernarg, 0, slot * 16);

  unsigned long long seq = 0, arrived = 0;
  auto fill_args = [&](Args* a, int spin_us) {
    a->flags = flags_dev;
    a->count = count;
    a->seq = ++seq;
    arrived += wgs / 8;
    a->target = arrived;
    a->spin_cycles = spin_us * 100;  // 100 MHz constant clock
    a->pad = 0;
  };
  auto wait_flags = [&](unsigned long long s) {
    const volatile unsigned long long* fl = flags_host;
    for (;;) {
      bool all = true;
      for (int k = 0; k < 8; k++) all = all && fl[k] == s;
      if (all) return;
    }
  };
  auto hip_launch = [&](int spin_us) {
    Args a;
    fill_args(&a, spin_us);
    hipLaunchKernelGGL(probe_kernel, dim3(wgs), dim3(256), 0, stream, a);
    wait_flags(a.seq);
  };
  uint64_t slot_idx = 0;
  auto aql_launch = [&](int spin_us) {
    char* ka = kernarg + (slot_idx++ % 16) * slot;
    Args* a = reinterpret_cast<Args*>(ka);
    fill_args(a, spin_us);
    // code object v5 implicit arguments behind the explicit ones (8-byte aligned): block counts, group sizes, remainders, ..., global offsets, grid dims
    char* hid = ka + ((sizeof(Args) + 7) & ~size_t(7));
    if (kernarg_size >= ((sizeof(Args) + 7) & ~size_t(7)) + 72) {
      uint32_t* bc = reinterpret_cast<uint32_t*>(hid);
      bc[0] = wgs, bc[1] = 1, bc[2] = 1;
      uint16_t* gs = reinterpret_cast<uint16_t*>(hid + 12);
      gs[0] = 256, gs[1] = 1, gs[2] = 1, gs[3] = 0, gs[4] = 0, gs[5] = 0;
      reinterpret_cast<uint16_t*>(hid + 64)[0] = 1;
    }
    const uint64_t idx = hsa_queue_add_write_index_relaxed(queue, 1);
    hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(queue->base_address) + (idx & (queue->size - 1));
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = 256, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = 256u * wgs, p->grid_size_y = 1, p->grid_size_z = 1;
    p->private_segment_size = private_size;
    p->group_segment_size = group_size;
    p->kernel_object = kernel_object;
    p->kernarg_address = ka;
    p->reserved2 = 0;
    p->completion_signal = done_signal;
    const int scope = variant & 1 ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_SYSTEM;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(queue->doorbell_signal, (hsa_signal_value_t)(idx + (variant & 16 ? 1 : 0)));
    wait_flags(a->seq);
  };

  for (int spin_us : {0, 12}) {
    for (int i = 0; i < 50; i++) hip_launch(spin_us);
    for (int i = 0; i < 50; i++) aql_launch(spin_us);
    std::vector<double> th, ta;
    for (int rep = 0; rep < 2; rep++) {
      for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        hip_launch(spin_us);
        th.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        aql_launch(spin_us);
        ta.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
    }
    printf("body %2d us, %d workgroups, host to host (median of %zu): hipLaunchKernelGGL %.2f us   own AQL queue %.2f us\n", spin_us, wgs, th.size(), median(th), median(ta));
  }
  CK(hipStreamSynchronize(stream));
  hsa_queue_destroy(queue);
  return 0;
}
