// gp_host.hpp -- host-side internals shared by the translation units of libgtsam_points_hip.so
#pragma once
#include <algorithm>

#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <chrono>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "gp_device.hpp"

struct gp_vgicp_batch;
namespace gp {
// gp_vgicp.hip: sibling batches on the same device re-read this batch's source clouds (gp_multi.hip, several shards per device)
void batch_set_sources_shared(gp_vgicp_batch* batch, bool shared);
}  // namespace gp

namespace gp {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t err, const char* expr, const char* file, int line);

#define GP_HIP(expr)                                                      \
  do {                                                                    \
    hipError_t gp_err__ = (expr);                                         \
    if (gp_err__ != hipSuccess) return gp::hip_fail(gp_err__, #expr, __FILE__, __LINE__); \
  } while (0)

#define GP_TRY(expr)                 \
  do {                               \
    int gp_rc__ = (expr);            \
    if (gp_rc__ != GP_OK) return gp_rc__; \
  } while (0)

// Per-thread cache of stream-ordered blocks.  hipMallocAsync / hipFreeAsync cost 20-50 us apiece on this stack even when the pool
// holds the memory, and the structure builds (bin_points, voxel-map insert, k-NN grid) take ~15 scratch arrays each: the allocator
// calls were 1-2 ms of a 3 ms covariance estimation.  A released block is parked here, tagged with the stream in whose order it
// was released (kSyncedRelease: the owner synchronised the work that used it first, so any stream may take it), and handed to the next
// request of the same device and stream whose size it fits (<= 2x).  A block released in the order of the NULL stream is only handed
// back to requests on the NULL stream: non-blocking streams are not ordered behind it.  Bounded: kMaxEntries blocks / kMaxBytes; beyond that the oldest blocks go back to
// the pool.  gp_trim_device_cache() empties it.
struct BlockCache {
  struct Entry {
    void* ptr;
    size_t bytes;
    hipStream_t stream;
    int device;
    unsigned long long age;  // value of `clock` when the block was parked
  };
  unsigned long long clock = 0;
  static hipStream_t synced_release() { return reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(1)); }  // tag, never a real stream
  static constexpr size_t kMaxEntries = 96;
  static constexpr size_t kMaxBytes = size_t(4) << 30;
  std::vector<Entry> entries;
  size_t total = 0;
  static BlockCache& get() {
    static thread_local BlockCache c;
    return c;
  }
  void* take(size_t n, hipStream_t stream, int device, size_t* got) {
    int best = -1;
    for (int i = 0; i < (int)entries.size(); i++) {
      const Entry& e = entries[i];
      if (e.device != device || (e.stream != synced_release() && e.stream != stream) || e.bytes < n || e.bytes > 2 * n + 4096) continue;
      if (best < 0 || e.bytes < entries[best].bytes) best = i;
    }
    if (best < 0) return nullptr;
    void* p = entries[best].ptr;
    *got = entries[best].bytes;
    total -= entries[best].bytes;
    entries[best] = entries.back();
    entries.pop_back();
    return p;
  }
  // A full cache makes room by giving its OLDEST blocks of this device back to the pool (round 4: it used to refuse the new block instead -- after a phase that
  // parked many blocks of other sizes, every later call paid hipFreeAsync + hipMallocAsync for all of its scratch arrays, up to milliseconds per call:
  // bench.py's C5 behind C3's 64 map builds).  The stream an old block is tagged with may have been destroyed since: the blocks are freed on the NULL stream
  // behind ONE synchronisation of the device, a quarter of the cache at a time, so that a change of phase pays it once and not per block.
  bool put(void* p, size_t bytes, hipStream_t stream, int device) {
    if (bytes > kMaxBytes / 2) return false;
    if (entries.size() >= kMaxEntries || total + bytes > kMaxBytes) {
      int cur = -1;
      if (hipGetDevice(&cur) != hipSuccess || cur != device) return false;  // (an array of another device is being released: it goes back to the pool directly)
      std::vector<int> mine;
      for (int i = 0; i < (int)entries.size(); i++)
        if (entries[i].device == device) mine.push_back(i);
      if (mine.empty()) return false;  // (full of other devices' blocks)
      std::sort(mine.begin(), mine.end(), [&](int a, int b) { return entries[a].age < entries[b].age; });
      size_t evict = std::max<size_t>(kMaxEntries / 4, 1), freed = 0;
      (void)hipDeviceSynchronize();
      std::vector<bool> gone(entries.size(), false);
      for (size_t j = 0; j < mine.size() && (j < evict || total - freed + bytes > kMaxBytes); j++) {
        (void)hipFreeAsync(entries[mine[j]].ptr, nullptr);
        freed += entries[mine[j]].bytes;
        gone[mine[j]] = true;
      }
      std::vector<Entry> kept;
      for (size_t i = 0; i < entries.size(); i++)
        if (!gone[i]) kept.push_back(entries[i]);
      entries.swap(kept);
      total -= freed;
      if (entries.size() >= kMaxEntries || total + bytes > kMaxBytes) return false;
    }
    entries.push_back({p, bytes, stream, device, ++clock});
    total += bytes;
    return true;
  }
  void trim() {
    if (entries.empty()) return;
    // the tagged streams may be gone by now: every block is returned on the NULL stream of ITS device, behind a synchronisation of that device
    // (a single-process multi-GPU batch parks blocks of several devices in one thread's cache)
    int cur = 0;
    (void)hipGetDevice(&cur);
    std::vector<int> devices;
    for (const Entry& e : entries)
      if (std::find(devices.begin(), devices.end(), e.device) == devices.end()) devices.push_back(e.device);
    for (int dev : devices) {
      if (hipSetDevice(dev) != hipSuccess) continue;
      (void)hipDeviceSynchronize();
      for (const Entry& e : entries)
        if (e.device == dev) (void)hipFreeAsync(e.ptr, nullptr);
    }
    (void)hipSetDevice(cur);
    entries.clear();
    total = 0;
  }
  ~BlockCache() {}  // at thread exit the blocks stay with the pool's owner (the process is usually going down; the runtime may be gone)
};

// RAII device allocation used for library-owned arrays
struct DeviceArray {
  void* ptr = nullptr;
  size_t bytes = 0;
  DeviceArray() = default;
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  ~DeviceArray() { release(); }
  int alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&ptr, n);
    if (e != hipSuccess) {
      ptr = nullptr;
      return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    }
    bytes = n;
    return GP_OK;
  }
  int ensure(size_t n) { return (n <= bytes && ptr) ? GP_OK : alloc(n + n / 5); }
  void swap(DeviceArray& o) {
    std::swap(ptr, o.ptr);
    std::swap(bytes, o.bytes);
    std::swap(pooled, o.pooled);
    std::swap(pool_stream, o.pool_stream);
    std::swap(device, o.device);
  }
  // stream-ordered allocation from the device's default memory pool (cudaMallocAsync upstream, cuda/cuda_malloc_async.hpp):
  // for short-lived scratch -- a pooled block is reused by the next call instead of going through hipMalloc / hipFree, which
  // cost more than the kernels they serve.  The block may only be used by work ordered after this call on `stream`.
  int alloc_async(size_t n, hipStream_t stream) {
    release();
    if (n == 0) n = 16;
    keep_pool_memory();
    (void)hipGetDevice(&device);
    size_t got = 0;
    if (void* cached = BlockCache::get().take(n, stream, device, &got)) {
      ptr = cached;
      bytes = got;
    } else {
      hipError_t e = hipMallocAsync(&ptr, n, stream);
      if (e != hipSuccess) {  // out of memory while up to 4 GiB sit parked in this thread's cache: give them back and try once more
        (void)hipGetLastError();
        BlockCache::get().trim();
        e = hipMallocAsync(&ptr, n, stream);
      }
      if (e != hipSuccess) {
        ptr = nullptr;
        return hip_fail(e, "hipMallocAsync", __FILE__, __LINE__);
      }
      bytes = n;
    }
    pooled = true;
    pool_stream = stream;
    return GP_OK;
  }
  // library-owned arrays that outlive the call (voxel maps, bins): pooled like the reference's cudaMallocAsync'ed members, but
  // not released in the order of the creating stream -- it is the caller's and may be gone by then; owners synchronise the work
  // that used such arrays before they let go of them (gp_voxelmap_destroy synchronises the device), so the block may be re-used on
  // any stream
  int alloc_pooled(size_t n, hipStream_t stream) {
    const int rc = alloc_async(n, stream);
    pool_stream = BlockCache::synced_release();
    return rc;
  }
  // for owners that know every use of the array was ordered on `stream`: return it to the pool in that stream's order
  void release_on(hipStream_t stream) {
    if (pooled) pool_stream = stream;
    release();
  }
  int ensure_pooled(size_t n, hipStream_t stream) { return (n <= bytes && ptr) ? GP_OK : alloc_pooled(n + n / 5, stream); }
  void release() {
    if (ptr) {
      if (pooled) {
        if (!BlockCache::get().put(ptr, bytes, pool_stream, device)) (void)hipFreeAsync(ptr, pool_stream == BlockCache::synced_release() ? nullptr : pool_stream);
      } else {
        (void)hipFree(ptr);
      }
    }
    ptr = nullptr;
    bytes = 0;
    pooled = false;
  }
  bool pooled = false;
  hipStream_t pool_stream = nullptr;
  int device = 0;
  // the default pool hands memory back to the driver at every synchronisation unless told to keep it
  static void keep_pool_memory() {
    static thread_local int configured_device = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev == configured_device) return;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
      uint64_t threshold = ~0ull;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
    }
    configured_device = dev;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(ptr);
  }
};

struct PinnedArray {
  void* ptr = nullptr;
  size_t bytes = 0;
  PinnedArray() = default;
  PinnedArray(const PinnedArray&) = delete;
  PinnedArray& operator=(const PinnedArray&) = delete;
  ~PinnedArray() { release(); }
  int ensure(size_t n) {
    if (n <= bytes && ptr) return GP_OK;
    release();
    if (n == 0) n = 16;
    n += n / 5;
    hipError_t e = hipHostMalloc(&ptr, n, hipHostMallocDefault);
    if (e != hipSuccess) {
      ptr = nullptr;
      return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    bytes = n;
    return GP_OK;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(ptr);
  }
};

// Packed private mirror of a source cloud (round 4; gp_cloud.hip).  The VGICP stream kernel uses only the symmetric part of a source covariance, so
// the API layout's 12 + 36 B per point (types/point_cloud.hpp:114-118, read by include/gtsam_points/cuda/kernels/vgicp_derivatives.cuh:36-50) is
// repacked once per cloud into 36 B per point, chunk-major: per 64 points 2304 contiguous bytes = 64 x (x, y, z) | 64 x (c00, c01, c02) | 64 x (c11, c12, c22)
// -- three 768-byte rows, each ONE 12-B-per-lane LDS-DMA instruction.  A mirror is built only when every covariance is symmetric to the last bit
// (estimate_covariances' output is), so the six floats are the caller's own and records are bit-identical to the unpacked stream; otherwise
// `usable` is false and the kernels keep the caller's arrays (which carry the (a_ij + a_ji) / 2 symmetrisation in f64).
// Shared by every factor that reads the same (points, covs, n) -- a submap that is the source of eight factors has ONE mirror, and its re-reads hit L2 --
// through a registry of weak references: the mirror lives as long as some factor holds it.  The caller's arrays are immutable while a factor borrows
// them (the reference holds them through PointCloud::ConstPtr); an owner that rewrites or frees them calls gp_source_mirror_invalidate.
struct SourceMirror {
  DeviceArray data;
  const float* points = nullptr;
  const float* covs = nullptr;
  int n = 0;
  int device = 0;
  bool usable = false;  // false: some covariance is not bit-symmetric (or non-finite): the kernels stream the caller's arrays
  ~SourceMirror();
};
constexpr int kMirrorChunkBytes = 2304;
// *out: the shared mirror of (points, covs, n) on `device`, packed on `stream` (synchronised before return) when it does not exist yet; null when n < 64
int acquire_source_mirror(const float* points, const float* covs, int n, int device, hipStream_t stream, std::shared_ptr<SourceMirror>* out);

// A few words of host-mapped pinned memory per host thread and device: where a structure build's kernels leave the counts the host sizes the next step by (bounding
// box, number of cells, failed insertions).  Reading them is a load behind the stream's synchronisation -- a D2H copy of device words is a copy KERNEL plus its launch
// (~7 us apiece, six per map build: profiles/r04_map_build_stats.txt).  Never freed (the runtime may be gone when a thread ends).
// A fill as a side job of a kernel that runs anyway (round 4): every kernel on this device costs ~8 us around its workgroups (dispatch to the first instruction, and
// the write-back of its dirty L2 lines before the next kernel may start on another XCD), so a hipMemsetAsync in front of a kernel is that much for a few
// microseconds of stores.  Called by every thread of a kernel whose successors -- not the kernel itself -- read the filled memory; 16-byte granules.
struct FillJob {
  uint4* ptr = nullptr;       // 16-byte aligned
  unsigned long long count = 0;  // granules
  unsigned value = 0;         // every 32-bit word
};
inline FillJob fill_job(void* p, size_t bytes, unsigned value) {  // bytes: a multiple of 16
  FillJob f;
  f.ptr = static_cast<uint4*>(p);
  f.count = bytes / 16;
  f.value = value;
  return f;
}
__device__ __forceinline__ void run_fill_job(const FillJob& f) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const uint4 v = make_uint4(f.value, f.value, f.value, f.value);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < f.count; i += stride) f.ptr[i] = v;
}

// the last "kernel" of a step whose real last kernel has no single finishing thread: stores (optionally a device word into a host word, then) the sequence number into the
// flag word.  In stream order behind the step's kernels, so the flag also means that they have finished.
template <int UNUSED = 0>
__global__ void host_flag_kernel(int* __restrict__ flag, int seq, const int* __restrict__ copy_src, int* __restrict__ copy_dst) {
  if (copy_src) {
    *copy_dst = *copy_src;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  *flag = seq;
}

struct HostWords {
  int* host = nullptr;
  int* dev = nullptr;
  int* seq = nullptr;  // this thread's sequence counter for word kFlag (below)
  static constexpr int kWords = 64;
  // Word kFlag is a completion flag: the host draws a fresh sequence number (next_seq), hands it to the LAST kernel of a step, and that kernel stores it behind its
  // result words (stores to host memory, `s_waitcnt vmcnt(0)` between the results and the flag).  wait_flag polls the word -- the host sees it ~1 us behind the store,
  // where hipStreamSynchronize adds the runtime's wake-up (5-10 us) -- and falls back to the synchronisation after 500 us (a failed kernel never stores).
  // The flag says "the results are there", NOT "the kernel is finished": only results may be read behind it; everything else stays ordered by the stream.
  static constexpr int kFlag = 15;
  int next_seq() const { return ++*seq; }
  // "everything issued on `s` so far has finished" without hipStreamSynchronize's wake-up: a one-thread kernel behind it stores the flag (and copies a device int into host word
  // `copy_to_word` first, when asked to); the host polls
  int finish(hipStream_t s, const int* copy_src_dev = nullptr, int copy_to_word = 0) const {
    const int q = next_seq();
    hipLaunchKernelGGL(host_flag_kernel<0>, dim3(1), dim3(1), 0, s, dev + kFlag, q, copy_src_dev, dev + copy_to_word);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "host_flag_kernel", __FILE__, __LINE__);
    return wait_flag(q, s);
  }
  int wait_flag(int expect, hipStream_t s) const {
    const volatile int* f = host + kFlag;
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0; *f != expect; spins++) {
      if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
        break;
      }
    }
    return GP_OK;
  }
  static int get(HostWords* out) {
    static thread_local HostWords w[16];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) return fail(GP_ERROR_HIP, "HostWords: no current device");
    if (!w[d].host) {
      void* p = nullptr;
      hipError_t e = hipHostMalloc(&p, sizeof(int) * kWords, hipHostMallocMapped);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
      void* dp = nullptr;
      e = hipHostGetDevicePointer(&dp, p, 0);
      if (e != hipSuccess) return hip_fail(e, "hipHostGetDevicePointer", __FILE__, __LINE__);
      w[d].host = static_cast<int*>(p);
      w[d].dev = static_cast<int*>(dp);
      static thread_local int seqs[16];
      w[d].seq = &seqs[d];
      for (int i = 0; i < kWords; i++) w[d].host[i] = 0;
    }
    *out = w[d];
    return GP_OK;
  }
};

// Per-thread, per-device host-mapped slots the workgroups of a kernel leave small records in for the HOST to combine (the bounding box of a structure build: one
// 32-byte record per workgroup, the last word a sequence number stored behind the others) -- no reducing kernel, no ticket, no atomics, and the host has the result
// ~1 us after the last record lands instead of a kernel boundary + a reduce kernel later.  kSlots records of eight ints; never freed.
struct HostSlots {
  int* host = nullptr;
  int* dev = nullptr;
  static constexpr int kSlots = 2048;
  static int get(HostSlots* out) {
    static thread_local HostSlots w[16];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) return fail(GP_ERROR_HIP, "HostSlots: no current device");
    if (!w[d].host) {
      void* p = nullptr;
      hipError_t e = hipHostMalloc(&p, sizeof(int) * 8 * kSlots, hipHostMallocMapped);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
      void* dp = nullptr;
      e = hipHostGetDevicePointer(&dp, p, 0);
      if (e != hipSuccess) return hip_fail(e, "hipHostGetDevicePointer", __FILE__, __LINE__);
      w[d].host = static_cast<int*>(p);
      w[d].dev = static_cast<int*>(dp);
      for (int i = 0; i < 8 * kSlots; i++) w[d].host[i] = 0;
    }
    *out = w[d];
    return GP_OK;
  }
};

void release_side_streams();  // gp_knn.hip (SideStream): what gp_trim_device_cache releases beside the parked blocks
}  // namespace gp

// TempBufferManager (cuda/stream_temp_buffer_roundrobin.cu:11-47)
struct gp_temp_buffer {
  struct Buffer {
    size_t size = 0;
    char* buffer = nullptr;
  };
  std::vector<Buffer> buffers;
};

// GaussianVoxelMapGPU device state
struct gp_voxelmap {
  double resolution = 0.0;
  int init_num_buckets = 16384;
  double target_points_drop_rate = 1e-3;
  hipStream_t stream = nullptr;
  gp_voxelmap_info info{};

  gp::DeviceArray buckets;      // gp_voxel_bucket[num_buckets]
  gp::DeviceArray records;      // gp::VoxelRecord[num_voxels]   (kernel gather layout)
  gp::DeviceArray num_points;   // int[num_voxels]               (reference-visible arrays below)
  gp::DeviceArray voxel_means;  // float[num_voxels][3]
  gp::DeviceArray voxel_covs;   // float[num_voxels][9]
  gp::DeviceArray voxel_intensities;  // float[num_voxels]
  gp::DeviceArray voxel_coords; // int[num_voxels][3] voxel coordinate of each voxel index
  gp::DeviceArray plines;  // line table of the hashed VGICP kernel family (gp::VoxelMapView::plines): built on first use when the map has its block grid (round 4: its
                           // 16 MB fill and its kernel were 12 us of every map build, for a table no kernel of the default path reads)
  uint32_t plmask = 0;
  bool private_built = false;
  std::mutex private_mutex;    // (batches of several threads may ask at once)
  int ensure_private_table();  // (gp_voxelmap.hip) builds plines on the map's stream and synchronises it; no-op when built
  gp::DeviceArray gblocks;  // occupancy-block grid (gp::GridBlock[gdim0 * gdim1 * gdim2]); empty when the box is too large
  int glo[3] = {0, 0, 0}, gdim[3] = {0, 0, 0};
  bool has_grid = false;
  bool force_hashed_build = false;  // gp_voxelmap_set_tuning(GP_TUNE_MAP_BUILD): the next insert() uses the hashed build
  int bucket_load_percent = 33;     // GP_TUNE_BUCKET_LOAD: the binned build enters the reference's doubling sequence at the first size that holds the voxels at this load factor

  // offloaded copies (OffloadableGPU)
  bool offloaded = false;
  uint64_t generation = 0;  // bumped whenever the device arrays are (re)allocated: factor tables built from view() go stale
  std::vector<char> h_buckets, h_records, h_num_points, h_means, h_covs, h_intensities, h_coords, h_gblocks;

  bool loaded() const { return buckets.ptr != nullptr && !offloaded; }
  gp::VoxelMapView view() const;
};
