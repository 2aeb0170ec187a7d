// gp_peer.hip -- the exchange step of the one-process-per-GPU form as DIRECT stores over xGMI (round 5).
//
// BASELINE.json's north_star ends a sharded linearise with "an all-reduce of the stacked 6 x 6 H blocks": every rank needs every rank's records.  For the
// headline's shape -- one factor per GPU, 8 x 976 B -- a collective library's ring or tree is all latency: RCCL's all-gather of that size is 30-50 us behind a
// 21 us step.  xGMI is point to point: every GPU can store straight into every other GPU's memory.  So each rank keeps a small
// buffer [generation][rank][row] that its peers have mapped (hipIpcGetMemHandle / hipIpcOpenMemHandle; the handles travel once, through whatever the processes
// already share -- torch.distributed in gtsam_points_amd/distributed.py), and ONE kernel per step and rank
//   1. stores the rank's rows into its slot of EVERY peer's buffer                      (system-scope stores; world - 1 links used at once),
//   2. releases them and stores the step's sequence number into its arrival word at every peer,
//   3. waits until every peer's arrival word in its OWN buffer carries the sequence number  (bounded: a peer that never arrives raises an error, it does not hang the job),
//   4. hands the complete stack to the host (pinned memory) with the sequence number behind it.
// Two generations alternate: a rank can be at most one exchange ahead of a peer (it needs the peer's arrival word of step s + 1 to get past s + 1, and the peer
// sends that only after it has finished step s), so the rows of step s + 2 never land in a buffer somebody still reads.
// The buffers are fine-grained / uncached device memory: lines a REMOTE agent writes must not sit stale in the owner's L2.
// No reference counterpart (the reference has no multi-GPU code); the loop being sharded is cuda/nonlinear_factor_set_gpu.cpp:64-139.
#include <cstring>
#include <vector>

#include "gp_host.hpp"

namespace gp {
constexpr int kPeerMaxWorld = 16;
constexpr int kPeerMaxRowDoubles = 8192;        // per rank and step: small, latency-bound exchanges (larger ones are bandwidth: the collective library's job)
constexpr unsigned long long kPeerWaitTicksDefault = 200000000ull;  // 2 s on the 100 MHz clock: a peer that has not arrived by then is not coming (gp_peer_exchange_set_timeout_ms)
constexpr unsigned long long kPeerPoison = ~0ull;                   // arrival word of a rank that gave up: its peers fail in the SAME step instead of the next one

struct PeerView {
  int world, rank, row_doubles;
  unsigned long long seq, wait_ticks;
  double* rows[kPeerMaxWorld];              // [world][row_doubles] of this step's generation: [rank] = own buffer, the others = the peers' buffers (mapped)
  unsigned long long* arrived[kPeerMaxWorld];  // [world] arrival words of this step's generation, same indexing
  unsigned long long* arrived_other[kPeerMaxWorld];  // ... and of the other generation (only written when this rank gives up)
  double* host_out;                         // pinned [world][row_doubles], may be null
  unsigned long long* host_done;            // pinned: the sequence number when the stack is complete; ~0 when a peer did not arrive
};

__global__ void __launch_bounds__(256) peer_exchange_kernel(const PeerView v) {
  __shared__ int failed;
  const int n = v.row_doubles, t = threadIdx.x;
  if (t == 0) failed = 0;
  double* own = v.rows[v.rank];
  // 1. this rank's rows (written by the kernels in front of this one on the stream) into every peer's buffer
  for (int i = t; i < n; i += 256) {
    const double x = own[(size_t)v.rank * n + i];
    for (int p = 0; p < v.world; p++)
      if (p != v.rank) __hip_atomic_store(v.rows[p] + (size_t)v.rank * n + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (system scope: every lane's stores are out before the arrival words)
  __syncthreads();
  // 2. arrival words, 3. wait for the peers'
  if (t < v.world && t != v.rank) {
    __hip_atomic_store(v.arrived[t] + v.rank, v.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const unsigned long long a = __hip_atomic_load(v.arrived[v.rank] + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a == v.seq) break;
      if (a == kPeerPoison || __builtin_amdgcn_s_memrealtime() - t0 > v.wait_ticks) {  // the peer gave up, or never came
        failed = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  // a rank that gives up says so in every peer's arrival word (its own word of this step is already out, so without this the peers would complete the step and
  // fail only in the next one); the words stay poisoned: the exchange is broken for good and every later step fails at once on every rank
  if (failed && t < v.world && t != v.rank) {
    __hip_atomic_store(v.arrived[t] + v.rank, kPeerPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(v.arrived_other[t]) + v.rank, kPeerPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 4. the stack to the host
  if (v.host_out && !failed)
    for (int i = t; i < v.world * n; i += 256) v.host_out[i] = __hip_atomic_load(own + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (t == 0) __hip_atomic_store(v.host_done, failed ? ~0ull : v.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace gp

struct gp_peer_exchange {
  int world = 0, rank = 0, row_doubles = 0, device = 0;
  unsigned long long seq = 0, wait_ticks = gp::kPeerWaitTicksDefault;
  void* own = nullptr;                       // [2][world][row_doubles] f64, then [2][kPeerMaxWorld] arrival words
  void* peer[gp::kPeerMaxWorld] = {};        // the peers' `own`, mapped into this process (peer[rank] = own)
  bool opened[gp::kPeerMaxWorld] = {};
  const double* validated_host = nullptr;    // the last host_out that was found to be pinned host memory
  gp::PinnedArray done;                      // [1] sequence number of the last finished exchange
  size_t rows_bytes() const { return sizeof(double) * 2 * (size_t)world * (size_t)row_doubles; }
  size_t total_bytes() const { return rows_bytes() + sizeof(unsigned long long) * 2 * gp::kPeerMaxWorld; }
  double* rows_of(void* base, int gen) const { return reinterpret_cast<double*>(base) + (size_t)gen * world * row_doubles; }
  unsigned long long* arrived_of(void* base, int gen) const {
    return reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(base) + rows_bytes()) + (size_t)gen * gp::kPeerMaxWorld;
  }
  ~gp_peer_exchange() {  // (every error path of create / connect ends here: nothing is leaked)
    for (int p = 0; p < world; p++)
      if (opened[p] && peer[p]) (void)hipIpcCloseMemHandle(peer[p]);
    if (own) (void)hipFree(own);
    (void)hipGetLastError();
  }
};

extern "C" {

int gp_peer_exchange_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int gp_peer_exchange_create(int world, int rank, int row_doubles, gp_peer_exchange_t** out, void* handle_out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_create: null out");
  *out = nullptr;
  if (world < 1 || world > gp::kPeerMaxWorld || rank < 0 || rank >= world || row_doubles < 1 || row_doubles > gp::kPeerMaxRowDoubles || !handle_out)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_create: 1 <= world <= 16, 0 <= rank < world, 1 <= row_doubles <= 8192, handle_out = gp_peer_exchange_handle_bytes() bytes");
  auto px = std::make_unique<gp_peer_exchange>();
  px->world = world, px->rank = rank, px->row_doubles = row_doubles;
  GP_HIP(hipGetDevice(&px->device));
  // memory a remote agent writes while the owner polls it: uncached (every access goes to memory), else fine-grained; plain device memory would leave the owner's L2 stale
  const size_t bytes = px->total_bytes();
  if (hipExtMallocWithFlags(&px->own, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    px->own = nullptr;
    if (hipExtMallocWithFlags(&px->own, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      px->own = nullptr;
      return gp::fail(GP_ERROR_HIP, "gp_peer_exchange_create: no uncached / fine-grained device memory for the exchange buffers");
    }
  }
  GP_HIP(hipMemset(px->own, 0, bytes));
  GP_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, px->own) != hipSuccess) {
    (void)hipGetLastError();
    return gp::fail(GP_ERROR_HIP, "gp_peer_exchange_create: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this stack)");
  }
  memcpy(handle_out, &h, sizeof(h));
  GP_TRY(px->done.ensure(64));
  *px->done.as<unsigned long long>() = 0;
  px->peer[rank] = px->own;
  *out = px.release();
  return GP_OK;
}

int gp_peer_exchange_connect(gp_peer_exchange_t* px, const void* handles) {
  if (!px || !handles) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_connect: null");
  for (int p = 0; p < px->world; p++) {
    if (p == px->rank || px->opened[p]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, reinterpret_cast<const char*>(handles) + (size_t)p * sizeof(h), sizeof(h));
    void* ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !ptr) {
      (void)hipGetLastError();
      return gp::fail(GP_ERROR_HIP, "gp_peer_exchange_connect: hipIpcOpenMemHandle failed for a peer's buffer");
    }
    px->peer[p] = ptr;
    px->opened[p] = true;
  }
  return GP_OK;
}

// The step's sequence number advances in gp_peer_exchange_finish, when the exchange kernel has been launched -- not here (a caller whose own kernels fail
// between begin and finish must not leave this rank one generation ahead of its peers for good).  begin only names the generation the NEXT exchange will use.
int gp_peer_exchange_begin(gp_peer_exchange_t* px) {
  if (!px) return -1;
  return (int)((px->seq + 1) & 1ull);
}

int gp_peer_exchange_set_timeout_ms(gp_peer_exchange_t* px, double ms) {
  if (!px || !(ms > 0.0) || ms > 3.6e6) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_set_timeout_ms: 0 < ms <= 3.6e6");
  px->wait_ticks = (unsigned long long)(ms * 1e5);  // 100 MHz constant clock
  return GP_OK;
}

int gp_peer_exchange_finish(gp_peer_exchange_t* px, gp_stream_t stream, double* host_out_pinned) {
  if (!px) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_finish: null");
  for (int p = 0; p < px->world; p++)
    if (!px->peer[p]) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_finish: connect the peers first");
  if (host_out_pinned && host_out_pinned != px->validated_host) {  // the kernel stores world x row_doubles doubles through this pointer: it must be host memory the device can address (pinned / registered)
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, host_out_pinned) != hipSuccess || attr.type != hipMemoryTypeHost) {
      (void)hipGetLastError();
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_finish: host_out must be pinned host memory of world x row_doubles doubles");
    }
    px->validated_host = host_out_pinned;  // (asked once per buffer: a step is tens of microseconds)
  }
  gp::PeerView v{};
  const unsigned long long seq = px->seq + 1;
  v.world = px->world, v.rank = px->rank, v.row_doubles = px->row_doubles, v.seq = seq, v.wait_ticks = px->wait_ticks;
  const int gen = (int)(seq & 1ull);
  for (int p = 0; p < px->world; p++) {
    v.rows[p] = px->rows_of(px->peer[p], gen);
    v.arrived[p] = px->arrived_of(px->peer[p], gen);
    v.arrived_other[p] = px->arrived_of(px->peer[p], gen ^ 1);
  }
  v.host_out = host_out_pinned;
  v.host_done = px->done.as<unsigned long long>();
  hipLaunchKernelGGL(gp::peer_exchange_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v);
  GP_HIP(hipGetLastError());
  px->seq = seq;  // (launched: the step exists)
  return GP_OK;
}

int gp_peer_exchange_check(const gp_peer_exchange_t* px) {
  if (!px) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_peer_exchange_check: null");
  const unsigned long long d = *reinterpret_cast<const volatile unsigned long long*>(px->done.ptr);
  if (d == ~0ull) return gp::fail(GP_ERROR_HIP, "gp_peer_exchange: a peer did not arrive within the time box");
  if (d != px->seq) return gp::fail(GP_ERROR_HIP, "gp_peer_exchange: the last exchange has not finished (synchronise the stream first)");
  return GP_OK;
}

void* gp_peer_exchange_rows(gp_peer_exchange_t* px, int generation) {
  if (!px || generation < 0 || generation > 1) return nullptr;
  return px->rows_of(px->own, generation);
}

int gp_peer_exchange_destroy(gp_peer_exchange_t* px) {
  if (!px) return GP_OK;
  (void)hipDeviceSynchronize();
  delete px;  // (the destructor unmaps the peers and frees the buffer)
  return GP_OK;
}

}  // extern "C"
