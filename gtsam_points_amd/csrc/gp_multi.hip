// gp_multi.hip -- many-factor VGICP batches sharded over the GPUs of one node, driven from ONE process.
//
// The reference has no multi-GPU code: NonlinearFactorSetGPU::linearize (src/gtsam_points/cuda/nonlinear_factor_set_gpu.cpp:64-139)
// walks its factors on one device.  Factors are independent units (one source cloud, one target map, one pose pair in, one
// 122-scalar record out), so the factor list is partitioned into shards, every shard runs the batched kernels of gp_vgicp.hip on
// its own device and stream, and the one exchange step is the all-reduce BASELINE.json's north_star names:
//
//     every shard writes its records into its rows of a zeroed [F x 122] f64 stack on its device
//     ONE ncclAllReduce(sum) per device over that stack (RCCL over xGMI; every row has exactly one writer, so the sum is exact)
//     ONE D2H of the complete stack from the first shard's device
//
// Round 4: the stack has one writer per row and gp_shard_plan deals contiguous ranges, so when the shards are EQUAL contiguous ranges in rank order (C4: 8 x 512
// factors) the same exchange is an in-place ncclAllGather -- (N-1)/N of the stack per device instead of 2 (N-1)/N, and no zeroing ("the equivalent
// cheaper form").  use_rccl = 2 asks for it (falls back to the all-reduce when the plan does not qualify); gp_vgicp_multi_batch_uses_rccl tells which runs.
//
// RCCL is loaded with dlopen when a multi-batch really spans several devices (single-GPU users never load it; its six entry points are declared below, the
// library is not needed to build).  Where it is not used -- all shards on one device (the single-GPU rehearsal of an N-shard plan), a missing library, or
// use_rccl == 0 -- NO collective and no copy runs at all: every shard's finalize kernel stores its records straight into its rows of ONE host-pinned, portable
// stack (the consumer is host-side), and the pass ends with the shards' streams' synchronisation.
//
// Everything here is built on the public per-device entry points (gp_vgicp_batch_*), one host thread issuing to all devices.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <deque>
#include <numeric>
#include <string>

#include "gp_host.hpp"

// ---- shard plan: contiguous partition of the factor list balanced by weight (pure host code) -------------------------------
struct gp_shard_plan {
  std::vector<int> begin;  // num_shards + 1 boundaries
};

namespace {

// contiguous partition of w[0..n) into k ranges minimising the largest range sum (binary search on the bound + greedy fill)
std::vector<int> balanced_boundaries(const int64_t* w, int n, int k) {
  std::vector<int> best((size_t)k + 1, n);
  best[0] = 0;
  if (n == 0) return best;
  int64_t lo = 0, hi = 0;
  for (int i = 0; i < n; i++) {
    lo = std::max(lo, w[i]);
    hi += w[i];
  }
  auto fill = [&](int64_t bound, std::vector<int>* out) {
    int shard = 0;
    int64_t acc = 0;
    std::vector<int> b((size_t)k + 1, n);
    b[0] = 0;
    for (int i = 0; i < n; i++) {
      if (acc + w[i] > bound && acc > 0) {
        shard++;
        if (shard >= k) return false;
        b[(size_t)shard] = i;
        acc = 0;
      }
      acc += w[i];
    }
    if (out) *out = b;
    return true;
  };
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (fill(mid, nullptr)) hi = mid;
    else lo = mid + 1;
  }
  fill(lo, &best);
  // the greedy fill may need fewer than k ranges: split the heaviest range that still holds two factors (at the point that
  // balances its halves) until every shard has work or every factor stands alone -- splitting never raises the maximum
  std::vector<int> cuts;  // begins of the non-empty ranges
  for (int s = 0; s < k; s++)
    if (best[(size_t)s] < best[(size_t)s + 1]) cuts.push_back(best[(size_t)s]);
  auto sum = [&](int a, int b) {
    int64_t t = 0;
    for (int i = a; i < b; i++) t += w[i];
    return t;
  };
  while ((int)cuts.size() < std::min(k, n)) {
    int pick = -1;
    int64_t heaviest = -1;
    for (size_t r = 0; r < cuts.size(); r++) {
      const int a = cuts[r], b = r + 1 < cuts.size() ? cuts[r + 1] : n;
      if (b - a >= 2 && sum(a, b) > heaviest) {
        heaviest = sum(a, b);
        pick = (int)r;
      }
    }
    if (pick < 0) break;
    const int a = cuts[(size_t)pick], b = (size_t)pick + 1 < cuts.size() ? cuts[(size_t)pick + 1] : n;
    int at = a + 1;
    int64_t best_max = -1;
    for (int c = a + 1; c < b; c++) {
      const int64_t m = std::max(sum(a, c), sum(c, b));
      if (best_max < 0 || m < best_max) {
        best_max = m;
        at = c;
      }
    }
    cuts.insert(cuts.begin() + pick + 1, at);
  }
  for (int s = 0; s <= k; s++) best[(size_t)s] = s < (int)cuts.size() ? cuts[(size_t)s] : n;
  return best;
}

}  // namespace

extern "C" {

int gp_shard_plan_create(const int64_t* weights, int num_factors, int num_shards, gp_shard_plan_t** out) {
  if (!out || num_factors < 0 || num_shards <= 0 || (num_factors > 0 && !weights)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_shard_plan_create: bad arguments");
  for (int i = 0; i < num_factors; i++)
    if (weights[i] < 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_shard_plan_create: negative weight");
  auto* p = new gp_shard_plan;
  p->begin = balanced_boundaries(weights, num_factors, num_shards);
  *out = p;
  return GP_OK;
}

int gp_shard_plan_num_shards(const gp_shard_plan_t* plan) { return plan ? (int)plan->begin.size() - 1 : 0; }

int gp_shard_plan_range(const gp_shard_plan_t* plan, int shard, int* begin, int* end) {
  if (!plan || shard < 0 || shard + 1 >= (int)plan->begin.size() || !begin || !end) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_shard_plan_range: bad arguments");
  *begin = plan->begin[(size_t)shard];
  *end = plan->begin[(size_t)shard + 1];
  return GP_OK;
}

int gp_shard_plan_destroy(gp_shard_plan_t* plan) {
  delete plan;
  return GP_OK;
}

}  // extern "C"

// ---- RCCL, loaded on demand ---------------------------------------------------------------------------------------------------
// the part of rccl.h this file uses (nccl.h's public ABI: opaque communicator, result / type / op enumerations), so that building the library does not need RCCL's headers
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;  // ncclFloat64
typedef enum { ncclSum = 0 } ncclRedOp_t;

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) return r;
  auto sym = [&](const char* n) { return dlsym(r.handle, n); };
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
  return r;
}

int nccl_fail(ncclResult_t e, const char* what) { return gp::fail(GP_ERROR_HIP, std::string("RCCL: ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?")); }

#define GP_NCCL(expr)                                      \
  do {                                                     \
    ncclResult_t gp_nccl__ = (expr);                       \
    if (gp_nccl__ != ncclSuccess) return nccl_fail(gp_nccl__, #expr); \
  } while (0)

// out[index[f]][c] = local[f][c]: a shard whose factors are not one contiguous range of the global list
__global__ void __launch_bounds__(256) scatter_rows_kernel(const double* __restrict__ local, const int* __restrict__ index, double* __restrict__ stack, int n, int width) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n * width) return;
  const int f = t / width, c = t % width;
  stack[(size_t)index[f] * width + c] = local[(size_t)f * width + c];
}

constexpr int kRecordDoubles = (int)(sizeof(gp_linearized6) / sizeof(double));  // 122

struct Shard {
  int device = 0;
  hipStream_t stream = nullptr;
  gp_vgicp_batch_t* batch = nullptr;
  std::vector<int> index;  // global indices of this shard's factors, ascending
  bool contiguous = true;
  int64_t points = 0;
  gp::DeviceArray d_stack;   // [F_total x 122] f64 (RCCL) or [F_local x 122] (host gather)
  gp::DeviceArray d_err;     // [F_total] / [F_local] f64
  gp::DeviceArray d_index;   // int[F_local] when !contiguous
  gp::DeviceArray d_local;   // [F_local x 122] staging when !contiguous under RCCL
  std::vector<double> poses, poses_eval;
  hipEvent_t e_begin = nullptr, e_compute = nullptr, e_done = nullptr;
  ncclComm_t comm = nullptr;
};

// In-place all-gather: a shard's send buffer is where its own rows already lie in the stack, i.e. at its first global factor index (the gather plan makes shard k
// exactly rows [k * rows, (k + 1) * rows)). Derived from the plan, never from where the Shard object lives (the shards sit in a std::deque, whose
// elements are not contiguous -- `&s - &shards[0]` is undefined for every shard but the first).
inline size_t gather_send_offset(const std::vector<int>& index, int width) { return index.empty() ? 0 : (size_t)index[0] * (size_t)width; }

// the shards' factor lists from an explicit assignment (ascending global index inside a shard)
inline std::vector<std::vector<int>> shard_index_lists(const int* shard_of, int num_factors, int num_shards) {
  std::vector<std::vector<int>> lists((size_t)num_shards);
  for (int i = 0; i < num_factors; i++) lists[(size_t)shard_of[i]].push_back(i);
  return lists;
}

// all-gather applies iff shard k is exactly rows [k * F / N, (k + 1) * F / N); returns rows per shard or 0
inline size_t gather_rows_of(const std::vector<std::vector<int>>& lists, size_t F) {
  const size_t N = lists.size();
  if (N == 0 || F == 0 || F % N != 0) return 0;
  const size_t rows = F / N;
  for (size_t k = 0; k < N; k++) {
    const auto& ix = lists[k];
    if (ix.size() != rows || (size_t)ix[0] != k * rows) return 0;
    for (size_t j = 1; j < ix.size(); j++)
      if (ix[j] != ix[j - 1] + 1) return 0;
  }
  return rows;
}

struct DeviceGuard {
  int saved = 0;
  DeviceGuard() { (void)hipGetDevice(&saved); }
  ~DeviceGuard() { (void)hipSetDevice(saved); }
};

}  // namespace

struct gp_vgicp_multi_batch {
  std::deque<Shard> shards;  // (a Shard owns device arrays: it never moves)
  int num_factors = 0;
  bool use_rccl = false;
  bool gather = false;      // the exchange is an in-place ncclAllGather (equal contiguous shards in rank order) instead of the all-reduce
  size_t gather_rows = 0;   // ... rows per shard
  void* h_stack = nullptr;  // [F_total x 122] f64, pinned + portable: results land here (written by the finalize kernels themselves when no collective runs)
  void* h_err = nullptr;    // [F_total] f64, likewise
  float last_ms_compute = 0.f, last_ms_exchange = 0.f;
};

namespace {

void destroy_multi(gp_vgicp_multi_batch* mb) {
  if (!mb) return;
  DeviceGuard guard;
  for (auto& s : mb->shards) {
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.comm && rccl().ok) (void)rccl().CommDestroy(s.comm);
    if (s.batch) gp_vgicp_batch_destroy(s.batch);
    s.d_stack.release();
    s.d_err.release();
    s.d_index.release();
    s.d_local.release();
    if (s.e_begin) (void)hipEventDestroy(s.e_begin);
    if (s.e_compute) (void)hipEventDestroy(s.e_compute);
    if (s.e_done) (void)hipEventDestroy(s.e_done);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  if (mb->h_stack) (void)hipHostFree(mb->h_stack);
  if (mb->h_err) (void)hipHostFree(mb->h_err);
  delete mb;
}

// one pass over all shards: WIDTH doubles per factor (122: linearise, 1: error)
template <typename Issue>
int run_pass_impl(gp_vgicp_multi_batch* mb, int width, bool err_pass, Issue issue, double* out_host, bool* group_open);

// A failure in the middle of a pass must not leave an RCCL group open or work in flight on the other devices' streams (the next collective of
// this thread would hang or misbehave): the group is closed and every shard's stream drained before the error is handed up.
template <typename Issue>
int run_pass(gp_vgicp_multi_batch* mb, int width, bool err_pass, Issue issue, double* out_host) {
  bool group_open = false;
  const int rc = run_pass_impl(mb, width, err_pass, issue, out_host, &group_open);
  if (rc != GP_OK) {
    const std::string keep = gp_last_error() ? gp_last_error() : "";
    if (group_open) (void)rccl().GroupEnd();
    for (auto& s : mb->shards) {
      (void)hipSetDevice(s.device);
      if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    (void)hipGetLastError();
    gp::fail(rc, keep.c_str());  // (the clean-up calls may have replaced the message)
  }
  return rc;
}

template <typename Issue>
int run_pass_impl(gp_vgicp_multi_batch* mb, int width, bool err_pass, Issue issue, double* out_host, bool* group_open) {
  DeviceGuard guard;
  const size_t F = (size_t)mb->num_factors;
  // ---- compute: every shard issues its batched kernels into its rows ----
  double* h = static_cast<double*>(err_pass ? mb->h_err : mb->h_stack);
  for (auto& s : mb->shards) {
    GP_HIP(hipSetDevice(s.device));
    gp::DeviceArray& dst = err_pass ? s.d_err : s.d_stack;
    GP_HIP(hipEventRecord(s.e_begin, s.stream));
    const bool zero = mb->use_rccl && !mb->gather;  // (the all-gather overwrites every row; the all-reduce adds zeros to all rows but the writer's)
    if (s.index.empty()) {
      if (zero) GP_HIP(hipMemsetAsync(dst.ptr, 0, sizeof(double) * width * F, s.stream));
      GP_HIP(hipEventRecord(s.e_compute, s.stream));
      continue;
    }
    if (mb->use_rccl) {
      if (zero) GP_HIP(hipMemsetAsync(dst.ptr, 0, sizeof(double) * width * F, s.stream));
      if (s.contiguous) {
        GP_TRY(issue(s, dst.as<double>() + (size_t)s.index[0] * width));
      } else {
        GP_TRY(issue(s, s.d_local.as<double>()));
        const int n = (int)s.index.size();
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((n * width + 255) / 256), dim3(256), 0, s.stream, (const double*)s.d_local.as<double>(),
                           (const int*)s.d_index.as<int>(), dst.as<double>(), n, width);
        GP_HIP(hipGetLastError());
      }
    } else if (s.contiguous) {
      // no collective: the finalize kernel's record stores go straight into the shard's rows of the host stack (pinned, portable: every device sees it at the
      // same address): no device-side stack, no copy operation, nothing left to do but wait for the streams
      void* hd = nullptr;
      GP_HIP(hipHostGetDevicePointer(&hd, h + (size_t)s.index[0] * width, 0));
      GP_TRY(issue(s, static_cast<double*>(hd)));
    } else {
      GP_TRY(issue(s, dst.as<double>()));  // local rows, copied out one by one below
    }
    GP_HIP(hipEventRecord(s.e_compute, s.stream));
  }
  // ---- exchange ----
  if (mb->use_rccl) {
    Rccl& r = rccl();
    GP_NCCL(r.GroupStart());
    *group_open = true;
    for (auto& s : mb->shards) {
      gp::DeviceArray& dst = err_pass ? s.d_err : s.d_stack;
      if (mb->gather) {
        const size_t count = mb->gather_rows * (size_t)width;  // in place: rank k's send buffer is its own slot of the receive buffer
        GP_NCCL(r.AllGather(dst.as<double>() + gather_send_offset(s.index, width), dst.ptr, count, ncclDouble, s.comm, s.stream));
      } else {
        GP_NCCL(r.AllReduce(dst.ptr, dst.ptr, (size_t)width * F, ncclDouble, ncclSum, s.comm, s.stream));
      }
    }
    *group_open = false;
    GP_NCCL(r.GroupEnd());
    Shard& s0 = mb->shards[0];
    GP_HIP(hipSetDevice(s0.device));
    GP_HIP(hipMemcpyAsync(h, (err_pass ? s0.d_err : s0.d_stack).ptr, sizeof(double) * width * F, hipMemcpyDeviceToHost, s0.stream));
    for (auto& s : mb->shards) {
      GP_HIP(hipSetDevice(s.device));
      GP_HIP(hipEventRecord(s.e_done, s.stream));
    }
  } else {
    for (auto& s : mb->shards) {
      GP_HIP(hipSetDevice(s.device));
      if (!s.index.empty() && !s.contiguous) {
        const double* src = (err_pass ? s.d_err : s.d_stack).as<double>();
        for (size_t k = 0; k < s.index.size(); k++)
          GP_HIP(hipMemcpyAsync(h + (size_t)s.index[k] * width, src + k * width, sizeof(double) * width, hipMemcpyDeviceToHost, s.stream));
      }
      GP_HIP(hipEventRecord(s.e_done, s.stream));
    }
  }
  float ms_compute = 0.f, ms_total = 0.f;
  for (auto& s : mb->shards) {
    GP_HIP(hipSetDevice(s.device));
    GP_HIP(hipEventSynchronize(s.e_done));
    float a = 0.f, b = 0.f;
    GP_HIP(hipEventElapsedTime(&a, s.e_begin, s.e_compute));
    GP_HIP(hipEventElapsedTime(&b, s.e_begin, s.e_done));
    ms_compute = std::max(ms_compute, a);
    ms_total = std::max(ms_total, b);
  }
  mb->last_ms_compute = ms_compute;
  mb->last_ms_exchange = std::max(0.f, ms_total - ms_compute);
  memcpy(out_host, h, sizeof(double) * width * F);
  return GP_OK;
}

}  // namespace

extern "C" {

int gp_vgicp_multi_batch_create(gp_vgicp_factor_t* const* factors, int num_factors, const int* shard_of_factor, int num_shards, int use_rccl,
                                gp_vgicp_multi_batch_t** out) {
  if (!out || num_factors < 0 || (num_factors > 0 && !factors)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_create: bad arguments");
  DeviceGuard guard;
  auto* mb = new gp_vgicp_multi_batch;
  mb->num_factors = num_factors;
  // ---- shards: given explicitly, or one per device the factors live on (ascending device id) ----
  std::vector<int> shard_of((size_t)num_factors, 0);
  if (shard_of_factor) {
    if (num_shards <= 0) {
      delete mb;
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_create: num_shards must be positive with an explicit assignment");
    }
    for (int i = 0; i < num_factors; i++) {
      if (shard_of_factor[i] < 0 || shard_of_factor[i] >= num_shards) {
        delete mb;
        return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_create: shard index out of range");
      }
      shard_of[(size_t)i] = shard_of_factor[i];
    }
  } else {
    std::vector<int> devs;
    for (int i = 0; i < num_factors; i++) devs.push_back(gp_vgicp_factor_device(factors[i]));
    std::vector<int> uniq = devs;
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    num_shards = std::max<int>(1, (int)uniq.size());
    for (int i = 0; i < num_factors; i++) shard_of[(size_t)i] = (int)(std::lower_bound(uniq.begin(), uniq.end(), devs[(size_t)i]) - uniq.begin());
  }
  mb->shards.resize((size_t)num_shards);
  int current = 0;
  (void)hipGetDevice(&current);
  for (auto& s : mb->shards) s.device = -1;
  for (int i = 0; i < num_factors; i++) {
    if (!factors[i]) {
      destroy_multi(mb);
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_create: null factor");
    }
    Shard& s = mb->shards[(size_t)shard_of[(size_t)i]];
    const int dev = gp_vgicp_factor_device(factors[i]);
    if (s.device < 0) s.device = dev;
    if (s.device != dev) {
      destroy_multi(mb);
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_create: the factors of one shard live on different devices");
    }
    s.index.push_back(i);
    s.points += gp_vgicp_factor_num_points(factors[i]);
  }
  for (auto& s : mb->shards)
    if (s.device < 0) s.device = current;  // an empty shard still takes part in the collective
  // ---- exchange mode: RCCL needs distinct devices (one rank per device in a communicator) ----
  std::vector<int> devlist;
  for (auto& s : mb->shards) devlist.push_back(s.device);
  std::vector<int> uniq = devlist;
  std::sort(uniq.begin(), uniq.end());
  const bool distinct = std::unique(uniq.begin(), uniq.end()) == uniq.end();
  // use_rccl: 0 = no collective (records straight into the host stack), 1 = ncclAllReduce of the zeroed stack, 2 = in-place ncclAllGather when the shards are equal
  // contiguous ranges in rank order (else the all-reduce), < 0 = automatic (2 when the batch spans several devices)
  bool want_rccl = use_rccl > 0 || (use_rccl < 0 && num_shards > 1);
  const bool want_gather = use_rccl == 2 || use_rccl < 0;
  if (want_rccl && (!distinct || !rccl().ok)) {
    if (use_rccl > 0) {
      destroy_multi(mb);
      return gp::fail(GP_ERROR_INVALID_ARGUMENT, distinct ? "gp_vgicp_multi_batch_create: librccl.so could not be loaded" : "gp_vgicp_multi_batch_create: RCCL needs one shard per device");
    }
    want_rccl = false;
  }
  mb->use_rccl = want_rccl;
  // ---- per-shard resources on the shard's device ----
  const size_t F = (size_t)num_factors;
  int rc = GP_OK;
  for (auto& s : mb->shards) {
    if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
      rc = gp::fail(GP_ERROR_HIP, "gp_vgicp_multi_batch_create: cannot create a stream on the shard's device");
      break;
    }
    s.contiguous = true;
    for (size_t k = 1; k < s.index.size(); k++)
      if (s.index[k] != s.index[k - 1] + 1) s.contiguous = false;
    std::vector<gp_vgicp_factor_t*> mine;
    for (int i : s.index) mine.push_back(factors[i]);
    if ((rc = gp_vgicp_batch_create(mine.data(), (int)mine.size(), s.stream, &s.batch)) != GP_OK) break;
    if (!distinct) gp::batch_set_sources_shared(s.batch, true);  // sibling shards on this device re-read the clouds: keep the stream cacheable
    const size_t rows = mb->use_rccl ? std::max<size_t>(F, 1) : std::max<size_t>(s.index.size(), 1);
    if ((rc = s.d_stack.alloc(sizeof(double) * kRecordDoubles * rows)) != GP_OK || (rc = s.d_err.alloc(sizeof(double) * rows)) != GP_OK) break;
    if (mb->use_rccl && !s.contiguous) {
      if ((rc = s.d_local.alloc(sizeof(double) * kRecordDoubles * s.index.size())) != GP_OK || (rc = s.d_index.alloc(sizeof(int) * s.index.size())) != GP_OK) break;
      if (hipMemcpy(s.d_index.ptr, s.index.data(), sizeof(int) * s.index.size(), hipMemcpyHostToDevice) != hipSuccess) {
        rc = gp::fail(GP_ERROR_HIP, "gp_vgicp_multi_batch_create: index upload failed");
        break;
      }
    }
    s.poses.resize(16 * s.index.size());
    s.poses_eval.resize(16 * s.index.size());
    if (hipEventCreate(&s.e_begin) != hipSuccess || hipEventCreate(&s.e_compute) != hipSuccess || hipEventCreate(&s.e_done) != hipSuccess) {
      rc = gp::fail(GP_ERROR_HIP, "gp_vgicp_multi_batch_create: hipEventCreate failed");
      break;
    }
  }
  if (rc == GP_OK) {
    // portable: one address every device can store to (the no-collective pass) and the copy engine can write (the collective passes)
    if (hipHostMalloc(&mb->h_stack, sizeof(double) * kRecordDoubles * std::max<size_t>(F, 1), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(&mb->h_err, sizeof(double) * std::max<size_t>(F, 1), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
      rc = gp::fail(GP_ERROR_HIP, "gp_vgicp_multi_batch_create: hipHostMalloc (portable) failed");
  }
  if (rc == GP_OK && mb->use_rccl && want_gather) {
    std::vector<std::vector<int>> lists;
    for (auto& s : mb->shards) lists.push_back(s.index);
    mb->gather_rows = gather_rows_of(lists, F);
    mb->gather = mb->gather_rows > 0;
  }
  if (rc == GP_OK && mb->use_rccl) {
    std::vector<ncclComm_t> comms((size_t)num_shards, nullptr);
    const ncclResult_t e = rccl().CommInitAll(comms.data(), num_shards, devlist.data());
    if (e != ncclSuccess) {
      rc = nccl_fail(e, "ncclCommInitAll");
    } else {
      for (int k = 0; k < num_shards; k++) mb->shards[(size_t)k].comm = comms[(size_t)k];
    }
  }
  if (rc != GP_OK) {
    destroy_multi(mb);
    return rc;
  }
  *out = mb;
  return GP_OK;
}

// Host only (no device is touched): what the in-place all-gather of a pass would send from, per shard, for this assignment -- the functions the pass itself
// uses. rows_per_shard = 0: the plan does not allow the all-gather (the pass would all-reduce).
int gp_debug_multi_gather_plan(const int* shard_of_factor, int num_factors, int num_shards, int width, int64_t* rows_per_shard, int64_t* send_offset_doubles) {
  if (!shard_of_factor || num_factors < 0 || num_shards <= 0 || width <= 0 || !rows_per_shard || !send_offset_doubles)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_multi_gather_plan: bad arguments");
  for (int i = 0; i < num_factors; i++)
    if (shard_of_factor[i] < 0 || shard_of_factor[i] >= num_shards) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_multi_gather_plan: shard index out of range");
  std::deque<Shard> shards((size_t)num_shards);  // the container the batch keeps its shards in
  const auto lists = shard_index_lists(shard_of_factor, num_factors, num_shards);
  for (int k = 0; k < num_shards; k++) shards[(size_t)k].index = lists[(size_t)k];
  *rows_per_shard = (int64_t)gather_rows_of(lists, (size_t)num_factors);
  int k = 0;
  for (auto& s : shards) send_offset_doubles[k++] = (int64_t)gather_send_offset(s.index, width);
  return GP_OK;
}

int gp_vgicp_multi_batch_destroy(gp_vgicp_multi_batch_t* mb) {
  destroy_multi(mb);
  return GP_OK;
}

int gp_vgicp_multi_batch_size(const gp_vgicp_multi_batch_t* mb) { return mb ? mb->num_factors : 0; }
int gp_vgicp_multi_batch_num_shards(const gp_vgicp_multi_batch_t* mb) { return mb ? (int)mb->shards.size() : 0; }
int gp_vgicp_multi_batch_uses_rccl(const gp_vgicp_multi_batch_t* mb) { return mb && mb->use_rccl ? (mb->gather ? 2 : 1) : 0; }  // 0 no collective, 1 all-reduce, 2 all-gather

int gp_vgicp_multi_batch_shard_info(const gp_vgicp_multi_batch_t* mb, int shard, int* device, int* num_factors, int64_t* num_points) {
  if (!mb || shard < 0 || shard >= (int)mb->shards.size()) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_shard_info: bad arguments");
  const Shard& s = mb->shards[(size_t)shard];
  if (device) *device = s.device;
  if (num_factors) *num_factors = (int)s.index.size();
  if (num_points) *num_points = s.points;
  return GP_OK;
}

int gp_vgicp_multi_batch_last_timing(const gp_vgicp_multi_batch_t* mb, float* ms_compute, float* ms_exchange) {
  if (!mb) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_last_timing: null");
  if (ms_compute) *ms_compute = mb->last_ms_compute;
  if (ms_exchange) *ms_exchange = mb->last_ms_exchange;
  return GP_OK;
}

int gp_vgicp_multi_batch_linearize(gp_vgicp_multi_batch_t* mb, const double* poses_host, gp_linearized6* out_host) {
  if (!mb || !poses_host || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_linearize: null");
  if (mb->num_factors == 0) return GP_OK;
  for (auto& s : mb->shards)
    for (size_t k = 0; k < s.index.size(); k++) memcpy(s.poses.data() + 16 * k, poses_host + 16 * (size_t)s.index[k], sizeof(double) * 16);
  return run_pass(
    mb, kRecordDoubles, false, [](Shard& s, double* dst) { return gp_vgicp_batch_issue_linearize(s.batch, s.poses.data(), reinterpret_cast<gp_linearized6*>(dst)); },
    reinterpret_cast<double*>(out_host));
}

int gp_vgicp_multi_batch_compute_error(gp_vgicp_multi_batch_t* mb, const double* poses_lin_host, const double* poses_eval_host, double* out_host) {
  if (!mb || !poses_lin_host || !poses_eval_host || !out_host) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_vgicp_multi_batch_compute_error: null");
  if (mb->num_factors == 0) return GP_OK;
  for (auto& s : mb->shards)
    for (size_t k = 0; k < s.index.size(); k++) {
      memcpy(s.poses.data() + 16 * k, poses_lin_host + 16 * (size_t)s.index[k], sizeof(double) * 16);
      memcpy(s.poses_eval.data() + 16 * k, poses_eval_host + 16 * (size_t)s.index[k], sizeof(double) * 16);
    }
  return run_pass(
    mb, 1, true, [](Shard& s, double* dst) { return gp_vgicp_batch_issue_compute_error(s.batch, s.poses.data(), s.poses_eval.data(), dst); }, out_host);
}

}  // extern "C"
