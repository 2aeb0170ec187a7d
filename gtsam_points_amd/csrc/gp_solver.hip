// gp_solver.hip -- the step after the path: the damped normal equations of a pose graph
// of VGICP factors are assembled and solved on the device, so that the H/b records never have to leave HBM between two
// Levenberg-Marquardt iterations.
//
// Replaces (reference, host side, on top of GTSAM / Eigen):
//   optimizers/linear_system_builder.cpp:39-48    DenseLinearSystemBuilder: A = sum of Hessian blocks scattered by key,
//                                                  b = sum of the factors' g, c = sum of the constant terms
//   optimizers/levenberg_marquardt_ext.cpp:146-161 buildDampedSystem: A + lambda I, or A + lambda clamp(diag A) with diagonalDamping
//   optimizers/linear_solver.hpp:18-22             DenseLinearSolver::solve(A, b): A x = b
//
// Shape: variables are 6-dof poses in `num_slots` slots (a factor key that has no slot -- a fixed pose -- drops out of the
// system, exactly as a constant drops out of a GaussianFactorGraph).  Assembly is a GATHER: the host turns the factor key
// list into one contribution list per destination 6x6 block, so every block is summed in a fixed order (deterministic, no
// atomics).  The solve is a dense blocked Cholesky (LL^T, 6x6 blocks = one pose) in f64 held in HBM: per block column one
// kernel for the diagonal block and the panel below it, one grid-wide rank-6 update of the trailing matrix; then
// forward / backward substitution in one workgroup.  Dense O(n^3): meant for the hundreds of poses of a submap graph, not
// tuned (the block-sparse factorisation is the obvious next step).
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "gp_host.hpp"
#include "gp_lm_poses.hpp"

namespace gp {

// which 6x6 of a record a contribution takes: H_target, H_source, H_target_source (as is: row = target, col = source) or
// its transpose (row = source, col = target)
enum : int { TAKE_HT = 0, TAKE_HS = 1, TAKE_HTS = 2, TAKE_HTS_T = 3 };

struct BlockDest {
  int row, col;      // block coordinates (row >= col: lower triangle)
  int begin, count;  // range in the contribution list
};

struct Contribution {
  int factor;
  int take;
};

constexpr int REC_HT = 2, REC_HS = 38, REC_HTS = 74, REC_BT = 110, REC_BS = 116;  // offsets (doubles) inside gp_linearized6

// A (n x n, column-major, lower triangle + diagonal) and, for diagonal destinations, b: one 64-thread workgroup per block
__global__ void __launch_bounds__(64) assemble_kernel(const BlockDest* __restrict__ dests, const Contribution* __restrict__ contribs, const double* __restrict__ records,
                                                      int n, double* __restrict__ A, double* __restrict__ b) {
  const BlockDest d = dests[blockIdx.x];
  const int t = threadIdx.x;
  if (t < 36) {
    const int r = t % 6, c = t / 6;
    double s = 0.0;
    for (int k = 0; k < d.count; k++) {
      const Contribution q = contribs[d.begin + k];
      const double* rec = records + 122 * (size_t)q.factor;
      double v;
      if (q.take == TAKE_HT) {
        v = rec[REC_HT + c * 6 + r];
      } else if (q.take == TAKE_HS) {
        v = rec[REC_HS + c * 6 + r];
      } else if (q.take == TAKE_HTS) {
        v = rec[REC_HTS + c * 6 + r];
      } else {
        v = rec[REC_HTS + r * 6 + c];
      }
      s += v;
    }
    A[(size_t)(6 * d.col + c) * n + 6 * d.row + r] = s;
  } else if (t < 42 && d.row == d.col) {
    // b = sum of g = -b_target / -b_source (HessianFactor(.., -b_t, .., -b_s, ..), integrated_matching_cost_factor.cpp:49)
    const int r = t - 36;
    double s = 0.0;
    for (int k = 0; k < d.count; k++) {
      const Contribution q = contribs[d.begin + k];
      const double* rec = records + 122 * (size_t)q.factor;
      s -= q.take == TAKE_HT ? rec[REC_BT + r] : rec[REC_BS + r];
    }
    b[6 * d.row + r] = s;
  }
}

__global__ void __launch_bounds__(256) sum_errors_kernel(const double* __restrict__ records, int num_factors, double* __restrict__ c_out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int f = threadIdx.x; f < num_factors; f += 256) s += records[122 * (size_t)f + 1];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *c_out = part[0];
}

// buildDampedSystem: diag += lambda (identity damping) or lambda * clamp(diag, min, max) (diagonalDamping), + optional prior
__global__ void __launch_bounds__(256) damp_kernel(double* __restrict__ A, int n, double lambda, int diagonal, double min_diag, double max_diag,
                                                   const double* __restrict__ prior_diag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = A[(size_t)i * n + i];
  double add = diagonal ? lambda * fmin(fmax(d, min_diag), max_diag) : lambda;
  if (prior_diag) add += prior_diag[i];
  A[(size_t)i * n + i] = d + add;
}

// ---- blocked Cholesky, block size 6 -------------------------------------------------------------------------------
// One launch per block column k does both the diagonal block and the panel below it: workgroup 0 factors A_kk into Ldiag[k];
// workgroup i - k (block row i > k) factors A_kk once more for itself (6x6: ~100 flops, cheaper than a kernel boundary) and
// turns A_ik into L_ik = A_ik L_kk^-T in place.  A_kk itself is left untouched, so the redundant factorisations all read the
// same data; the substitution kernel takes the diagonal blocks from Ldiag.
__device__ __forceinline__ bool chol6(double (*a)[6]) {  // in place, lower triangle; false when a pivot is not positive
  bool ok = true;
  for (int j = 0; j < 6; j++) {
    double d = a[j][j];
    for (int p = 0; p < j; p++) d -= a[j][p] * a[j][p];
    if (!(d > 0.0)) {
      ok = false;
      d = 1.0;
    }
    const double l = sqrt(d);
    a[j][j] = l;
    for (int i = j + 1; i < 6; i++) {
      double s = a[i][j];
      for (int p = 0; p < j; p++) s -= a[i][p] * a[j][p];
      a[i][j] = s / l;
    }
  }
  return ok;
}

__global__ void __launch_bounds__(64) chol_panel_kernel(double* __restrict__ A, int n, int k, double* __restrict__ Ldiag, int* __restrict__ status) {
  __shared__ double l[6][6];
  const int t = threadIdx.x;
  const double* diag = A + (size_t)(6 * k) * n + 6 * k;
  if (t < 36) l[t % 6][t / 6] = diag[(size_t)(t / 6) * n + t % 6];
  __syncthreads();
  if (t == 0) {
    if (!chol6(l) && blockIdx.x == 0) atomicExch(status, k + 1);  // not positive definite at this pose block
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (t < 36) Ldiag[36 * (size_t)k + t] = (t % 6 >= t / 6) ? l[t % 6][t / 6] : 0.0;  // column-major 6x6, lower triangle
    return;
  }
  const int i = k + blockIdx.x;
  if (t < 6) {
    double* row = A + (size_t)(6 * k) * n + 6 * i + t;  // element (6i + t, 6k + c) at row[c * n]
    double x[6];
    for (int c = 0; c < 6; c++) {
      double s = row[(size_t)c * n];
      for (int p = 0; p < c; p++) s -= x[p] * l[c][p];
      x[c] = s / l[c][c];
    }
    for (int c = 0; c < 6; c++) row[(size_t)c * n] = x[c];
  }
}

// trailing update A_ij -= L_ik L_jk^T for k < j <= i.  One 256-thread workgroup per 8x8 tile of pose blocks (48 x 48 entries) of the
// lower triangle: the two 48 x 6 panels go through LDS once and every thread produces 9 entries.
// TB = pose blocks per tile edge: 8 for a large trailing matrix, 1 (one 6x6 block per 64-thread workgroup) when it is small and
// parallelism matters more than panel reuse
template <int TB>
__global__ void __launch_bounds__(TB == 1 ? 64 : 256) chol_update_kernel(double* __restrict__ A, int n, int k, int m /* = P - k - 1 */) {
  constexpr int E = 6 * TB;
  __shared__ double Li[E][7], Lj[E][7];  // padded rows: conflict-free column access
  // blockIdx.x enumerates the lower triangle of the tile grid row by row
  const int q = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= q) ti++;
  while (ti * (ti + 1) / 2 > q) ti--;
  const int tj = q - ti * (ti + 1) / 2;
  const int row0 = 6 * (k + 1) + E * ti, col0 = 6 * (k + 1) + E * tj;  // first matrix row / column of the tile
  const int rows = min(E, n - row0), cols = min(E, n - col0);
  const double* panel = A + (size_t)(6 * k) * n;  // L(r, 6k + p) at panel[p * n + r]
  for (int e = threadIdx.x; e < E * 6; e += (int)blockDim.x) {
    const int r = e % E, p = e / E;
    Li[r][p] = r < rows ? panel[(size_t)p * n + row0 + r] : 0.0;
    Lj[r][p] = r < cols ? panel[(size_t)p * n + col0 + r] : 0.0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E * E; e += (int)blockDim.x) {
    const int r = e % E, c = e / E;
    if (r >= rows || c >= cols) continue;
    if (ti == tj && (row0 + r) / 6 < (col0 + c) / 6) continue;  // strictly upper pose blocks of a diagonal tile are never read
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 6; p++) s += Li[r][p] * Lj[c][p];
    A[(size_t)(col0 + c) * n + row0 + r] -= s;
  }
}

// forward then backward substitution in one workgroup: x <- L^-T L^-1 b   (n <= 6 * kMaxSlots)
constexpr int kSolveThreads = 384;  // 6 rows x 64 lanes
__global__ void __launch_bounds__(kSolveThreads) chol_solve_kernel(const double* __restrict__ A, const double* __restrict__ Ldiag, int n, int P,
                                                                   const double* __restrict__ b, double* __restrict__ x) {
  extern __shared__ double y[];  // n doubles
  __shared__ double rhs[6];
  const int t = threadIdx.x, row = t / 64, lane = t % 64;
  for (int i = t; i < n; i += kSolveThreads) y[i] = b[i];
  __syncthreads();
  for (int k = 0; k < P; k++) {  // L y = b
    double s = 0.0;
    for (int c = lane; c < 6 * k; c += 64) s += A[(size_t)c * n + 6 * k + row] * y[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) rhs[row] = y[6 * k + row] - s;
    __syncthreads();
    if (t == 0) {
      const double* d = Ldiag + 36 * (size_t)k;  // L_kk(r, p) at d[p * 6 + r]
      double v[6];
      for (int r = 0; r < 6; r++) {
        double q = rhs[r];
        for (int p = 0; p < r; p++) q -= d[p * 6 + r] * v[p];
        v[r] = q / d[r * 6 + r];
      }
      for (int r = 0; r < 6; r++) y[6 * k + r] = v[r];
    }
    __syncthreads();
  }
  for (int k = P - 1; k >= 0; k--) {  // L^T x = y
    double s = 0.0;
    for (int c = 6 * (k + 1) + lane; c < n; c += 64) s += A[(size_t)(6 * k + row) * n + c] * y[c];  // L(c, 6k+row), contiguous in c
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) rhs[row] = y[6 * k + row] - s;
    __syncthreads();
    if (t == 0) {
      const double* d = Ldiag + 36 * (size_t)k;
      double v[6];
      for (int r = 5; r >= 0; r--) {
        double q = rhs[r];
        for (int p = r + 1; p < 6; p++) q -= d[r * 6 + p] * v[p];  // L_kk(p, r)
        v[r] = q / d[r * 6 + r];
      }
      for (int r = 0; r < 6; r++) y[6 * k + r] = v[r];
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += kSolveThreads) x[i] = y[i];
}

// the last launch of gp_dense_system_step: x, b, c and the status word to where the host reads them (b is not touched by the factorisation)
__global__ void __launch_bounds__(256) dense_step_end_kernel(const double* __restrict__ x, const double* __restrict__ b, const double* __restrict__ c, const int* __restrict__ status, int n,
                                                             double* __restrict__ out_host) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    out_host[i] = x[i];
    out_host[n + i] = b[i];
  }
  if (i == 0) {
    out_host[2 * (size_t)n] = *c;
    out_host[2 * (size_t)n + 1] = (double)*status;
  }
}

// ---- ONE free pose: the whole damped step in one launch (round 6) ----------------------------------------------------------------------------------------
// BASELINE configs[0] as an optimisation (a scan onto a map: one free pose) runs gp_dense_system_step once per trial, and for a 6 x 6 system that step was nine stream
// operations -- two memsets, assembly, error sum, damping, the status memset, the Cholesky launch, the substitutions, the hand-over: ~25 us of launch boundaries around
// a microsecond of arithmetic.  One 64-thread workgroup does all of it with the arithmetic of the kernels above, operation for operation (assemble_kernel's sums in
// contribution order, sum_errors_kernel's 256 strided partials and halving tree, damp_kernel, chol6, chol_solve_kernel's two substitutions with their empty sums), so
// x, b, c and the status are the multi-launch path's bits (tests/test_solver_gpu.py::test_one_pose_dense_step_is_bit_identical).
__global__ void __launch_bounds__(64) dense_one_pose_step_kernel(const BlockDest* __restrict__ dests, const Contribution* __restrict__ contribs, const double* __restrict__ records,
                                                                 int num_factors, double lambda, int diagonal, double min_diag, double max_diag, double* __restrict__ A,
                                                                 double* __restrict__ b, double* __restrict__ c_out, double* __restrict__ x, double* __restrict__ Ldiag,
                                                                 int* __restrict__ status, double* __restrict__ out_host, const LmPoseView epi, const int has_epi) {
  __shared__ double part[256];
  __shared__ double l[6][6], a0[6][6], bb[6], xx[6];
  __shared__ int st;
  const int t = threadIdx.x;
  const BlockDest d = dests[0];
  // sum_errors_kernel: thread i of 256 adds the errors of factors i, i + 256, ...; the halving tree part[i] += part[i + w], w = 128 .. 1
  for (int i = t; i < 256; i += 64) {
    double s = 0.0;
    for (int f = i; f < num_factors; f += 256) s += records[122 * (size_t)f + 1];
    part[i] = s;
  }
  // assemble_kernel
  if (t < 36) {
    const int r = t % 6, c = t / 6;
    double s = 0.0;
    for (int k = 0; k < d.count; k++) {
      const Contribution q = contribs[d.begin + k];
      const double* rec = records + 122 * (size_t)q.factor;
      double v;
      if (q.take == TAKE_HT) v = rec[REC_HT + c * 6 + r];
      else if (q.take == TAKE_HS) v = rec[REC_HS + c * 6 + r];
      else if (q.take == TAKE_HTS) v = rec[REC_HTS + c * 6 + r];
      else v = rec[REC_HTS + r * 6 + c];
      s += v;
    }
    a0[r][c] = s;
  } else if (t < 42) {
    const int r = t - 36;
    double s = 0.0;
    for (int k = 0; k < d.count; k++) {
      const Contribution q = contribs[d.begin + k];
      const double* rec = records + 122 * (size_t)q.factor;
      s -= q.take == TAKE_HT ? rec[REC_BT + r] : rec[REC_BS + r];
    }
    bb[r] = s;
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    for (int i = t; i < w; i += 64) part[i] += part[i + w];
    __syncthreads();
  }
  // damp_kernel (no prior: a step with one goes through the multi-launch path)
  if (t < 6 && lambda > 0.0) {
    const double dd = a0[t][t];
    const double add = diagonal ? lambda * fmin(fmax(dd, min_diag), max_diag) : lambda;
    a0[t][t] = dd + add;
  }
  __syncthreads();
  if (t < 36) {
    A[(size_t)(t / 6) * 6 + t % 6] = a0[t % 6][t / 6];  // column-major 6 x 6 (the upper triangle holds what the assembly left: the same sums, mirrored)
    l[t % 6][t / 6] = a0[t % 6][t / 6];
  }
  if (t < 6) b[t] = bb[t];
  __syncthreads();
  if (t == 0) {
    st = chol6(l) ? 0 : 1;  // chol_panel_kernel: status = k + 1
    // chol_solve_kernel, P = 1: L y = b (the sum over earlier columns is empty: y - 0), then L^T x = y
    double v[6];
    for (int r = 0; r < 6; r++) {
      double q = bb[r] - 0.0;
      for (int p2 = 0; p2 < r; p2++) q -= l[r][p2] * v[p2];
      v[r] = q / l[r][r];
    }
    double y[6];
    for (int r = 0; r < 6; r++) y[r] = v[r];
    for (int r = 5; r >= 0; r--) {
      double q = y[r] - 0.0;
      for (int p2 = r + 1; p2 < 6; p2++) q -= l[p2][r] * v[p2];
      v[r] = q / l[r][r];
    }
    for (int r = 0; r < 6; r++) xx[r] = v[r];
  }
  __syncthreads();
  if (t < 36) Ldiag[t] = (t % 6 >= t / 6) ? l[t % 6][t / 6] : 0.0;
  if (t < 6) {
    x[t] = xx[t];
    out_host[t] = xx[t];
    out_host[6 + t] = bb[t];
  }
  if (t == 0) {
    *c_out = part[0];
    *status = st;
    out_host[12] = part[0];
    out_host[13] = (double)st;
  }
  if (has_epi) {  // the device-resident LM trial's poses (gp_lm_poses.hpp) while x is at hand: one launch less behind the step
    const int n = max(epi.F, epi.N);
    for (int i = t; i < n; i += 64) lm_poses_thread(epi, i, xx, st != 0);
  }
}

}  // namespace gp

constexpr int kMaxSlots = 2048;  // 12288 unknowns: 96 KB of LDS for the substitution vector, 1.2 GB for the dense matrix

struct gp_dense_system {
  int num_slots = 0, num_factors = 0, n = 0;
  hipStream_t stream = nullptr;
  std::vector<gp::BlockDest> dests;
  std::vector<gp::Contribution> contribs;
  gp::DeviceArray d_dests, d_contribs, A, b, c, x, status, prior, Ldiag;
  gp::PinnedArray pinned;  // gp_dense_system_step: x [n] | b [n] | c | status, written by the step's last kernel
  bool built = false;
  bool step_in_flight = false;  // gp_dense_system_issue_step went out, gp_dense_system_finish_step has not collected it
  bool one_launch = true;       // a system of ONE pose runs its step as one launch (dense_one_pose_step_kernel); gp_dense_system_set_one_launch(sys, 0): the multi-launch form
};

extern "C" {

int gp_dense_system_create(int num_slots, const int* factor_slots, int num_factors, gp_stream_t stream, gp_dense_system_t** out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_create: null out");
  *out = nullptr;
  if (num_slots <= 0 || num_slots > kMaxSlots || num_factors < 0 || (num_factors > 0 && !factor_slots))
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_create: 1 <= num_slots <= 2048, factor_slots = [num_factors][2] (target, source; < 0 = fixed)");
  // destination block -> ordered contribution list (factor order = summation order)
  std::map<std::pair<int, int>, std::vector<gp::Contribution>> lists;
  for (int f = 0; f < num_factors; f++) {
    const int st = factor_slots[2 * f], ss = factor_slots[2 * f + 1];
    if (st >= num_slots || ss >= num_slots) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_create: slot index out of range");
    if (st >= 0 && st == ss) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_create: a factor needs two different poses");
    if (st >= 0) lists[{st, st}].push_back({f, gp::TAKE_HT});
    if (ss >= 0) lists[{ss, ss}].push_back({f, gp::TAKE_HS});
    if (st >= 0 && ss >= 0) {
      if (st > ss) {
        lists[{st, ss}].push_back({f, gp::TAKE_HTS});    // row = target, col = source
      } else {
        lists[{ss, st}].push_back({f, gp::TAKE_HTS_T});  // row = source, col = target
      }
    }
  }
  auto* s = new gp_dense_system;
  s->num_slots = num_slots;
  s->num_factors = num_factors;
  s->n = 6 * num_slots;
  s->stream = (hipStream_t)stream;
  for (int p = 0; p < num_slots; p++) lists[{p, p}];  // every diagonal block exists (an unconstrained pose gives a singular system, reported by solve)
  for (auto& kv : lists) {
    gp::BlockDest d;
    d.row = kv.first.first;
    d.col = kv.first.second;
    d.begin = (int)s->contribs.size();
    d.count = (int)kv.second.size();
    s->contribs.insert(s->contribs.end(), kv.second.begin(), kv.second.end());
    s->dests.push_back(d);
  }
  const size_t n = (size_t)s->n;
  int rc = GP_OK;
  if ((rc = s->d_dests.alloc(sizeof(gp::BlockDest) * s->dests.size())) || (rc = s->d_contribs.alloc(sizeof(gp::Contribution) * std::max<size_t>(s->contribs.size(), 1))) ||
      (rc = s->A.alloc(sizeof(double) * n * n)) || (rc = s->b.alloc(sizeof(double) * n)) || (rc = s->x.alloc(sizeof(double) * n)) || (rc = s->c.alloc(sizeof(double))) ||
      (rc = s->status.alloc(sizeof(int))) || (rc = s->prior.alloc(sizeof(double) * n)) || (rc = s->Ldiag.alloc(sizeof(double) * 36 * (size_t)num_slots))) {
    delete s;
    return rc;
  }
  hipError_t e = hipMemcpy(s->d_dests.ptr, s->dests.data(), sizeof(gp::BlockDest) * s->dests.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !s->contribs.empty())
    e = hipMemcpy(s->d_contribs.ptr, s->contribs.data(), sizeof(gp::Contribution) * s->contribs.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    delete s;
    return gp::hip_fail(e, "gp_dense_system_create", __FILE__, __LINE__);
  }
  *out = s;
  return GP_OK;
}

int gp_dense_system_destroy(gp_dense_system_t* s) {
  if (!s) return GP_OK;
  (void)hipStreamSynchronize(s->stream);
  delete s;
  return GP_OK;
}

int gp_dense_system_size(const gp_dense_system_t* s) { return s ? s->n : 0; }

int gp_dense_system_build(gp_dense_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                          const double* prior_diag_host) {
  if (!s || (!records_dev && s->num_factors > 0) || !(lambda >= 0.0)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_build: bad arguments");
  const size_t n = (size_t)s->n;
  GP_HIP(hipMemsetAsync(s->A.ptr, 0, sizeof(double) * n * n, s->stream));
  GP_HIP(hipMemsetAsync(s->b.ptr, 0, sizeof(double) * n, s->stream));
  hipLaunchKernelGGL(gp::assemble_kernel, dim3((unsigned)s->dests.size()), dim3(64), 0, s->stream, s->d_dests.as<gp::BlockDest>(), s->d_contribs.as<gp::Contribution>(),
                     reinterpret_cast<const double*>(records_dev), s->n, s->A.as<double>(), s->b.as<double>());
  hipLaunchKernelGGL(gp::sum_errors_kernel, dim3(1), dim3(256), 0, s->stream, reinterpret_cast<const double*>(records_dev), s->num_factors, s->c.as<double>());
  const double* prior = nullptr;
  if (prior_diag_host) {
    GP_HIP(hipMemcpyAsync(s->prior.ptr, prior_diag_host, sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
    prior = s->prior.as<double>();
  }
  if (lambda > 0.0 || prior) {
    hipLaunchKernelGGL(gp::damp_kernel, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, s->A.as<double>(), s->n, lambda, diagonal_damping, min_diagonal, max_diagonal,
                       prior);
  }
  GP_HIP(hipGetLastError());
  if (prior_diag_host) GP_HIP(hipStreamSynchronize(s->stream));  // the caller's pageable array may go away
  s->built = true;
  return GP_OK;
}

int gp_dense_system_download(const gp_dense_system_t* s, double* A_host, double* b_host, double* c_host) {
  if (!s || !s->built) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_download: build the system first");
  const size_t n = (size_t)s->n;
  GP_HIP(hipStreamSynchronize(s->stream));
  if (A_host) {
    GP_HIP(hipMemcpy(A_host, s->A.ptr, sizeof(double) * n * n, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < n; c++)  // selfadjointView: mirror the lower triangle (linear_system_builder.cpp:42)
      for (size_t r = c + 1; r < n; r++) A_host[r * n + c] = A_host[c * n + r];
  }
  if (b_host) GP_HIP(hipMemcpy(b_host, s->b.ptr, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (c_host) GP_HIP(hipMemcpy(c_host, s->c.ptr, sizeof(double), hipMemcpyDeviceToHost));
  return GP_OK;
}

// the factorisation and the two substitutions on the system's stream (status cleared in front)
static int launch_dense_solve(gp_dense_system_t* s) {
  const int P = s->num_slots, n = s->n;
  double* A = s->A.as<double>();
  GP_HIP(hipMemsetAsync(s->status.ptr, 0, sizeof(int), s->stream));
  for (int k = 0; k < P; k++) {
    const int m = P - k - 1;
    hipLaunchKernelGGL(gp::chol_panel_kernel, dim3(m + 1), dim3(64), 0, s->stream, A, n, k, s->Ldiag.as<double>(), s->status.as<int>());
    if (m >= 128) {
      const int tiles = (m + 7) / 8;
      hipLaunchKernelGGL(gp::chol_update_kernel<8>, dim3((unsigned)((size_t)tiles * (tiles + 1) / 2)), dim3(256), 0, s->stream, A, n, k, m);
    } else if (m > 0) {
      hipLaunchKernelGGL(gp::chol_update_kernel<1>, dim3((unsigned)((size_t)m * (m + 1) / 2)), dim3(64), 0, s->stream, A, n, k, m);
    }
  }
  GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gp::chol_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * (size_t)n)));
  hipLaunchKernelGGL(gp::chol_solve_kernel, dim3(1), dim3(gp::kSolveThreads), sizeof(double) * (size_t)n, s->stream, A, s->Ldiag.as<double>(), n, P, s->b.as<double>(),
                     s->x.as<double>());
  return GP_OK;
}

// DenseLinearSolver::solve(A, b): A x = b by LL^T.  A is overwritten by its factor (build again before the next solve).
int gp_dense_system_solve(gp_dense_system_t* s, double* x_host, double* x_dev_out) {
  if (!s || !s->built) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_solve: build the system first");
  const int n = s->n;
  GP_TRY(launch_dense_solve(s));
  GP_HIP(hipGetLastError());
  s->built = false;
  int h_status = 0;
  GP_HIP(hipMemcpyAsync(&h_status, s->status.ptr, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  if (x_dev_out) GP_HIP(hipMemcpyAsync(x_dev_out, s->x.ptr, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s->stream));
  if (x_host) GP_HIP(hipMemcpyAsync(x_host, s->x.ptr, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
  GP_HIP(hipStreamSynchronize(s->stream));
  if (h_status != 0) return gp::fail(GP_ERROR_INDETERMINATE, "gp_dense_system_solve: the system is not positive definite (indeterminate linear system)");
  return GP_OK;
}

// buildDampedSystem + solve (levenberg_marquardt_ext.cpp:146-161, 200-220) in one stream-ordered pass and ONE synchronisation (TWO when prior_diag_host is given: the
// build waits once more so that the caller's pageable prior array may go away -- ADVICE r05): gp_dense_system_build, _download(b, c) and _solve
// without the waits and copies between them; x, b, c and the status arrive through one pinned block written by the last kernel.  Bit-identical to the three calls.
// b_host / c_host are valid also when the system is indeterminate.
// the step's device work, queued on the system's stream (a prior diagonal is uploaded by gp_dense_system_build: its own synchronisation)
static int issue_step_impl(gp_dense_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                           const double* prior_diag_host, const gp::LmPoseView* epi, bool* fused) {
  if (fused) *fused = false;
  if (!s) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_step: null system");
  const size_t n = (size_t)s->n;
  GP_TRY(s->pinned.ensure(sizeof(double) * (2 * n + 2)));
  if (s->num_slots == 1 && !prior_diag_host && s->one_launch) {
    if ((!records_dev && s->num_factors > 0) || !(lambda >= 0.0)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_step: bad arguments");
    hipLaunchKernelGGL(gp::dense_one_pose_step_kernel, dim3(1), dim3(64), 0, s->stream, s->d_dests.as<gp::BlockDest>(), s->d_contribs.as<gp::Contribution>(),
                       reinterpret_cast<const double*>(records_dev), s->num_factors, lambda, diagonal_damping, min_diagonal, max_diagonal, s->A.as<double>(), s->b.as<double>(),
                       s->c.as<double>(), s->x.as<double>(), s->Ldiag.as<double>(), s->status.as<int>(), s->pinned.as<double>(), epi ? *epi : gp::LmPoseView{}, epi ? 1 : 0);
    if (fused) *fused = epi != nullptr;
    GP_HIP(hipGetLastError());
    s->built = false;
    s->step_in_flight = true;
    return GP_OK;
  }
  GP_TRY(gp_dense_system_build(s, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, prior_diag_host));
  GP_TRY(launch_dense_solve(s));
  double* h = s->pinned.as<double>();
  hipLaunchKernelGGL(gp::dense_step_end_kernel, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, (const double*)s->x.as<double>(), (const double*)s->b.as<double>(),
                     (const double*)s->c.as<double>(), (const int*)s->status.as<int>(), s->n, h);
  GP_HIP(hipGetLastError());
  s->built = false;
  s->step_in_flight = true;
  return GP_OK;
}

int gp_dense_system_issue_step(gp_dense_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                               const double* prior_diag_host) {
  return issue_step_impl(s, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, prior_diag_host, nullptr, nullptr);
}

static int finish_step(gp_dense_system_t* s, double* x_host, double* b_host, double* c_host, bool wait) {
  if (!s || !s->step_in_flight) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_finish_step: no step was issued");
  const size_t n = (size_t)s->n;
  const double* h = s->pinned.as<double>();
  s->step_in_flight = false;
  if (wait) GP_HIP(hipStreamSynchronize(s->stream));
  if (b_host) memcpy(b_host, h + n, sizeof(double) * n);
  if (c_host) *c_host = h[2 * n];
  if (h[2 * n + 1] != 0.0) return gp::fail(GP_ERROR_INDETERMINATE, "gp_dense_system_step: the system is not positive definite (indeterminate linear system)");
  if (x_host) memcpy(x_host, h, sizeof(double) * n);
  return GP_OK;
}

int gp_dense_system_finish_step(gp_dense_system_t* s, double* x_host, double* b_host, double* c_host) { return finish_step(s, x_host, b_host, c_host, true); }
// ... for a caller that has SEEN the stream pass the step (a completion word of work it queued behind the step on the same stream): no wait of its own
int gp_dense_system_collect_step(gp_dense_system_t* s, double* x_host, double* b_host, double* c_host) { return finish_step(s, x_host, b_host, c_host, false); }

int gp_dense_system_step(gp_dense_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                         const double* prior_diag_host, double* x_host, double* b_host, double* c_host) {
  GP_TRY(gp_dense_system_issue_step(s, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, prior_diag_host));
  return gp_dense_system_finish_step(s, x_host, b_host, c_host);
}

// 0: a one-pose system's step takes the multi-launch path too (the bit-identity test, A/B timing); returns what the next step of this system runs (1: one launch)
int gp_dense_system_set_one_launch(gp_dense_system_t* s, int enable) {
  if (!s) return 0;
  s->one_launch = enable != 0;
  return (s->one_launch && s->num_slots == 1) ? 1 : 0;
}

// gp_sparse_system_device_solution's dense form
int gp_dense_system_device_solution(gp_dense_system_t* s, const double** x_dev, const int** status_dev) {
  if (!s) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_dense_system_device_solution: null system");
  if (x_dev) *x_dev = s->x.as<double>();
  if (status_dev) *status_dev = s->status.as<int>();
  return GP_OK;
}

}  // extern "C"

namespace gp {
int dense_issue_step_with_poses(gp_dense_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                                const LmPoseView& poses, bool* fused) {
  return issue_step_impl(sys, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, nullptr, &poses, fused);
}
}  // namespace gp
