// gp_sparse.hip -- the step after the path, block-sparse: the damped normal equations of a pose
// graph assembled as a block-sparse lower triangle (6x6 blocks = one pose) and solved by a sparse LL^T on the device.
//
// Replaces (reference, host side, on top of GTSAM / Eigen):
//   include/gtsam_points/optimizers/linear_system_builder.hpp:41-72   SparseLinearSystemBuilder<BLOCK_SIZE>: lower-triangular
//                                                                    block-sparse A in a given ordering, b, c
//   src/gtsam_points/optimizers/levenberg_marquardt_ext.cpp:146-161    buildDampedSystem
//   include/gtsam_points/optimizers/linear_solver.hpp:18-30            SparseLinearSolver::solve(A, b)   (called at levenberg_marquardt_ext.cpp:200-220)
//
// Host, once per graph (gp_sparse_system_create): elimination order (the slot order of the factor key list, or nested dissection
// by BFS bisection of the pose graph), symbolic factorisation (elimination tree, block structure of L incl. fill), and for every
// block of L the ordered list of block products that update it -- the numeric phase is then a pure GATHER: every block is
// computed by one group of lanes in a fixed order, no atomics, bit-reproducible.
//
// Device, per solve: left-looking block Cholesky scheduled over the elimination tree.  The tree is cut into independent subtrees
// (one workgroup each, columns in elimination order: a column only gathers from its descendants, which are in the same subtree)
// and a top part of separator columns (one workgroup, after the subtrees).  A pose-graph chain ordered by nested dissection
// is ~log2(P) separator columns on top of P / 8-column subtrees; in the natural order it is one path, i.e. one workgroup walking P
// columns.  The forward substitution rides along (y_k is computed when column k is finished); the backward substitution runs
// the same schedule in reverse.  Four launches per solve, whatever the graph.
#include <algorithm>
#include <cstring>
#include <functional>
#include <iterator>
#include <map>
#include <numeric>
#include <queue>
#include <vector>

#include "gp_host.hpp"
#include "gp_lm_poses.hpp"

namespace gp {

// ---- symbolic phase (pure host code; also exported for the CPU tests) ------------------------------------------------------------

struct SparseSymbolic {
  int P = 0;
  std::vector<int> perm, iperm;    // perm[k] = slot eliminated k-th; iperm[slot] = k
  std::vector<int> parent;         // elimination tree (-1: root)
  std::vector<int> colptr;         // [P + 1] block offsets; column k = its diagonal block followed by its sub-diagonal blocks
  std::vector<int> rowidx;         // [nnzL] row (elimination index) of every block
  std::vector<int> upd_ptr;        // [nnzL + 1] -> upd_a / upd_b: block (i, k) -= L[upd_a] * L[upd_b]^T, in ascending source column
  std::vector<int> upd_a, upd_b;
  std::vector<int> upd_bpos;       // position of block upd_b[u] in the row list of its row k (row_ptr[k] + upd_bpos[u] -> row_blk): the separator kernel stages row k in LDS
  std::vector<int> row_ptr;        // [P + 1] -> row_blk / row_col: the blocks L_kj (j < k) of row k and their columns, ascending j
  std::vector<int> row_blk, row_col;
  std::vector<int> work_ptr, work_cols;  // work lists: [num_subtrees] subtrees, then the CHAINS of the top part (separator columns), grouped by level
  std::vector<int> level_ptr;            // [num_levels + 1] -> work lists: level 0 = the subtrees, level l >= 1 = the top chains whose children are all in lower levels
  int num_subtrees = 0;
  long long nnzA = 0;
};

// nested dissection by BFS bisection: order = nd(A) ++ nd(B) ++ separator
static void nested_dissection(const std::vector<std::vector<int>>& adj, std::vector<int>* order) {
  const int P = (int)adj.size();
  std::vector<int> label(P, 0), level(P, -1);  // label: id of the current piece a node belongs to
  order->clear();
  order->reserve(P);
  int next_label = 1;
  std::function<void(const std::vector<int>&)> rec = [&](const std::vector<int>& nodes) {
    if (nodes.size() <= 8) {
      for (int v : nodes) order->push_back(v);
      return;
    }
    const int mine = next_label++;
    for (int v : nodes) label[v] = mine;
    // connected components of the piece
    std::vector<std::vector<int>> comps;
    for (int v : nodes) level[v] = -1;
    for (int s : nodes) {
      if (level[s] != -1) continue;
      comps.emplace_back();
      std::vector<int>& c = comps.back();
      level[s] = 0;
      c.push_back(s);
      for (size_t h = 0; h < c.size(); h++)
        for (int w : adj[c[h]])
          if (label[w] == mine && level[w] == -1) {
            level[w] = 0;
            c.push_back(w);
          }
    }
    if (comps.size() > 1) {
      for (auto& c : comps) {
        std::sort(c.begin(), c.end());
        rec(c);
      }
      return;
    }
    // one component: BFS from a pseudo-peripheral node (two sweeps), cut at the thinnest level near the middle
    auto bfs = [&](int s, std::vector<int>* seq) {
      for (int v : nodes) level[v] = -1;
      seq->clear();
      level[s] = 0;
      seq->push_back(s);
      for (size_t h = 0; h < seq->size(); h++)
        for (int w : adj[(*seq)[h]])
          if (label[w] == mine && level[w] == -1) {
            level[w] = level[(*seq)[h]] + 1;
            seq->push_back(w);
          }
    };
    std::vector<int> seq;
    bfs(nodes[0], &seq);
    bfs(seq.back(), &seq);
    const int depth = level[seq.back()] + 1;
    if (depth < 3) {  // (nearly) complete piece: no useful separator
      for (int v : nodes) order->push_back(v);
      return;
    }
    std::vector<int> width(depth, 0);
    for (int v : nodes) width[level[v]]++;
    std::vector<long long> below(depth + 1, 0);
    for (int l = 0; l < depth; l++) below[l + 1] = below[l] + width[l];
    int cut = -1;
    double best = 1e300;
    for (int l = 1; l + 1 < depth; l++) {
      const double a = (double)below[l], b = (double)(below[depth] - below[l + 1]);
      if (a < 0.25 * nodes.size() || b < 0.25 * nodes.size()) continue;  // balanced enough
      const double score = width[l] + 0.01 * std::abs(a - b);
      if (score < best) {
        best = score;
        cut = l;
      }
    }
    if (cut < 0) cut = depth / 2;
    if (4 * (size_t)width[cut] > nodes.size()) {
      // the separator would be a quarter of the piece or more (band-like pieces a few separators wide): dissecting further only
      // moves columns into the sequential top part.  The piece is eliminated in BFS order instead (a band ordering: low fill).
      for (int v : seq) order->push_back(v);
      return;
    }
    std::vector<int> A, B, S;
    for (int v : nodes) (level[v] < cut ? A : (level[v] > cut ? B : S)).push_back(v);
    rec(A);
    rec(B);
    for (int v : S) order->push_back(v);
  };
  std::vector<int> all(P);
  std::iota(all.begin(), all.end(), 0);
  rec(all);
}

// minimum degree by multiple elimination (ordering = 2): every round eliminates an independent set of the nodes whose degree in the current
// elimination graph is within `slack` of the minimum -- the fill-reducing heuristic of GTSAM's default COLAMD-class orderings (which is what
// the reference's solves run with, optimizers/levenberg_marquardt_ext.cpp:200-220), in its multiple-elimination form because independent nodes
// of one round are siblings in the elimination tree: the rounds are what the device schedule runs side by side.  slack = 0 is the classical
// MMD; slack = 1 lets e.g. the interior nodes of a chain (degree 2, against 1 at the two ends) go in the first round, which makes a chain's
// tree logarithmic instead of one path.  Deterministic: ties by node index.  The graph is explicit (sorted neighbour lists incl. fill), fine
// for pose graphs of 10^4 nodes.
static void minimum_degree(const std::vector<std::vector<int>>& adj0, int slack, std::vector<int>* order) {
  const int P = (int)adj0.size();
  std::vector<std::vector<int>> adj = adj0;
  std::vector<char> gone((size_t)P, 0), blocked((size_t)P, 0);
  order->clear();
  order->reserve(P);
  std::vector<int> cand, merged;
  int left = P;
  while (left > 0) {
    int dmin = P + 1;
    for (int v = 0; v < P; v++)
      if (!gone[v]) dmin = std::min(dmin, (int)adj[v].size());
    cand.clear();
    for (int v = 0; v < P; v++)
      if (!gone[v] && (int)adj[v].size() <= dmin + slack) cand.push_back(v);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    std::fill(blocked.begin(), blocked.end(), 0);
    for (int v : cand) {
      if (blocked[v]) continue;
      // eliminate v: its neighbours become a clique
      const std::vector<int> nb = adj[v];
      for (int a : nb) {
        blocked[a] = 1;  // not in this round: its degree has just changed
        merged.clear();
        std::set_union(adj[a].begin(), adj[a].end(), nb.begin(), nb.end(), std::back_inserter(merged));
        merged.erase(std::remove_if(merged.begin(), merged.end(), [&](int w) { return w == a || w == v; }), merged.end());
        adj[a].swap(merged);
      }
      adj[v].clear();
      gone[v] = 1;
      left--;
      order->push_back(v);
    }
  }
}

static int sparse_symbolic_with(int num_slots, const int* factor_slots, int num_factors, int ordering, SparseSymbolic* out) {
  SparseSymbolic& S = *out;
  const int P = num_slots;
  S.P = P;
  // pose graph
  std::vector<std::vector<int>> adj((size_t)P);
  for (int f = 0; f < num_factors; f++) {
    const int a = factor_slots[2 * f], b = factor_slots[2 * f + 1];
    if (a >= P || b >= P) return fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system: slot index out of range");
    if (a >= 0 && a == b) return fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system: a factor needs two different poses");
    if (a >= 0 && b >= 0) {
      adj[a].push_back(b);
      adj[b].push_back(a);
    }
  }
  for (auto& v : adj) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  }
  S.perm.resize(P);
  if (ordering == 1) {
    nested_dissection(adj, &S.perm);
  } else if (ordering == 2 || ordering == 3) {
    minimum_degree(adj, ordering == 2 ? 0 : 1, &S.perm);
  } else {
    std::iota(S.perm.begin(), S.perm.end(), 0);
  }
  S.iperm.assign(P, -1);
  for (int k = 0; k < P; k++) S.iperm[S.perm[k]] = k;
  // structure of A below the diagonal, per column, in elimination indices
  std::vector<std::vector<int>> col((size_t)P);
  S.nnzA = P;
  for (int v = 0; v < P; v++)
    for (int w : adj[v]) {
      const int i = S.iperm[v], j = S.iperm[w];
      if (i > j) {
        col[j].push_back(i);
        S.nnzA++;
      }
    }
  // symbolic Cholesky: struct(L_j) = struct(A_j) U (struct(L_c) \ {j}) over the children c of j in the elimination tree
  S.parent.assign(P, -1);
  std::vector<std::vector<int>> children((size_t)P);
  for (int j = 0; j < P; j++) {
    std::vector<int>& s = col[j];
    for (int c : children[j])
      for (int r : col[c])
        if (r != j) s.push_back(r);
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    if (!s.empty()) {
      S.parent[j] = s[0];
      children[s[0]].push_back(j);
    }
  }
  S.colptr.assign(P + 1, 0);
  for (int j = 0; j < P; j++) S.colptr[j + 1] = S.colptr[j] + 1 + (int)col[j].size();
  const int nnzL = S.colptr[P];
  S.rowidx.resize(nnzL);
  for (int j = 0; j < P; j++) {
    S.rowidx[S.colptr[j]] = j;
    for (size_t q = 0; q < col[j].size(); q++) S.rowidx[S.colptr[j] + 1 + q] = col[j][q];
  }
  auto find_block = [&](int i, int k) {  // block (i, k), i >= k, must exist
    if (i == k) return S.colptr[k];
    const auto it = std::lower_bound(col[k].begin(), col[k].end(), i);
    return S.colptr[k] + 1 + (int)(it - col[k].begin());
  };
  // update lists (two passes: count, fill), source columns in ascending order; row lists for the substitutions
  S.upd_ptr.assign(nnzL + 1, 0);
  S.row_ptr.assign(P + 1, 0);
  for (int j = 0; j < P; j++)
    for (size_t q = 0; q < col[j].size(); q++) {
      const int k = col[j][q];
      S.row_ptr[k + 1]++;
      for (size_t p = q; p < col[j].size(); p++) S.upd_ptr[find_block(col[j][p], k) + 1]++;
    }
  for (int d = 0; d < nnzL; d++) S.upd_ptr[d + 1] += S.upd_ptr[d];
  for (int k = 0; k < P; k++) S.row_ptr[k + 1] += S.row_ptr[k];
  S.upd_a.resize(S.upd_ptr[nnzL]);
  S.upd_b.resize(S.upd_ptr[nnzL]);
  S.upd_bpos.resize(S.upd_ptr[nnzL]);
  S.row_blk.resize(S.row_ptr[P]);
  S.row_col.resize(S.row_ptr[P]);
  {
    std::vector<int> ucur(S.upd_ptr.begin(), S.upd_ptr.end() - 1), rcur(S.row_ptr.begin(), S.row_ptr.end() - 1);
    for (int j = 0; j < P; j++)
      for (size_t q = 0; q < col[j].size(); q++) {
        const int k = col[j][q], bkj = S.colptr[j] + 1 + (int)q;
        S.row_blk[rcur[k]] = bkj;
        S.row_col[rcur[k]++] = j;
        for (size_t p = q; p < col[j].size(); p++) {
          const int d = find_block(col[j][p], k);
          S.upd_a[ucur[d]] = S.colptr[j] + 1 + (int)p;
          S.upd_bpos[ucur[d]] = rcur[k] - 1 - S.row_ptr[k];
          S.upd_b[ucur[d]++] = bkj;
        }
      }
  }
  // schedule: cut the elimination forest into independent subtrees + the top
  std::vector<long long> weight((size_t)P, 0);
  for (int j = 0; j < P; j++) {
    const long long nb = 1 + (long long)col[j].size();
    weight[j] += nb * nb;  // ~ block products + solves of the column
    if (S.parent[j] >= 0) weight[S.parent[j]] += weight[j];
  }
  std::vector<char> in_top((size_t)P, 0);
  std::priority_queue<std::pair<long long, int>> heap;
  for (int j = 0; j < P; j++)
    if (S.parent[j] < 0) heap.push({weight[j], j});
  // the heaviest subtree is split at its first branching column: the path from its root down to that column goes to the top, the
  // branches become subtrees of their own.  A subtree that is one path down to a leaf cannot be split (nothing in it runs in parallel).
  const size_t want = 256;
  std::vector<int> roots;
  while (!heap.empty() && heap.size() + roots.size() < want) {
    const int r = heap.top().second;
    heap.pop();
    int b = r;
    while (children[b].size() == 1) b = children[b][0];
    if (children[b].empty()) {
      roots.push_back(r);
      continue;
    }
    for (int v = r;; v = children[v][0]) {
      in_top[v] = 1;
      if (v == b) break;
    }
    for (int c : children[b]) heap.push({weight[c], c});
  }
  std::vector<int> owner((size_t)P, -1);  // subtree id; -1: top
  S.num_subtrees = 0;
  while (!heap.empty()) {
    roots.push_back(heap.top().second);
    heap.pop();
  }
  std::sort(roots.begin(), roots.end());
  for (int r : roots) owner[r] = S.num_subtrees++;
  for (int j = P - 1; j >= 0; j--)
    if (!in_top[j] && owner[j] < 0) owner[j] = owner[S.parent[j]];  // parent index > child index: the parent is done
  // the top part, level-scheduled: its columns are cut into CHAINS (a column with exactly one child in the top continues that child's chain)
  // and a chain's level is one above the highest level among the chains below it (subtrees: level 0).  One launch per level, one workgroup
  // per chain: the separators of a dissection's level l run side by side instead of behind each other on one workgroup (round 2: band graphs
  // spent 4-5 ms there).
  std::vector<int> chain_of((size_t)P, -1), chain_level;
  std::vector<std::vector<int>> chain_cols;
  for (int j = 0; j < P; j++) {  // children precede parents
    if (!in_top[j]) continue;
    int top_children = 0, only = -1, lvl = 1;
    for (int c : children[j]) {
      if (in_top[c]) {
        top_children++;
        only = c;
        lvl = std::max(lvl, chain_level[(size_t)chain_of[c]] + 1);
      }
    }
    if (top_children == 1) {
      chain_of[j] = chain_of[only];
      chain_cols[(size_t)chain_of[j]].push_back(j);
    } else {
      chain_of[j] = (int)chain_cols.size();
      chain_cols.push_back({j});
      chain_level.push_back(lvl);
    }
  }
  int num_levels = 1;
  for (int l : chain_level) num_levels = std::max(num_levels, l + 1);
  std::vector<std::vector<int>> by_level((size_t)num_levels);
  for (size_t c = 0; c < chain_cols.size(); c++) by_level[(size_t)chain_level[c]].push_back((int)c);
  const int num_lists = S.num_subtrees + (int)chain_cols.size();
  S.work_ptr.assign(num_lists + 1, 0);
  S.work_cols.clear();
  S.work_cols.reserve(P);
  S.level_ptr.assign(1, 0);
  {
    std::vector<std::vector<int>> sub((size_t)S.num_subtrees);
    for (int j = 0; j < P; j++)
      if (!in_top[j]) sub[(size_t)owner[j]].push_back(j);  // ascending inside every list
    int list = 0;
    for (auto& cols : sub) {
      S.work_cols.insert(S.work_cols.end(), cols.begin(), cols.end());
      S.work_ptr[++list] = (int)S.work_cols.size();
    }
    S.level_ptr.push_back(list);
    for (int l = 1; l < num_levels; l++) {
      for (int c : by_level[(size_t)l]) {
        S.work_cols.insert(S.work_cols.end(), chain_cols[(size_t)c].begin(), chain_cols[(size_t)c].end());
        S.work_ptr[++list] = (int)S.work_cols.size();
      }
      S.level_ptr.push_back(list);
    }
  }
  return GP_OK;
}

// what one solve walks sequentially: the sum over the levels of the longest work list of the level
static int critical_columns(const SparseSymbolic& S) {
  int crit = 0;
  for (size_t l = 0; l + 1 < S.level_ptr.size(); l++) {
    int longest = 0;
    for (int w = S.level_ptr[l]; w < S.level_ptr[l + 1]; w++) longest = std::max(longest, S.work_ptr[w + 1] - S.work_ptr[w]);
    crit += longest;
  }
  return crit;
}

// the one-launch step of small graphs (sparse_small_step_kernel below): one 512-thread workgroup (eight waves of up to 256 registers: the wave form's rows live in registers;
// at 1024 threads the 128-register cap spilled them), up to eight work lists of a level side by side
constexpr int kSmallThreads = 512, kSmallTeams = 8;
constexpr int kSmallTeamDoubles = 158;  // per team: D[6][7] | rhs[6] | v[6] (backward) | pad 2 | part: rpart[16][6] (forward) / bpart[6][6] (backward) | dinv[6]
// LDS bytes that step needs for a symbolic factorisation: L's blocks, y, x, the teams' scratch, the error partials, the index lists (0: the factor does not qualify)
static size_t small_step_lds_bytes(const SparseSymbolic& S, size_t* arena_words_out = nullptr) {
  const size_t P = (size_t)S.P, nnzL = (size_t)S.colptr[S.P];
  const size_t words = S.colptr.size() + S.rowidx.size() + S.upd_ptr.size() + S.upd_a.size() + S.upd_b.size() + S.row_ptr.size() + S.row_blk.size() + S.row_col.size() +
                       S.work_ptr.size() + S.work_cols.size() + S.rowidx.size() /* the blocks' columns */ + S.work_ptr.size() /* the lists' widest columns */ + 2;
  if (arena_words_out) *arena_words_out = words;
  if (P > 128) return 0;  // (row lists stay below the staged kernel's 128-block stage, which the one-launch form assumes)
  const size_t bytes = sizeof(double) * (36 * nnzL + 24 * P + (size_t)kSmallTeams * kSmallTeamDoubles) + sizeof(int) * words + 64;
  return bytes <= 160 * 1024 - 1024 ? bytes : 0;
}
static bool small_step_fits(const SparseSymbolic& S) { return small_step_lds_bytes(S) != 0; }

// ordering = 4 (automatic): nested dissection and minimum degree with slack are both tried and the schedule with the shorter critical path
// (then the smaller factor) is kept -- band-like graphs want the dissection (separators side by side), graphs with random loop closures and
// grids the minimum degree (less fill AND a shorter path); the symbolic phase is host code run once per graph
static int sparse_symbolic(int num_slots, const int* factor_slots, int num_factors, int ordering, SparseSymbolic* out) {
  if (ordering != 4) return sparse_symbolic_with(num_slots, factor_slots, num_factors, ordering, out);
  SparseSymbolic a, b;
  GP_TRY(sparse_symbolic_with(num_slots, factor_slots, num_factors, 1, &a));
  GP_TRY(sparse_symbolic_with(num_slots, factor_slots, num_factors, 3, &b));
  const int ca = critical_columns(a), cb = critical_columns(b);
  bool take_b = cb < ca || (cb == ca && b.colptr[num_slots] < a.colptr[num_slots]);
  // round 6: a factor that fits one compute unit's LDS is solved by ONE launch with every operand in LDS (sparse_small_step_kernel: ~2 us per column instead of 6-8), which
  // outweighs a shorter critical path in columns -- BASELINE configs[2]'s graph: dissection 21 columns / 522 blocks (150 KB: does not fit), minimum degree 34 columns /
  // 305 blocks (fits).  Between two that qualify, or two that do not, the rule above stands.
  const bool fa = small_step_fits(a), fb = small_step_fits(b);
  if (fa != fb) take_b = fb;
  *out = take_b ? std::move(b) : std::move(a);
  return GP_OK;
}

// ---- numeric phase ----------------------------------------------------------------------------------------------------------------

enum : int { STAKE_HT = 0, STAKE_HS = 1, STAKE_HTS = 2, STAKE_HTS_T = 3 };
constexpr int SREC_HT = 2, SREC_HS = 38, SREC_HTS = 74, SREC_BT = 110, SREC_BS = 116;  // offsets (doubles) inside gp_linearized6

struct SparseDest {
  int block;         // destination block of L
  int diag_col;      // >= 0: the block is the diagonal block of this column (b is assembled with it)
  int begin, count;  // range in the contribution list
};
struct SparseContribution {
  int factor, take;
};

// what gp_sparse_system_step folds into the assembly (one launch for: assemble, damp, b and c handed to the host, the status word cleared)
struct SparseStepExtras {
  double lambda, min_diag, max_diag;
  int diagonal;
  const double* prior_diag;  // elimination order, may be null
  const int* perm;           // elimination order -> slot
  double* b_slots_host;      // pinned: b in slot order
  double* c_dev;
  double* c_host;            // pinned
  int* status;
  int num_factors, num_dests;
  double* diag0;             // [6 P], elimination order: the assembled (damped) diagonal of A, the scale a pivot is held against (kPivotTolerance)
};

// A's blocks into their places in L's storage (every block of L has a destination: fill blocks have an empty contribution list and become zero) + b, one 64-lane
// workgroup per destination, factor order.  STEP: the damping of sparse_damp_kernel applied to the diagonal as it is written (same operations in the same order:
// bit-identical), b also stored in slot order where the host reads it, and one extra workgroup that sums the errors (sparse_sum_errors_kernel's order) and clears the status
template <bool STEP>
__global__ void __launch_bounds__(64) sparse_assemble_kernel(const SparseDest* __restrict__ dests, const SparseContribution* __restrict__ contribs,
                                                             const double* __restrict__ records, double* __restrict__ L, double* __restrict__ b, const SparseStepExtras ex) {
  const int t = threadIdx.x;
  if (STEP && (int)blockIdx.x == ex.num_dests) {
    // c = sum of the factors' errors: 256 strided partial sums folded pairwise, exactly as sparse_sum_errors_kernel's 256 threads do (lane t carries threads t, t + 64, t + 128, t + 192)
    __shared__ double part[256];
    for (int q = 0; q < 4; q++) {
      double s = 0.0;
      for (int f = t + 64 * q; f < ex.num_factors; f += 256) s += records[122 * (size_t)f + 1];
      part[t + 64 * q] = s;
    }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      for (int i = t; i < w; i += 64) part[i] += part[i + w];
      __syncthreads();
    }
    if (t == 0) {
      *ex.c_dev = part[0];
      *ex.c_host = part[0];
      *ex.status = 0;
    }
    return;
  }
  const SparseDest d = dests[blockIdx.x];
  if (t < 36) {
    const int r = t % 6, c = t / 6;
    double s = 0.0;
    for (int k = 0; k < d.count; k++) {
      const SparseContribution q = contribs[d.begin + k];
      const double* rec = records + 122 * (size_t)q.factor;
      double v;
      if (q.take == STAKE_HT) {
        v = rec[SREC_HT + c * 6 + r];
      } else if (q.take == STAKE_HS) {
        v = rec[SREC_HS + c * 6 + r];
      } else if (q.take == STAKE_HTS) {
        v = rec[SREC_HTS + c * 6 + r];
      } else {
        v = rec[SREC_HTS + r * 6 + c];
      }
      s += v;
    }
    if (STEP && d.diag_col >= 0 && r == c && (ex.lambda > 0.0 || ex.prior_diag)) {
      double add = ex.diagonal ? __dmul_rn(ex.lambda, fmin(fmax(s, ex.min_diag), ex.max_diag)) : ex.lambda;  // (explicit roundings: no contraction here or in sparse_damp_kernel)
      if (ex.prior_diag) add = __dadd_rn(add, ex.prior_diag[6 * (size_t)d.diag_col + r]);
      s = __dadd_rn(s, add);
    }
    L[36 * (size_t)d.block + t] = s;
    if (d.diag_col >= 0 && r == c) ex.diag0[6 * (size_t)d.diag_col + r] = s;
  } else if (t < 42 && d.diag_col >= 0) {
    const int r = t - 36;
    double s = 0.0;
    for (int k = 0; k < d.count; k++) {
      const SparseContribution q = contribs[d.begin + k];
      const double* rec = records + 122 * (size_t)q.factor;
      s -= q.take == STAKE_HT ? rec[SREC_BT + r] : rec[SREC_BS + r];  // g = -b (integrated_matching_cost_factor.cpp:49)
    }
    b[6 * (size_t)d.diag_col + r] = s;
    if (STEP) ex.b_slots_host[6 * (size_t)ex.perm[d.diag_col] + r] = s;
  }
}

__global__ void __launch_bounds__(256) sparse_damp_kernel(double* __restrict__ L, const int* __restrict__ colptr, int n, double lambda, int diagonal, double min_diag,
                                                          double max_diag, const double* __restrict__ prior_diag /*elimination order*/, double* __restrict__ diag0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double* p = L + 36 * (size_t)colptr[i / 6] + 7 * (i % 6);
  const double d = *p;
  double add = diagonal ? __dmul_rn(lambda, fmin(fmax(d, min_diag), max_diag)) : lambda;
  if (prior_diag) add = __dadd_rn(add, prior_diag[i]);
  *p = __dadd_rn(d, add);
  diag0[i] = *p;
}

struct SparseView {
  const int* colptr;
  const int* rowidx;
  const int* upd_ptr;
  const int* upd_a;
  const int* upd_b;
  const int* upd_bpos;
  const int* row_ptr;
  const int* row_blk;
  const int* row_col;
  const int* work_ptr;
  const int* work_cols;
  double* L;      // [nnzL][36], column-major 6x6 blocks
  double* y;      // [6 P] forward-substituted right-hand side (in: b), elimination order
  double* x;      // [6 P] solution, elimination order
  double* dinv;   // [6 P] reciprocals of L's diagonal, written with every factored column (the substitutions multiply by them)
  const double* diag0;  // [6 P] the assembled diagonal of A (what a pivot is compared with)
  int* status;    // != 0: a pivot was not positive
};


// visibility of LDS writes between the lanes of ONE wave (its LDS operations execute in program order; this keeps the compiler from
// moving them and makes it wait for the writes): what __syncthreads() is for a workgroup, without the s_barrier
#define GP_WAVE_SYNC_LDS()                                   \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
  } while (0)

// ---- the 6 x 6 pieces every numeric kernel below shares (round 6) ------------------------------------------------------------------------------------------------------
// A column of the factorisation is ONE dependent chain -- gather -> 6 x 6 Cholesky -> forward substitution / the blocks below -- and rounds 1-5 spent it on IEEE f64
// divisions and square roots (18 divisions + 6 square roots per column on the chain, ~300 clocks each) and on LDS round trips between them: 6-8 us per column whether
// the operands sat in L2 or in LDS (profiles/r06_solver_step_time_one.jsonl).  Now a pivot costs one reciprocal square root (hardware estimate + two Newton steps, ~1 ulp)
// and everything that divided by a diagonal entry multiplies by the stored reciprocal; the substitutions run out of registers.  All four kernels call THESE functions, so
// the one-launch step and the multi-launch form stay bit-identical by construction.
__device__ __forceinline__ double rsqrt_f64(double x) {  // x > 0
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
#pragma unroll
  for (int it = 0; it < 2; it++) {  // y <- y + y (1/2 - (x / 2) y^2)
    const double e = __builtin_fma(-(hx * y), y, 0.5);
    y = __builtin_fma(y, e, y);
  }
  return y;
}
// ONE wave, all 64 lanes call (lane j < 36 = entry (r, c) = (j % 6, j / 6)): the lower triangle of D (LDS, row stride 7) becomes its Cholesky factor, dinv[p] = 1 / L_pp.
// A pivot that is not positive raises *bad and is replaced by 1 (the caller reports an indeterminate system).
// The entries live in the lanes' registers between the one read and the one write of D: a pivot is broadcast with v_readlane, the two column entries an update needs come
// through ds_bpermute -- no LDS round trip + wait per step (the LDS form measured 480 clocks per pivot: scripts/r06/solver_trace.py).
__device__ __forceinline__ double lane_bcast_f64(double v, const int src_lane /* compile-time constant after unrolling */) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, src_lane), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src_lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
// A pivot must exceed kPivotTolerance x the diagonal entry of A it came from (scale6: the column's six assembled diagonal entries).  Rounds 1-5 asked for piv > 0: whether
// a SINGULAR system (gauge freedom: no pose held) was reported then depended on the sign rounding left on a pivot of magnitude ~1e-13 A_pp; GTSAM's own Cholesky treats
// pivots below a threshold as zero (base/cholesky.cpp: zeroPivotThreshold) for the same reason.  1e-11 relative: a system conditioned worse than that has no digits left.
constexpr double kPivotTolerance = 1e-11;
__device__ __forceinline__ void chol6_wave(double (*D)[7], double* dinv, const int j, int* bad, const double* scale6) {
  const int r = j % 6, c = j / 6;
  double a = j < 36 ? D[r][c] : 0.0;
  double dv[6], sc[6];
#pragma unroll
  for (int p = 0; p < 6; p++) sc[p] = scale6[p];
  bool any_bad = false;
#pragma unroll
  for (int p = 0; p < 6; p++) {
    double piv = lane_bcast_f64(a, 7 * p);  // D[p][p]: wave-uniform
    if (!(piv > kPivotTolerance * sc[p])) {
      any_bad = true;
      piv = 1.0;
    }
    const double rl = rsqrt_f64(piv);
    const double l = piv * rl;
    dv[p] = rl;
    if (j < 36 && c == p && r >= p) a = r == p ? l : a * rl;
    // the column just scaled: entry (r, p) for the lane's row, entry (c, p) for its column (every lane asks: the shuffles are wave-wide)
    const double arp = __shfl(a, (r + 6 * p) & 63, 64), acp = __shfl(a, (c + 6 * p) & 63, 64);
    if (j < 36 && c > p && r >= c) a = __builtin_fma(-arp, acp, a);
  }
  if (j < 36) D[r][c] = a;
  if (j == 36) {
#pragma unroll
    for (int p = 0; p < 6; p++) dinv[p] = dv[p];
  }
  if (j == 0 && any_bad) *bad = 1;
  GP_WAVE_SYNC_LDS();
}
// ONE lane: rhs <- L^-1 rhs by forward substitution, out of registers (every operand is requested before the chain starts)
__device__ __forceinline__ void forward6(double (*D)[7], const double* dinv, double* rhs) {
  double d[6][6], iv[6], y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    iv[i] = dinv[i];
    y[i] = rhs[i];
#pragma unroll
    for (int q = 0; q < 6; q++)
      if (q < i) d[i][q] = D[i][q];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double sum = y[i];
#pragma unroll
    for (int q = 0; q < 6; q++)
      if (q < i) sum = __builtin_fma(-d[i][q], y[q], sum);
    y[i] = sum * iv[i];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) rhs[i] = y[i];
}
// ONE lane: row r of a block below the diagonal, B <- B L^-T (forward substitution along the row); Bk = the block, column-major
// (RM: the block is stored row-major -- the one-launch step's LDS copy -- instead of column-major: same operands, same operations)
template <bool RM = false>
__device__ __forceinline__ void trsm_row6(double* Bk, const int r, double (*D)[7], const double* dinv) {
  double d[6][6], iv[6], o[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    iv[c] = dinv[c];
    o[c] = RM ? Bk[6 * r + c] : Bk[r + 6 * c];
#pragma unroll
    for (int q = 0; q < 6; q++)
      if (q < c) d[c][q] = D[c][q];
  }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    double sum = o[c];
#pragma unroll
    for (int q = 0; q < 6; q++)
      if (q < c) sum = __builtin_fma(-o[q], d[c][q], sum);
    o[c] = sum * iv[c];
  }
#pragma unroll
  for (int c = 0; c < 6; c++) (RM ? Bk[6 * r + c] : Bk[r + 6 * c]) = o[c];
}
// ONE lane: x_k = L_kk^-T v by backward substitution; Dk = the factored diagonal block, column-major; dinv_k = the reciprocals of its diagonal
template <bool RM = false>
__device__ __forceinline__ void back6(const double* Dk, const double* dinv_k, const double* v, double* xk_out) {
  double d[6][6], iv[6], xk[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    iv[i] = dinv_k[i];
    xk[i] = v[i];
#pragma unroll
    for (int q = 0; q < 6; q++)
      if (q > i) d[q][i] = RM ? Dk[6 * q + i] : Dk[q + 6 * i];  // (L^T)_{iq} = L_{qi}
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double sum = xk[i];
#pragma unroll
    for (int q = 0; q < 6; q++)
      if (q > i) sum = __builtin_fma(-d[q][i], xk[q], sum);
    xk[i] = sum * iv[i];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) xk_out[i] = xk[i];
}

// left-looking block Cholesky of the columns of work list (first_list + blockIdx.x), in elimination order
template <int THREADS>
__global__ void __launch_bounds__(THREADS) sparse_factor_kernel(SparseView S, int first_list) {
  __shared__ double D[6][7];   // diagonal block (row stride 7: the column sweeps below are conflict-free), becomes L_kk
  __shared__ double rhs[6], dinv[6];
  __shared__ int bad;
  const int list = first_list + blockIdx.x;
  const int t = threadIdx.x;
  if (t == 0) bad = 0;
  for (int w = S.work_ptr[list]; w < S.work_ptr[list + 1]; w++) {
    const int k = S.work_cols[w];
    const int base = S.colptr[k], nb = S.colptr[k + 1] - base;
    // 1. gather: every entry of every block of the column collects its updates (ascending source column)
    for (int e = t; e < 36 * nb; e += THREADS) {
      const int d = base + e / 36, r = (e % 36) % 6, c = (e % 36) / 6;
      double acc = S.L[36 * (size_t)d + (e % 36)];
      // the products of an entry are a chain of dependent round trips (index -> block -> value) when taken one at a time: the separator columns of a
      // band graph carry ~100 products per block and took ~27 us apiece.  Four products' indices, then their 48 operands, are requested together;
      // the subtractions keep the order of the list (ascending source column), so the factor is bit for bit what the one-at-a-time loop gives
      int u = S.upd_ptr[d];
      const int ue = S.upd_ptr[d + 1];
      constexpr int kBatch = 4;
      for (; u < ue; u += kBatch) {
        int ia[kBatch], ib[kBatch];
#pragma unroll
        for (int w = 0; w < kBatch; w++) {
          const int uu = u + w < ue ? u + w : ue - 1;  // (a short tail repeats the last product's operands and skips its subtraction)
          ia[w] = S.upd_a[uu];
          ib[w] = S.upd_b[uu];
        }
        double av[kBatch][6], bv[kBatch][6];
#pragma unroll
        for (int w = 0; w < kBatch; w++) {
          const double* A = S.L + 36 * (size_t)ia[w];
          const double* B = S.L + 36 * (size_t)ib[w];
#pragma unroll
          for (int q = 0; q < 6; q++) {
            av[w][q] = A[r + 6 * q];
            bv[w][q] = B[c + 6 * q];
          }
        }
#pragma unroll
        for (int w = 0; w < kBatch; w++)
          if (u + w < ue) {
#pragma unroll
            for (int q = 0; q < 6; q++) acc -= av[w][q] * bv[w][q];
          }
      }
      if (d == base) {
        D[r][c] = acc;
      } else {
        S.L[36 * (size_t)d + (e % 36)] = acc;
      }
    }
    // right-hand side of the forward substitution: b_k - sum_j L_kj y_j
    if (t >= 64 && t < 70) {
      const int r = t - 64;
      double acc = S.y[6 * (size_t)k + r];
      int u = S.row_ptr[k];
      const int ue = S.row_ptr[k + 1];
      for (; u + 4 <= ue; u += 4) {  // four blocks of the row requested together, subtracted in list order (see the gather above)
        int ib[4], ic[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
          ib[w] = S.row_blk[u + w];
          ic[w] = S.row_col[u + w];
        }
        double av[4][6], yv[4][6];
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
          for (int q = 0; q < 6; q++) {
            av[w][q] = S.L[36 * (size_t)ib[w] + r + 6 * q];
            yv[w][q] = S.y[6 * (size_t)ic[w] + q];
          }
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
          for (int q = 0; q < 6; q++) acc -= av[w][q] * yv[w][q];
      }
      for (; u < ue; u++) {
        const double* A = S.L + 36 * (size_t)S.row_blk[u];
        const double* yj = S.y + 6 * (size_t)S.row_col[u];
#pragma unroll
        for (int q = 0; q < 6; q++) acc -= A[r + 6 * q] * yj[q];
      }
      rhs[r] = acc;
    }
    __syncthreads();
    // 2. the diagonal block: 6x6 Cholesky by the first wave (chol6_wave), then y_k = L_kk^-1 rhs by one lane out of registers
    if (t < 64) {
      chol6_wave(D, dinv, t, &bad, S.diag0 + 6 * (size_t)k);
      if (t == 0) forward6(D, dinv, rhs);
    }
    __syncthreads();
    if (t < 36) S.L[36 * (size_t)base + t] = (t % 6) >= (t / 6) ? D[t % 6][t / 6] : 0.0;
    if (t >= 64 && t < 70) S.y[6 * (size_t)k + (t - 64)] = rhs[t - 64];
    if (t >= 70 && t < 76) S.dinv[6 * (size_t)k + (t - 70)] = dinv[t - 70];
    // 3. the blocks below: L_ik = B_ik L_kk^-T, one lane per (block, row): forward substitution along the row (trsm_row6)
    for (int e = t; e < 6 * (nb - 1); e += THREADS) trsm_row6(S.L + 36 * (size_t)(base + 1 + e / 6), e % 6, D, dinv);
    __syncthreads();  // the next column of this list reads these blocks (same compute unit: global writes are visible after the barrier)
  }
  if (t == 0 && bad) atomicOr(S.status, 1);
}

// The same for the chains of SEPARATOR columns (levels > 0): columns of 40-70 blocks whose entries each collect ~100 block products.  In the kernel above
// every entry of a block walks its product list by itself -- 12 loads per product and entry (a 6-row of A, a 6-row of B), four products per dependent round
// trip -- and a 512-pose band graph spent ~20 us per such column, 60 columns one after the other (profiles/r03_solver_time.txt).  Here:
//   * the B operands of a column k are the blocks L_kj of ROW k, the same for every block of the column: they are staged once in LDS (with y_j for the forward
//     substitution), and a product names its B by position in that row (upd_bpos);
//   * one thread owns a ROW of a block (six entries): 6 global loads per product instead of 72 per block-entry set, 36 fma against LDS operands;
//   * the product list of a block row is cut into G contiguous slices (G = what fits the workgroup, <= 8), whose partial sums meet in LDS in slice order:
//     a row's chain of dependent round trips is 1 / G as long.  Fixed order => bit-reproducible; the rounding differs from the one-at-a-time order of the kernel
//     above at the 1e-16 level.
// Columns whose row list exceeds kStageBlocks take the per-entry gather of the kernel above (same results as there).
constexpr int kStageBlocks = 128;
template <int THREADS>
__global__ void __launch_bounds__(THREADS) sparse_factor_staged_kernel(SparseView S, int first_list) {
  __shared__ double Bs[kStageBlocks][36];  // L_kj, j = row list of the column being factored
  __shared__ double ys[kStageBlocks][6];   // y_j
  __shared__ double part[THREADS][6];      // slice partials: [row item * G + slice][c]
  __shared__ double rpart[16][6];
  __shared__ double D[6][7];
  __shared__ double rhs[6], dinv[6];
  __shared__ int bad;
  const int list = first_list + blockIdx.x;
  const int t = threadIdx.x;
  if (t == 0) bad = 0;
  for (int w = S.work_ptr[list]; w < S.work_ptr[list + 1]; w++) {
    const int k = S.work_cols[w];
    const int base = S.colptr[k], nb = S.colptr[k + 1] - base;
    const int rb = S.row_ptr[k], nrow = S.row_ptr[k + 1] - rb;
    const bool staged = nrow <= kStageBlocks;
    if (staged) {
      for (int e = t; e < 36 * nrow; e += THREADS) Bs[e / 36][e % 36] = S.L[36 * (size_t)S.row_blk[rb + e / 36] + (e % 36)];
      for (int e = t; e < 6 * nrow; e += THREADS) ys[e / 6][e % 6] = S.y[6 * (size_t)S.row_col[rb + e / 6] + (e % 6)];
      __syncthreads();
      // 1. gather, one thread per (block row, slice)
      const int R = 6 * nb;
      int G = THREADS / R;
      G = G < 1 ? 1 : (G > 8 ? 8 : G);
      for (int item = t; item < R * G; item += THREADS) {  // (one pass unless the column has more than THREADS / 6 blocks)
        const int ri = item / G, g = item % G;
        const int d = base + ri / 6, r = ri % 6;
        const int ub = S.upd_ptr[d], len = S.upd_ptr[d + 1] - ub;
        const int per = (len + G - 1) / G;
        int u = ub + g * per;
        const int ue = min(ub + len, u + per);
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        constexpr int kBatch = 4;
        for (; u < ue; u += kBatch) {
          int ia[kBatch], bp[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; q++) {
            const int uu = u + q < ue ? u + q : ue - 1;
            ia[q] = S.upd_a[uu];
            bp[q] = S.upd_bpos[uu];
          }
          double av[kBatch][6];
#pragma unroll
          for (int q = 0; q < kBatch; q++) {
            const double* A = S.L + 36 * (size_t)ia[q] + r;
#pragma unroll
            for (int m = 0; m < 6; m++) av[q][m] = A[6 * m];
          }
#pragma unroll
          for (int q = 0; q < kBatch; q++)
            if (u + q < ue) {
              const double* B = Bs[bp[q]];
#pragma unroll
              for (int m = 0; m < 6; m++)
#pragma unroll
                for (int c = 0; c < 6; c++) acc[c] -= av[q][m] * B[c + 6 * m];
            }
        }
        if (G == 1 && R <= THREADS) {
          // no slices to meet: the row goes to its place at once
#pragma unroll
          for (int c = 0; c < 6; c++) {
            const double v = S.L[36 * (size_t)d + r + 6 * c] + acc[c];
            if (d == base) D[r][c] = v;
            else S.L[36 * (size_t)d + r + 6 * c] = v;
          }
        } else if (R * G <= THREADS) {
#pragma unroll
          for (int c = 0; c < 6; c++) part[item][c] = acc[c];
        } else {
          // more rows than threads (G == 1, several passes): straight to memory as well
#pragma unroll
          for (int c = 0; c < 6; c++) {
            const double v = S.L[36 * (size_t)d + r + 6 * c] + acc[c];
            if (d == base) D[r][c] = v;
            else S.L[36 * (size_t)d + r + 6 * c] = v;
          }
        }
      }
      // right-hand side of the forward substitution: b_k - sum_j L_kj y_j, sixteen slices of the row list
      if (t < 96) {
        const int r = t % 6, g = t / 6;
        const int per = (nrow + 15) / 16;
        double a = 0.0;
        for (int j = g * per; j < min(nrow, (g + 1) * per); j++)
#pragma unroll
          for (int m = 0; m < 6; m++) a -= Bs[j][r + 6 * m] * ys[j][m];
        rpart[g][r] = a;
      }
      __syncthreads();
      if (G > 1 && R * G <= THREADS) {
        for (int e = t; e < 6 * R; e += THREADS) {
          const int ri = e / 6, c = e % 6;
          const int d = base + ri / 6, r = ri % 6;
          double v = S.L[36 * (size_t)d + r + 6 * c];
          for (int g = 0; g < G; g++) v += part[ri * G + g][c];
          if (d == base) D[r][c] = v;
          else S.L[36 * (size_t)d + r + 6 * c] = v;
        }
      }
      if (t < 6) {
        double a = S.y[6 * (size_t)k + t];
        for (int g = 0; g < 16; g++) a += rpart[g][t];
        rhs[t] = a;
      }
    } else {
      // row list too long for the stage: per-entry gather, as in sparse_factor_kernel
      for (int e = t; e < 36 * nb; e += THREADS) {
        const int d = base + e / 36, r = (e % 36) % 6, c = (e % 36) / 6;
        double acc = S.L[36 * (size_t)d + (e % 36)];
        for (int u = S.upd_ptr[d]; u < S.upd_ptr[d + 1]; u++) {
          const double* A = S.L + 36 * (size_t)S.upd_a[u];
          const double* B = S.L + 36 * (size_t)S.upd_b[u];
#pragma unroll
          for (int q = 0; q < 6; q++) acc -= A[r + 6 * q] * B[c + 6 * q];
        }
        if (d == base) D[r][c] = acc;
        else S.L[36 * (size_t)d + (e % 36)] = acc;
      }
      if (t >= 64 && t < 70) {
        const int r = t - 64;
        double acc = S.y[6 * (size_t)k + r];
        for (int u = rb; u < rb + nrow; u++) {
          const double* A = S.L + 36 * (size_t)S.row_blk[u];
          const double* yj = S.y + 6 * (size_t)S.row_col[u];
#pragma unroll
          for (int q = 0; q < 6; q++) acc -= A[r + 6 * q] * yj[q];
        }
        rhs[r] = acc;
      }
    }
    __syncthreads();
    // 2. the diagonal block: 6x6 Cholesky by the first wave (chol6_wave), then y_k = L_kk^-1 rhs by one lane out of registers
    if (t < 64) {
      chol6_wave(D, dinv, t, &bad, S.diag0 + 6 * (size_t)k);
      if (t == 0) forward6(D, dinv, rhs);
    }
    __syncthreads();
    if (t < 36) S.L[36 * (size_t)base + t] = (t % 6) >= (t / 6) ? D[t % 6][t / 6] : 0.0;
    if (t >= 64 && t < 70) S.y[6 * (size_t)k + (t - 64)] = rhs[t - 64];
    if (t >= 70 && t < 76) S.dinv[6 * (size_t)k + (t - 70)] = dinv[t - 70];
    // 3. the blocks below: L_ik = B_ik L_kk^-T, one lane per (block, row): forward substitution along the row (trsm_row6)
    for (int e = t; e < 6 * (nb - 1); e += THREADS) trsm_row6(S.L + 36 * (size_t)(base + 1 + e / 6), e % 6, D, dinv);
    __syncthreads();  // the next column of this list reads these blocks (same compute unit: global writes are visible after the barrier)
  }
  if (t == 0 && bad) atomicOr(S.status, 1);
}

// backward substitution x_k = L_kk^-T (y_k - sum_i L_ik^T x_i) over the columns of a work list in REVERSE elimination order
__global__ void __launch_bounds__(64) sparse_backsolve_kernel(SparseView S, int first_list) {
  __shared__ double part[6][6], v[6];
  const int list = first_list + blockIdx.x;
  const int t = threadIdx.x;
  for (int w = S.work_ptr[list + 1] - 1; w >= S.work_ptr[list]; w--) {
    const int k = S.work_cols[w];
    const int base = S.colptr[k], nb = S.colptr[k + 1] - base;
    if (t < 36) {
      // lane (c, slice): component c of sum_i L_ik^T x_i over the blocks slice, slice + 6, ... (fixed order)
      const int c = t % 6, slice = t / 6;
      double acc = 0.0;
      for (int p = 1 + slice; p < nb; p += 6) {
        const double* A = S.L + 36 * (size_t)(base + p);
        const double* xi = S.x + 6 * (size_t)S.rowidx[base + p];
#pragma unroll
        for (int q = 0; q < 6; q++) acc += A[q + 6 * c] * xi[q];
      }
      part[slice][c] = acc;
    }
    __syncthreads();
    if (t < 6) {
      double s = S.y[6 * (size_t)k + t];
      for (int sl = 0; sl < 6; sl++) s -= part[sl][t];
      v[t] = s;
    }
    __syncthreads();
    if (t == 0) back6(S.L + 36 * (size_t)base, S.dinv + 6 * (size_t)k, v, S.x + 6 * (size_t)k);
    __syncthreads();  // x_k is read by the next columns of this list (one wave per workgroup: the barrier is free)
  }
}

__global__ void __launch_bounds__(256) sparse_unpermute_kernel(const double* __restrict__ x_elim, const int* __restrict__ perm, int P, double* __restrict__ x_slots) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 6 * P) x_slots[6 * (size_t)perm[i / 6] + i % 6] = x_elim[i];
}

// the last launch of gp_sparse_system_step: x in slot order on the device and where the host reads it, and the status word beside it
__global__ void __launch_bounds__(256) sparse_step_end_kernel(const double* __restrict__ x_elim, const int* __restrict__ perm, int P, double* __restrict__ x_slots,
                                                              double* __restrict__ x_slots_host, const int* __restrict__ status, double* __restrict__ status_host) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 6 * P) {
    const double v = x_elim[i];
    const size_t to = 6 * (size_t)perm[i / 6] + i % 6;
    x_slots[to] = v;
    x_slots_host[to] = v;
  }
  if (i == 0) *status_host = (double)*status;
}

__global__ void __launch_bounds__(256) sparse_sum_errors_kernel(const double* __restrict__ records, int num_factors, double* __restrict__ c_out) {
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


// ---- the damped step of a SMALL graph: the assembly + ONE launch for everything behind it (round 6) -----------------------------------------------------
// The multi-launch step above costs 0.21 ms on BASELINE configs[2]'s graph (63 free poses, 256 factors) and its launches are NOT the cost: the kernels' own time is
// (profiles/r06_solver_kernel_stats.csv).  A column of the factorisation is a chain -- product indices -> operand blocks -> 6 x 6 Cholesky -> triangular solves -- and
// every link is a round trip to L2 behind a workgroup barrier.  For a graph whose whole factor fits the LDS of one compute unit (<= ~400 blocks of 288 B; the index
// lists beside it) ONE 512-thread workgroup factors and solves with every operand in LDS:
//   phase 0  the index lists and the ASSEMBLED system (L's blocks, b, the diagonal) from global memory into LDS: coalesced, ~3 us.  The assembly itself stays
//            sparse_assemble_kernel<true>, one 64-lane workgroup per block of L across the whole chip, in the launch in front: it is a gather of 8-byte values out of the
//            factors' records, and ONE compute unit's vector-memory path needs 40 us for it (measured: the first form of this kernel assembled in place, with its lists in
//            LDS and sixteen loads in flight per thread: 30 - 45 us against the assembly kernel's 6.5)
//   phase 1  the schedule's levels one after the other, the work lists of a level side by side.  Up to four lists in a level -- a team of 8 / lists waves per list,
//            which share a column's gather and meet through LDS words (small_entry_gather + small_panel_sweep, below; in level 0, the independent subtrees, pipelined:
//            the next column's older products are gathered while the team's first wave sweeps); more lists -- level 0: a list per lone wave (small_wave_column), levels
//            above (separator chains): a TEAM of 512 / lists threads in lock step, round r = the r-th column of every list -- gather, barrier, 6 x 6 Cholesky + forward
//            substitution by the team's first wave, barrier, the blocks below, barrier.  (gp_sparse_system_set_one_launch(sys, 2): the lock-step form everywhere, as
//            first built; 3: lone waves in level 0, lock step above.)
//   phase 2  the backward substitution, levels and columns in reverse: a list per wave, no barrier inside a list (the team form: two barriers per round)
//   phase 3  x in slot order to the device array and the host, the status word
// Every scalar is computed by the SAME sequence of operations as in sparse_factor_kernel<256> (lists of level 0) / sparse_factor_staged_kernel<1024> (levels above: the
// slice partials with G and `per` derived from the column's size exactly as there) / sparse_backsolve_kernel, so the step is bit-identical to the multi-launch form
// (tests/test_solver_gpu.py::test_one_launch_step_is_bit_identical); only where the operands live (row-major blocks in LDS) and which thread computes what differ.
// acc - sum over the products u in [u, ue) of (row r of block upd_a[u]) . (row c of block upd_b[u]), subtracted in list order; operands in LDS.  Four products' indices,
// then their 48 operands, are requested together (one entry's products are a chain of dependent LDS round trips otherwise: 800 clocks per product measured)
__device__ __forceinline__ double small_sub_products(double acc, const int* upd_a, const int* upd_b, int u, const int ue, const double* Ls, const int r, const int c) {
  constexpr int kBatch = 4;
  for (; u < ue; u += kBatch) {
    int ia[kBatch], ib[kBatch];
#pragma unroll
    for (int w = 0; w < kBatch; w++) {
      const int uu = u + w < ue ? u + w : ue - 1;  // (a short tail repeats the last product's operands and skips its subtraction)
      ia[w] = upd_a[uu];
      ib[w] = upd_b[uu];
    }
    double av[kBatch][6], bv[kBatch][6];
#pragma unroll
    for (int w = 0; w < kBatch; w++) {
      // (the LDS copy holds its blocks ROW-major: a row is 48 contiguous, 16-byte-aligned bytes = three 128-bit reads instead of six 64-bit ones)
      const double2* A = reinterpret_cast<const double2*>(Ls + 36 * (size_t)ia[w] + 6 * r);
      const double2* B = reinterpret_cast<const double2*>(Ls + 36 * (size_t)ib[w] + 6 * c);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double2 x = A[q], y = B[q];
        av[w][2 * q] = x.x, av[w][2 * q + 1] = x.y;
        bv[w][2 * q] = y.x, bv[w][2 * q + 1] = y.y;
      }
    }
#pragma unroll
    for (int w = 0; w < kBatch; w++)
      if (u + w < ue) {
#pragma unroll
        for (int q = 0; q < 6; q++) acc -= av[w][q] * bv[w][q];
      }
  }
  return acc;
}

// ---- a column by ONE wave (round 6, second form of the one-launch step) ------------------------------------------------------------------------------------------
// The team form below spends a round's 7900 clocks mostly WAITING: three workgroup barriers, one wave factoring the diagonal block while the others idle, lanes exchanging
// entries through LDS.  Here a work list belongs to one wave, which walks its columns with no barrier at all (the next column's operands were written by the same wave),
// and a lane owns a whole ROW of the column's panel -- lanes 0-5 the rows of the diagonal block, lanes 6-11 the six entries of the forward substitution's right-hand side
// while they are gathered (then lane 6 holds them as one row), lanes 12.. the rows of the blocks below -- with its six entries in registers:
//  1. gather: lane (row r of block d) subtracts, product by product in list order, (row r of block upd_a) . (row c of block upd_b) for its six c; the right-hand-side
//     lanes run the same loop over the row list with y in the place of block upd_b (one "row").  Per entry: the team form's and the multi-launch kernels' arithmetic.
//  2. the panel is swept right-looking, pivot by pivot: 1 / sqrt(pivot) and the diagonal block's column p are wave-uniform (v_readlane), every row scales its entry p and
//     updates its entries c > p from its own registers.  Entry by entry that is the SAME sequence of operations on the same operands as chol6_wave (diagonal block),
//     trsm_row6 (blocks below) and forward6 (right-hand side) -- fma(-x_rp, L_cp, x_rc) for p ascending, then x_rc * (1 / L_cc) -- so the factor is bit-identical to the
//     team form's and the multi-launch kernels'.  More than 52 rows below the diagonal: further passes of 64 rows with the pivots' scalars kept.
// Measured (scripts/r06/solver_trace.py, BASELINE configs[2]'s graph: two lists of 29 / 30 columns): a column 6600 clocks (gather 3700, panel 1550, write-back 1050)
// against the team form's 7900; a lone wave issues an instruction every ~4.5 clocks whatever it is, so what a column costs is its instruction count.  Three
// re-formulations of the gather were built and measured no better -- entries dealt three to a lane (4900 clocks: the index chain per turn), a static per-lane program of
// operand offsets in device memory (3600; with the next column's words prefetched 4800: the words' way from L2 is longer than a turn) -- and were removed.
__device__ __forceinline__ void small_wave_column(const int k, const int lane, double* Ls, double* ys, double* dis, const double* d0s, const int* colptr, const int* upd_ptr,
                                                  const int* upd_a, const int* upd_b, const int* row_ptr, const int* row_blk, const int* row_col, int* bad,
                                                  unsigned long long* stamp) {
  const int base = colptr[k], nb = colptr[k + 1] - base;
  const int rb = row_ptr[k], nrow = row_ptr[k + 1] - rb;
  const int rows = 6 * nb + 6;  // 6 (diagonal block) + 6 (right-hand-side entries; lane 6 becomes the row) + 6 (nb - 1)
  double rl[6], pl[6], lcp[6][6], sc[6];
#pragma unroll
  for (int p = 0; p < 6; p++) sc[p] = d0s[6 * (size_t)k + p];
  bool any_bad = false;
  for (int c0 = 0; c0 < rows; c0 += 64) {
    const int i = c0 + lane;
    const bool act = i < rows, is_rhs = i >= 6 && i < 12;
    const int ib = i < 12 ? 0 : (i - 12) / 6 + 1;
    const int r = i < 6 ? i : (i < 12 ? i - 6 : (i - 12) % 6);
    const int d = base + ib;
    // ---- gather ----
    double a[6];
    const int* la = is_rhs ? row_blk : upd_a;
    const int* lb = is_rhs ? row_col : upd_b;
    int u = !act ? 0 : (is_rhs ? rb : upd_ptr[d]);
    const int ue = !act ? 0 : (is_rhs ? rb + nrow : upd_ptr[d + 1]);
    const double* bbase = is_rhs ? ys : Ls;
    const int bblock = is_rhs ? 6 : 36, brow = is_rhs ? 0 : 6;  // doubles between the "blocks" / rows of the second operand (y: one row)
    if (act && !is_rhs) {
      const double2* A = reinterpret_cast<const double2*>(Ls + 36 * (size_t)d + 6 * r);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double2 x = A[q];
        a[2 * q] = x.x, a[2 * q + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 6; c++) a[c] = 0.0;
      if (act) a[0] = ys[6 * (size_t)k + r];
    }
    constexpr int kBatch = 2;
    for (; u < ue; u += kBatch) {
      double av[kBatch][6], bv[kBatch][6][6];
#pragma unroll
      for (int w = 0; w < kBatch; w++) {
        const int uu = u + w < ue ? u + w : ue - 1;  // (a short tail repeats the last product's operands and skips its subtraction)
        const double2* A = reinterpret_cast<const double2*>(Ls + 36 * (size_t)la[uu] + 6 * r);
        const double* Bb = bbase + (size_t)bblock * lb[uu];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const double2 x = A[q];
          av[w][2 * q] = x.x, av[w][2 * q + 1] = x.y;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double2* B = reinterpret_cast<const double2*>(Bb + brow * c);
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const double2 y = B[q];
            bv[w][c][2 * q] = y.x, bv[w][c][2 * q + 1] = y.y;
          }
        }
      }
#pragma unroll
      for (int w = 0; w < kBatch; w++)
        if (u + w < ue) {
#pragma unroll
          for (int c = 0; c < 6; c++) {
#pragma unroll
            for (int q = 0; q < 6; q++) a[c] -= av[w][q] * bv[w][c][q];
          }
        }
    }
    if (c0 == 0) {
      // the right-hand side's six entries (lanes 6-11, entry 0 each) become lane 6's row
      double rh[6];
#pragma unroll
      for (int q = 0; q < 6; q++) rh[q] = lane_bcast_f64(a[0], 6 + q);
      if (lane == 6) {
#pragma unroll
        for (int q = 0; q < 6; q++) a[q] = rh[q];
      }
    }
    if (stamp && c0 == 0) stamp[1] = __builtin_amdgcn_s_memtime();
    // ---- the panel, pivot by pivot ----
#pragma unroll
    for (int p = 0; p < 6; p++) {
      if (c0 == 0) {
        double piv = lane_bcast_f64(a[p], p);
        if (!(piv > kPivotTolerance * sc[p])) {
          any_bad = true;
          piv = 1.0;
        }
        rl[p] = rsqrt_f64(piv);
        pl[p] = piv * rl[p];
      }
      a[p] = (i == p) ? pl[p] : a[p] * rl[p];
      if (c0 == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++)
          if (c > p) lcp[c][p] = lane_bcast_f64(a[p], c);
      }
#pragma unroll
      for (int c = 0; c < 6; c++)
        if (c > p) a[c] = __builtin_fma(-a[p], lcp[c][p], a[c]);
    }
    if (stamp && c0 == 0) stamp[2] = __builtin_amdgcn_s_memtime();
    // ---- back to LDS ----
    if (act && i < 6) {
#pragma unroll
      for (int c = 0; c < 6; c++) Ls[36 * (size_t)base + 6 * r + c] = c <= r ? a[c] : 0.0;
    } else if (act && i == 6) {
#pragma unroll
      for (int c = 0; c < 6; c++) ys[6 * (size_t)k + c] = a[c];
    } else if (act && i >= 12) {
#pragma unroll
      for (int c = 0; c < 6; c++) Ls[36 * (size_t)d + 6 * r + c] = a[c];
    }
    if (c0 == 0 && lane == 7) {
#pragma unroll
      for (int p = 0; p < 6; p++) dis[6 * (size_t)k + p] = rl[p];
    }
  }
  if (any_bad && lane == 0) *bad = 1;
  GP_WAVE_SYNC_LDS();  // (the next column of this wave reads what its other lanes have just written)
  if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
}

// ---- a column by a TEAM of waves without workgroup barriers (the product where a level has at most four lists) ---------------------------------------------------------
// What a lone wave's column costs is its instruction count, and 3700 of its 6400 clocks are the gather.  Here the gather is dealt ENTRY by entry over the G >= 2 waves of
// the list's team (G = 8 / lists: BASELINE configs[2]'s two lists get four waves each; one entry per lane, its products subtracted in list order in place -- the lock-step
// form's arithmetic: small_sub_products; the right-hand side's six entries on the team's last wave, beside the blocks' entries, not behind them), the waves meet at a
// counter in LDS, the team's first wave sweeps the panel row by row (small_panel_sweep: small_wave_column's second half) and releases the others through a sequence word.
// No s_barrier: the other lists' teams run on at their own pace (all eight waves are resident: a wave that polls never keeps the wave it waits for from running).
// Measured (scripts/r06/solver_forms.py, solver_trace.py): a column 6400 -> 4100 clocks (gather + meeting 1850, sweep 1850, release 160), the kernel 276 k -> 245 k;
// pipelined (sparse_small_step_kernel: the other waves start the next column's older products during the sweep) 3650 clocks, 235 k.
// the forward substitution's right-hand side of column k, entry r (on: this lane holds one): b_k[r] - sum over the row list of L_kj[r, :] . y_j, in list order, in place
__device__ __forceinline__ void small_rhs_gather(const int k, const int r, const bool on, const double* Ls, double* ys, const int* row_ptr, const int* row_blk, const int* row_col) {
  if (!on) return;
  const int rb = row_ptr[k], nrow = row_ptr[k + 1] - rb;
  double acc = ys[6 * (size_t)k + r];
  int u = rb;
  const int ue = rb + nrow;
  for (; u < ue; u += 4) {
    int ib[4], ic[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int uu = u + w < ue ? u + w : ue - 1;
      ib[w] = row_blk[uu];
      ic[w] = row_col[uu];
    }
    double av[4][6], yv[4][6];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const double2* A = reinterpret_cast<const double2*>(Ls + 36 * (size_t)ib[w] + 6 * r);
      const double2* Y = reinterpret_cast<const double2*>(ys + 6 * (size_t)ic[w]);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double2 x = A[q], y2 = Y[q];
        av[w][2 * q] = x.x, av[w][2 * q + 1] = x.y;
        yv[w][2 * q] = y2.x, yv[w][2 * q + 1] = y2.y;
      }
    }
#pragma unroll
    for (int w = 0; w < 4; w++)
      if (u + w < ue) {
#pragma unroll
        for (int q = 0; q < 6; q++) acc -= av[w][q] * yv[w][q];
      }
  }
  ys[6 * (size_t)k + r] = acc;
}
// STAGED: the column belongs to a level above 0 -- the sums are those of sparse_factor_staged_kernel<1024>: an entry's product list in G contiguous slices, each summed
// from zero, met in slice order (G = clamp(1024 / (6 blocks), 1, 8)); the right-hand side's row list in sixteen such slices
template <bool STAGED>
__device__ __forceinline__ void small_entry_gather(const int k, const int wt, const int G, const int lane, double* Ls, double* ys, const int* colptr, const int* upd_ptr,
                                                   const int* upd_a, const int* upd_b, const int* row_ptr, const int* row_blk, const int* row_col) {
  const int base = colptr[k], nb = colptr[k + 1] - base;
  // the blocks' entries: one per lane, wave by wave
  for (int e = wt * 64 + lane; e < 36 * nb; e += 64 * G) {
    const int d = base + e / 36, r = (e % 36) % 6, c = (e % 36) / 6;
    double* dst = Ls + 36 * (size_t)d + 6 * r + c;
    if constexpr (!STAGED) {
      *dst = small_sub_products(*dst, upd_a, upd_b, upd_ptr[d], upd_ptr[d + 1], Ls, r, c);
    } else {
      int Gs = 1024 / (6 * nb);
      Gs = Gs < 1 ? 1 : (Gs > 8 ? 8 : Gs);
      const int ub = upd_ptr[d], ulen = upd_ptr[d + 1] - ub;
      const int per = (ulen + Gs - 1) / Gs;
      double v = *dst;
      for (int g = 0; g < Gs; g++) {
        const double acc = small_sub_products(0.0, upd_a, upd_b, ub + g * per, min(ub + ulen, ub + g * per + per), Ls, r, c);
        v += acc;
      }
      *dst = v;
    }
  }
  // the right-hand side's six entries: the team's LAST wave, lanes 58-63 (another instruction stream than the blocks' entries: beside them, where that wave holds none --
  // a column of five blocks leaves the fourth wave of four free --, not behind them)
  if constexpr (!STAGED) {
    small_rhs_gather(k, lane - 58, wt == G - 1 && lane >= 58, Ls, ys, row_ptr, row_blk, row_col);
  } else if (wt == G - 1 && lane >= 58) {
    const int r = lane - 58;
    const int rb = row_ptr[k], nrow = row_ptr[k + 1] - rb;
    const int per = (nrow + 15) / 16;
    double t = ys[6 * (size_t)k + r];
    for (int g = 0; g < 16; g++) {
      double a = 0.0;
      for (int q = g * per; q < min(nrow, (g + 1) * per); q++) {
        const double* A = Ls + 36 * (size_t)row_blk[rb + q] + 6 * r;
        const double* yj = ys + 6 * (size_t)row_col[rb + q];
#pragma unroll
        for (int m = 0; m < 6; m++) a -= A[m] * yj[m];
      }
      t += a;
    }
    ys[6 * (size_t)k + r] = t;
  }
}
// the gathered panel of column k, row by row (lanes 0-5 the diagonal block, lane 6 the right-hand side, lanes 7.. the rows of the blocks below): small_wave_column's sweep
// what the sweep of a column reads that does not depend on the gather: requested by the caller BEFORE it waits for the gathering waves
struct SmallSweepHead {
  int base, nb;
  double sc[6];  // the assembled diagonal entries the pivots are held against
};
__device__ __forceinline__ SmallSweepHead small_sweep_head(const int k, const double* d0s, const int* colptr) {
  SmallSweepHead h;
  h.base = colptr[k];
  h.nb = colptr[k + 1] - h.base;
  const double2* d = reinterpret_cast<const double2*>(d0s + 6 * (size_t)k);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const double2 x = d[q];
    h.sc[2 * q] = x.x, h.sc[2 * q + 1] = x.y;
  }
  return h;
}
__device__ __forceinline__ void small_panel_sweep(const int k, const SmallSweepHead& head, const int lane, double* Ls, double* ys, double* dis, int* bad) {
  const int base = head.base, nb = head.nb;
  const int rows = 6 * nb + 1;
  double rl[6], pl[6], lcp[6][6], sc[6];
#pragma unroll
  for (int p = 0; p < 6; p++) sc[p] = head.sc[p];
  bool any_bad = false;
  for (int c0 = 0; c0 < rows; c0 += 64) {
    const int i = c0 + lane;
    const bool act = i < rows;
    const int ib = i < 7 ? 0 : (i - 7) / 6 + 1;
    const int r = i < 6 ? i : (i < 7 ? 0 : (i - 7) % 6);
    double* row = i == 6 ? ys + 6 * (size_t)k : Ls + 36 * (size_t)(base + ib) + 6 * r;
    double a[6];
    if (act) {
      const double2* A = reinterpret_cast<const double2*>(row);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double2 x = A[q];
        a[2 * q] = x.x, a[2 * q + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 6; c++) a[c] = 0.0;
    }
#pragma unroll
    for (int p = 0; p < 6; p++) {
      if (c0 == 0) {
        double piv = lane_bcast_f64(a[p], p);
        if (!(piv > kPivotTolerance * sc[p])) {
          any_bad = true;
          piv = 1.0;
        }
        rl[p] = rsqrt_f64(piv);
        pl[p] = piv * rl[p];
      }
      a[p] = (i == p) ? pl[p] : a[p] * rl[p];
      if (c0 == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++)
          if (c > p) lcp[c][p] = lane_bcast_f64(a[p], c);
      }
#pragma unroll
      for (int c = 0; c < 6; c++)
        if (c > p) a[c] = __builtin_fma(-a[p], lcp[c][p], a[c]);
    }
#pragma unroll
    for (int c = 1; c < 6; c++) a[c] = (i < c) ? 0.0 : a[c];  // (rows 0-5 of the first pass: the diagonal block's upper triangle is stored as zeros)
    if (act) {
      double2* O = reinterpret_cast<double2*>(row);
#pragma unroll
      for (int q = 0; q < 3; q++) O[q] = make_double2(a[2 * q], a[2 * q + 1]);
    }
    if (c0 == 0 && lane < 6) {
      double v = rl[0];
#pragma unroll
      for (int p = 1; p < 6; p++) v = lane == p ? rl[p] : v;
      dis[6 * (size_t)k + lane] = v;
    }
  }
  if (any_bad && lane == 0) *bad = 1;
}

// backward substitution of a column by ONE wave: the team form's operations (six slices of the blocks below, met in slice order, then back6) without its barriers
__device__ __forceinline__ void small_wave_back_column(const int k, const int lane, const double* Ls, const double* ys, const double* dis, double* xs, const int* colptr,
                                                       const int* rowidx, double* scratch /* [48]: this wave's */) {
  const int base = colptr[k], nb = colptr[k + 1] - base;
  double (*bpart)[6] = reinterpret_cast<double (*)[6]>(scratch);
  if (lane < 36) {
    const int c = lane % 6, slice = lane / 6;
    double acc = 0.0;
    for (int p = 1 + slice; p < nb; p += 6) {
      const double* A = Ls + 36 * (size_t)(base + p);
      const double* xi = xs + 6 * (size_t)rowidx[base + p];
#pragma unroll
      for (int q = 0; q < 6; q++) acc += A[6 * q + c] * xi[q];
    }
    bpart[slice][c] = acc;
  }
  GP_WAVE_SYNC_LDS();
  double sum = 0.0;
  if (lane < 6) {
    sum = ys[6 * (size_t)k + lane];
    for (int sl = 0; sl < 6; sl++) sum -= bpart[sl][lane];
  }
  // (the six sums go to lane 0 through v_readlane instead of through LDS: one write, one fence and one read less on every column's chain)
  double v[6];
#pragma unroll
  for (int q = 0; q < 6; q++) v[q] = lane_bcast_f64(sum, q);
  if (lane == 0) back6<true>(Ls + 36 * (size_t)base, dis + 6 * (size_t)k, v, xs + 6 * (size_t)k);
  GP_WAVE_SYNC_LDS();
}

struct SparseSmallView {
  const double* L_global;      // [nnzL][36] column-major: the assembled (damped) blocks
  const double* y_global;      // [6 P] b, elimination order
  const double* diag0_global;  // [6 P] the assembled diagonal
  const int* arena;       // global copy of the index lists below, `arena_words` ints, copied to LDS first
  const int* level_ptr;   // [num_levels + 1] -> work lists (global; read once per level)
  int arena_words, num_levels, P, nnzL;
  int o_colptr, o_rowidx, o_upd_ptr, o_upd_a, o_upd_b, o_row_ptr, o_row_blk, o_row_col, o_work_ptr, o_work_cols, o_blkcol, o_list_maxnb;  // offsets (ints) inside the arena (blkcol: block -> its column; list_maxnb: work list -> the most blocks a column of it has)
  const int* perm;        // elimination order -> slot (global)
  double* x_slots;        // device, slot order
  double* x_slots_host;   // pinned
  double* status_host;    // pinned
  LmPoseView epi;             // has_epi: the retract of the device-resident LM trial as the kernel's epilogue (gp_lm_poses.hpp)
  int has_epi;
  int wave_columns;           // 1: level 0's work lists (and every level's backward substitution) by one wave each, no barriers inside a list (small_wave_column); 0: the team form
  unsigned long long* trace;  // measurement (gp_debug_sparse_step_trace): shader-clock stamps of thread 0 -- [0] start, [1] = [5] lists and system in LDS, [2] factored, [3] substituted, [4] end,
                              // [8 + 4 r + {0, 1, 2, 3}]: round r < 14 of the first level: start, gathered, diagonal done, blocks below done; null = off
};
__global__ void __launch_bounds__(kSmallThreads) sparse_small_step_kernel(const SparseSmallView V, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double small_lds[];
  double* Ls = small_lds;                                   // [nnzL][36]
  double* ys = Ls + 36 * (size_t)V.nnzL;                    // [6 P]
  double* xs = ys + 6 * (size_t)V.P;                        // [6 P]
  double* dis = xs + 6 * (size_t)V.P;                       // [6 P] reciprocals of L's diagonal
  double* d0s = dis + 6 * (size_t)V.P;                      // [6 P] the assembled diagonal of A (the pivots' scale)
  double* scr = d0s + 6 * (size_t)V.P;                      // [kSmallTeams][kSmallTeamDoubles]
  int* idx = reinterpret_cast<int*>(scr + kSmallTeams * kSmallTeamDoubles);  // the index lists
  __shared__ int bad;
  __shared__ unsigned team_arrive[kSmallTeams], team_done[kSmallTeams];  // (third form) the teams' handshakes: monotonic within a launch
  __shared__ unsigned long long tr_lds[64];  // (the stamps stay in LDS until the end: a store to memory in front of a fence would be waited for, and measured)
  const int t = threadIdx.x;
  const int* colptr = idx + V.o_colptr;
  const int* rowidx = idx + V.o_rowidx;
  const int* upd_ptr = idx + V.o_upd_ptr;
  const int* upd_a = idx + V.o_upd_a;
  const int* upd_b = idx + V.o_upd_b;
  const int* row_ptr = idx + V.o_row_ptr;
  const int* row_blk = idx + V.o_row_blk;
  const int* row_col = idx + V.o_row_col;
  const int* work_ptr = idx + V.o_work_ptr;
  const int* work_cols = idx + V.o_work_cols;
  const int* blkcol = idx + V.o_blkcol;
  const int* list_maxnb = idx + V.o_list_maxnb;
  // ---- phase 0 ----
#define GP_SMALL_STAMP(i)                                                       \
  do {                                                                          \
    if (V.trace && t == 0) tr_lds[i] = __builtin_amdgcn_s_memtime();            \
  } while (0)
  if (V.trace && t < 64) tr_lds[t] = 0;
  __syncthreads();
  GP_SMALL_STAMP(0);
  if (t == 0) bad = 0;
  for (int i = t; i < V.arena_words; i += kSmallThreads) idx[i] = V.arena[i];
  // the assembled system, written by sparse_assemble_kernel<true> in the launch in front of this one (b and c are already with the host): L's blocks -- column-major in
  // global memory, ROW-major in this copy --, b, and the assembled diagonal the pivots are held against.  Coalesced 8-byte reads, four in flight per thread.
  {
    const int nL = 36 * V.nnzL;
#pragma unroll 4
    for (int i = t; i < nL; i += kSmallThreads) {
      const int blk = i / 36, e = i % 36;
      Ls[36 * (size_t)blk + 6 * (e % 6) + e / 6] = V.L_global[i];
    }
    for (int i = t; i < 6 * V.P; i += kSmallThreads) {
      ys[i] = V.y_global[i];
      d0s[i] = V.diag0_global[i];
    }
  }
  __syncthreads();
  GP_SMALL_STAMP(5);  // lists and system are in LDS
  GP_SMALL_STAMP(1);
  // ---- phase 1: factorisation + forward substitution ----
  for (int lvl = 0; lvl < V.num_levels; lvl++) {
    const int first = V.level_ptr[lvl], nlists = V.level_ptr[lvl + 1] - first;
    const bool staged = lvl > 0;
    for (int b0 = 0; b0 < nlists; b0 += kSmallTeams) {  // (more than eight lists in a level: eight at a time)
      const int nb_lists = min(kSmallTeams, nlists - b0);
      if (V.wave_columns == 1 && nb_lists <= kSmallThreads / 128) {
        // up to four lists side by side (any level): a team of G >= 2 waves per list -- gather by all of them (small_entry_gather), panel sweep by the first
        // (small_panel_sweep), two LDS handshakes per column, no workgroup barrier inside the batch (teams that take different paths below meet at its end: s_barrier
        // counts arrivals, not call sites)
        const int G = (kSmallThreads / 64) / nb_lists;
        const int wv = t >> 6, team = wv / G, wt = wv % G, lane = t & 63;
        if (t < kSmallTeams) team_arrive[t] = 0, team_done[t] = 0;
        __syncthreads();
        // pipelined (every column of the list leaves the team's first wave free of entries: 36 nb <= 64 (G - 1)): the first wave gathers the right-hand side and sweeps;
        // the others gather the blocks' entries -- and, while the first wave sweeps column k, already the products of the NEXT column that do not come from column k
        // (only the previous column of the list can still be in the making: every other source column was swept before it); the product from column k, last in its
        // list, follows when the sweep is released.  Same products in the same order per entry: the same bits.
        const bool pipe = !staged && V.wave_columns == 1 && G >= 2 && team < nb_lists && 36 * list_maxnb[first + b0 + (team < nb_lists ? team : 0)] <= 64 * (G - 1);
        if (pipe) {
          const int list = first + b0 + team;
          const int wb = work_ptr[list], we = work_ptr[list + 1];
          if (wt == 0) {
            unsigned seq = 0;
            int rd = 0;
            for (int w = wb; w < we; w++, rd++) {
              const int k = work_cols[w];
              const bool st = V.trace && t == 0 && lvl == 0 && b0 == 0 && rd < 14;
              if (st) tr_lds[8 + 4 * rd] = __builtin_amdgcn_s_memtime();
              small_rhs_gather(k, lane - 58, lane >= 58, Ls, ys, row_ptr, row_blk, row_col);
              GP_WAVE_SYNC_LDS();
              seq++;
              const SmallSweepHead head = small_sweep_head(k, d0s, colptr);
              while (__hip_atomic_load(&team_arrive[team], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq * (unsigned)(G - 1)) __builtin_amdgcn_s_sleep(1);
              if (st) tr_lds[8 + 4 * rd + 1] = __builtin_amdgcn_s_memtime();
              small_panel_sweep(k, head, lane, Ls, ys, dis, &bad);
              GP_WAVE_SYNC_LDS();
              if (st) tr_lds[8 + 4 * rd + 2] = __builtin_amdgcn_s_memtime();
              if (lane == 0) __hip_atomic_store(&team_done[team], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (st) tr_lds[8 + 4 * rd + 3] = __builtin_amdgcn_s_memtime();
            }
          } else {
            const int e = (wt - 1) * 64 + lane;
            double acc = 0.0;
            bool started = false;  // the entry's value and its older products are already in acc
            unsigned seq = 0;
            for (int w = wb; w < we; w++) {
              const int k = work_cols[w], kprev = w > wb ? work_cols[w - 1] : -1;
              const int base = colptr[k], nb = colptr[k + 1] - base;
              const bool mine = e < 36 * nb;
              const int d = base + (mine ? e / 36 : 0), r = (e % 36) % 6, c = (e % 36) / 6;
              double* dst = Ls + 36 * (size_t)d + 6 * r + c;
              const int u0 = upd_ptr[d], ue = upd_ptr[d + 1];
              const bool tail = mine && ue > u0 && blkcol[upd_a[ue - 1]] == kprev;  // the last product comes from the column in the making
              if (mine && !started) acc = small_sub_products(*dst, upd_a, upd_b, u0, ue - (tail ? 1 : 0), Ls, r, c);
              if (seq > 0)
                while (__hip_atomic_load(&team_done[team], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) __builtin_amdgcn_s_sleep(1);
              if (tail) acc = small_sub_products(acc, upd_a, upd_b, ue - 1, ue, Ls, r, c);
              if (mine) *dst = acc;
              GP_WAVE_SYNC_LDS();
              seq++;
              if (lane == 0) __hip_atomic_fetch_add(&team_arrive[team], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              started = false;
              if (w + 1 < we) {  // the next column's older products, while the first wave sweeps this one
                const int kn = work_cols[w + 1];
                const int bn = colptr[kn], nbn = colptr[kn + 1] - bn;
                if (e < 36 * nbn) {
                  const int dn = bn + e / 36;
                  const int v0 = upd_ptr[dn], ve = upd_ptr[dn + 1];
                  const bool tn = ve > v0 && blkcol[upd_a[ve - 1]] == k;
                  acc = small_sub_products(Ls[36 * (size_t)dn + 6 * r + c], upd_a, upd_b, v0, ve - (tn ? 1 : 0), Ls, r, c);
                }
                started = true;
              }
            }
          }
          __syncthreads();
          continue;
        }
        if (team < nb_lists) {
          const int list = first + b0 + team;
          unsigned seq = 0;
          int rd = 0;
          for (int w = work_ptr[list]; w < work_ptr[list + 1]; w++, rd++) {
            const int k = work_cols[w];
            const bool st = V.trace && t == 0 && lvl == 0 && b0 == 0 && rd < 14;
            if (st) tr_lds[8 + 4 * rd] = __builtin_amdgcn_s_memtime();
            if (staged) small_entry_gather<true>(k, wt, G, lane, Ls, ys, colptr, upd_ptr, upd_a, upd_b, row_ptr, row_blk, row_col);
            else small_entry_gather<false>(k, wt, G, lane, Ls, ys, colptr, upd_ptr, upd_a, upd_b, row_ptr, row_blk, row_col);
            GP_WAVE_SYNC_LDS();
            seq++;
            if (lane == 0) __hip_atomic_fetch_add(&team_arrive[team], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wt == 0) {
              const SmallSweepHead head = small_sweep_head(k, d0s, colptr);
              while (__hip_atomic_load(&team_arrive[team], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq * (unsigned)G) __builtin_amdgcn_s_sleep(1);
              if (st) tr_lds[8 + 4 * rd + 1] = __builtin_amdgcn_s_memtime();
              small_panel_sweep(k, head, lane, Ls, ys, dis, &bad);
              GP_WAVE_SYNC_LDS();
              if (st) tr_lds[8 + 4 * rd + 2] = __builtin_amdgcn_s_memtime();
              if (lane == 0) __hip_atomic_store(&team_done[team], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              while (__hip_atomic_load(&team_done[team], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) __builtin_amdgcn_s_sleep(1);
            }
            if (st) tr_lds[8 + 4 * rd + 3] = __builtin_amdgcn_s_memtime();
          }
        }
        __syncthreads();
        continue;
      }
      if (!staged && V.wave_columns) {
        // level 0: a work list per WAVE, no barrier between its columns (small_wave_column)
        const int wv = t >> 6;
        if (wv < nb_lists) {
          const int list = first + b0 + wv;
          int rd = 0;
          for (int w = work_ptr[list]; w < work_ptr[list + 1]; w++, rd++) {
            unsigned long long* st = (V.trace && t < 64 && b0 == 0 && rd < 14) ? tr_lds + 8 + 4 * rd : nullptr;
            if (st && t == 0) st[0] = __builtin_amdgcn_s_memtime();
            small_wave_column(work_cols[w], t & 63, Ls, ys, dis, d0s, colptr, upd_ptr, upd_a, upd_b, row_ptr, row_blk, row_col, &bad, t == 0 ? st : nullptr);
          }
        }
        __syncthreads();
        continue;
      }
      const int T = (kSmallThreads / nb_lists) & ~63;     // threads per team: whole waves
      const int team = t / T, j = t % T;
      const bool member = team < nb_lists;
      const int list = first + b0 + (member ? team : 0);
      const int w0 = work_ptr[list], len = member ? work_ptr[list + 1] - w0 : 0;
      int rounds = 0;
      for (int q = 0; q < nb_lists; q++) rounds = max(rounds, work_ptr[first + b0 + q + 1] - work_ptr[first + b0 + q]);
      double* S = scr + (size_t)(member ? team : 0) * kSmallTeamDoubles;
      double (*D)[7] = reinterpret_cast<double (*)[7]>(S);
      double* rhs = S + 42;
      double (*rpart)[6] = reinterpret_cast<double (*)[6]>(S + 56);
      double* dinv = S + 152;
      for (int rd = 0; rd < rounds; rd++) {
        const bool on = member && rd < len;
        const int k = on ? work_cols[w0 + rd] : 0;
        const int base = colptr[k], nb = colptr[k + 1] - base;
        const int rb = row_ptr[k], nrow = row_ptr[k + 1] - rb;
        const bool stamp = lvl == 0 && b0 == 0 && rd < 14;
        if (stamp) GP_SMALL_STAMP(8 + 4 * rd);
        if (on) {
          // 1. gather: one thread per entry of the column's blocks
          if (!staged) {
            // (measured and removed, round 6: letting the team's idle waves subtract the next column's products of all source columns but this one while the first wave
            //  factors the diagonal block -- bit-identical, and no faster: a round's gather is ~2300 clocks with NO products at all)
            for (int e = j; e < 36 * nb; e += T) {
              const int d = base + e / 36, r = (e % 36) % 6, c = (e % 36) / 6;
              double acc = Ls[36 * (size_t)d + 6 * r + c];
              acc = small_sub_products(acc, upd_a, upd_b, upd_ptr[d], upd_ptr[d + 1], Ls, r, c);
              if (d == base) D[r][c] = acc;
              else Ls[36 * (size_t)d + 6 * r + c] = acc;
            }
            if (j < 6) {  // right-hand side of the forward substitution: b_k - sum_j L_kj y_j, in list order
              const int r = j;
              double acc = ys[6 * (size_t)k + r];
              // (four blocks of the row requested together, subtracted in list order: one at a time every block is two dependent LDS round trips + six dependent
              //  multiply-adds on the round's critical path -- the stamps showed the gather growing with the row list, not with the product lists)
              int u = rb;
              const int ue = rb + nrow;
              for (; u < ue; u += 4) {
                int ib[4], ic[4];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                  const int uu = u + w < ue ? u + w : ue - 1;
                  ib[w] = row_blk[uu];
                  ic[w] = row_col[uu];
                }
                double av[4][6], yv[4][6];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                  const double2* A = reinterpret_cast<const double2*>(Ls + 36 * (size_t)ib[w] + 6 * r);
                  const double2* Y = reinterpret_cast<const double2*>(ys + 6 * (size_t)ic[w]);
#pragma unroll
                  for (int q = 0; q < 3; q++) {
                    const double2 x = A[q], y2 = Y[q];
                    av[w][2 * q] = x.x, av[w][2 * q + 1] = x.y;
                    yv[w][2 * q] = y2.x, yv[w][2 * q + 1] = y2.y;
                  }
                }
#pragma unroll
                for (int w = 0; w < 4; w++)
                  if (u + w < ue) {
#pragma unroll
                    for (int q = 0; q < 6; q++) acc -= av[w][q] * yv[w][q];
                  }
              }
              rhs[r] = acc;
            }
          } else {
            // the staged kernel's sums: the product list of a block row in G contiguous slices, each summed from zero, met in slice order (G and `per` as there)
            const int R = 6 * nb;
            int G = 1024 / R;
            G = G < 1 ? 1 : (G > 8 ? 8 : G);
            for (int e = j; e < 36 * nb; e += T) {
              const int d = base + e / 36, r = (e % 36) % 6, c = (e % 36) / 6;
              const int ub = upd_ptr[d], ulen = upd_ptr[d + 1] - ub;
              const int per = (ulen + G - 1) / G;
              double v = Ls[36 * (size_t)d + 6 * r + c];
              for (int g = 0; g < G; g++) {
                const double acc = small_sub_products(0.0, upd_a, upd_b, ub + g * per, min(ub + ulen, ub + g * per + per), Ls, r, c);
                v += acc;
              }
              if (d == base) D[r][c] = v;
              else Ls[36 * (size_t)d + 6 * r + c] = v;
            }
            for (int jj = j; jj < 96; jj += T) {  // right-hand side: sixteen slices of the row list (a team may be one wave: 64 threads)
              const int r = jj % 6, g = jj / 6;
              const int per = (nrow + 15) / 16;
              double a = 0.0;
              for (int q = g * per; q < min(nrow, (g + 1) * per); q++) {
                const double* A = Ls + 36 * (size_t)row_blk[rb + q] + 6 * r;
                const double* yj = ys + 6 * (size_t)row_col[rb + q];
#pragma unroll
                for (int m = 0; m < 6; m++) a -= A[m] * yj[m];
              }
              rpart[g][r] = a;
            }
          }
        }
        __syncthreads();
        if (stamp) GP_SMALL_STAMP(8 + 4 * rd + 1);
        // 2. the diagonal block: 6 x 6 Cholesky by the team's first wave (lane = (row, column)), then y_k = L_kk^-1 rhs
        if (on && j < 64) {
          if (staged && j < 6) {
            double a = ys[6 * (size_t)k + j];
            for (int g = 0; g < 16; g++) a += rpart[g][j];
            rhs[j] = a;
          }
          GP_WAVE_SYNC_LDS();  // (rhs is complete before lane 0 reads it)
          chol6_wave(D, dinv, j, &bad, d0s + 6 * (size_t)k);
          if (j == 0) forward6(D, dinv, rhs);
        }
        __syncthreads();
        if (stamp) GP_SMALL_STAMP(8 + 4 * rd + 2);
        if (on) {
          if (j < 36) Ls[36 * (size_t)base + 6 * (j % 6) + j / 6] = (j % 6) >= (j / 6) ? D[j % 6][j / 6] : 0.0;
          if (j >= 36 && j < 42) ys[6 * (size_t)k + (j - 36)] = rhs[j - 36];
          if (j >= 42 && j < 48) dis[6 * (size_t)k + (j - 42)] = dinv[j - 42];
          // 3. the blocks below: L_ik = B_ik L_kk^-T, one thread per (block, row)
          for (int e = j; e < 6 * (nb - 1); e += T) trsm_row6<true>(Ls + 36 * (size_t)(base + 1 + e / 6), e % 6, D, dinv);
        }
        __syncthreads();
        if (stamp) GP_SMALL_STAMP(8 + 4 * rd + 3);
      }
    }
  }
  GP_SMALL_STAMP(2);
  // ---- phase 2: backward substitution, levels and columns in reverse ----
  for (int lvl = V.num_levels - 1; lvl >= 0; lvl--) {
    const int first = V.level_ptr[lvl], nlists = V.level_ptr[lvl + 1] - first;
    for (int b0 = 0; b0 < nlists; b0 += kSmallTeams) {
      const int nb_lists = min(kSmallTeams, nlists - b0);
      if (V.wave_columns) {
        const int wv = t >> 6;
        if (wv < nb_lists) {
          const int list = first + b0 + wv;
          for (int w = work_ptr[list + 1] - 1; w >= work_ptr[list]; w--)
            small_wave_back_column(work_cols[w], t & 63, Ls, ys, dis, xs, colptr, rowidx, scr + (size_t)wv * kSmallTeamDoubles + 56);
        }
        __syncthreads();
        continue;
      }
      const int T = (kSmallThreads / nb_lists) & ~63;
      const int team = t / T, j = t % T;
      const bool member = team < nb_lists;
      const int list = first + b0 + (member ? team : 0);
      const int w0 = work_ptr[list], len = member ? work_ptr[list + 1] - w0 : 0;
      int rounds = 0;
      for (int q = 0; q < nb_lists; q++) rounds = max(rounds, work_ptr[first + b0 + q + 1] - work_ptr[first + b0 + q]);
      double* S = scr + (size_t)(member ? team : 0) * kSmallTeamDoubles;
      double* vv = S + 48;
      double (*bpart)[6] = reinterpret_cast<double (*)[6]>(S + 56);
      for (int rd = 0; rd < rounds; rd++) {
        const bool on = member && rd < len;
        const int k = on ? work_cols[w0 + len - 1 - rd] : 0;
        const int base = colptr[k], nb = colptr[k + 1] - base;
        if (on && j < 36) {
          const int c = j % 6, slice = j / 6;
          double acc = 0.0;
          for (int p = 1 + slice; p < nb; p += 6) {
            const double* A = Ls + 36 * (size_t)(base + p);
            const double* xi = xs + 6 * (size_t)rowidx[base + p];
#pragma unroll
            for (int q = 0; q < 6; q++) acc += A[6 * q + c] * xi[q];
          }
          bpart[slice][c] = acc;
        }
        __syncthreads();
        if (on && j < 64) {
          if (j < 6) {
            double sum = ys[6 * (size_t)k + j];
            for (int sl = 0; sl < 6; sl++) sum -= bpart[sl][j];
            vv[j] = sum;
          }
          GP_WAVE_SYNC_LDS();
          if (j == 0) back6<true>(Ls + 36 * (size_t)base, dis + 6 * (size_t)k, vv, xs + 6 * (size_t)k);
        }
        __syncthreads();
      }
    }
  }
  GP_SMALL_STAMP(3);
  // ---- phase 3 ----
  for (int i = t; i < 6 * V.P; i += kSmallThreads) {
    const double v = xs[i];
    const size_t to = 6 * (size_t)V.perm[i / 6] + i % 6;
    V.x_slots[to] = v;
    V.x_slots_host[to] = v;
    if (V.has_epi) ys[to] = v;  // (y is done with: the epilogue reads x in slot order from here)
  }
  if (V.has_epi) {
    // the trial's poses (gp_lm.hip: lm_poses_kernel's work) while this workgroup still has x at hand: one launch less behind the step
    __syncthreads();
    const int n = max(V.epi.F, V.epi.N);
    for (int i = t; i < n; i += kSmallThreads) lm_poses_thread(V.epi, i, ys, bad != 0);
  }
  if (t == 0) {
    *status = bad;
    *V.status_host = (double)bad;
  }
  GP_SMALL_STAMP(4);
  if (V.trace && t == 0)
    for (int i = 0; i < 64; i++) V.trace[i] = tr_lds[i];
#undef GP_SMALL_STAMP
}

}  // namespace gp

struct gp_sparse_system {
  gp::SparseSymbolic sym;
  int num_factors = 0, n = 0;
  hipStream_t stream = nullptr;
  std::vector<gp::SparseDest> dests;
  std::vector<gp::SparseContribution> contribs;
  gp::DeviceArray d_dests, d_contribs, d_int, L, y, x, x_slots, c, status, prior, dinv, diag0;
  gp::PinnedArray pinned;  // gp_sparse_system_step: x [n] | b [n] | c | status, written by the step's kernels, read by the host behind ONE synchronisation
  gp::SparseView view{};
  const int* d_perm = nullptr;
  bool built = false;
  // the one-launch step of small graphs (sparse_small_step_kernel): eligible when factor + index lists fit one compute unit's LDS
  bool small_ok = false, one_launch = false;
  bool step_in_flight = false;        // gp_sparse_system_issue_step went out, gp_sparse_system_finish_step has not collected it
  std::vector<double> prior_staged;  // the permuted prior diagonal of the step in flight (source of its H2D copy)
  size_t small_lds_bytes = 0;
  gp::DeviceArray d_small_arena, d_level_ptr;
  gp::SparseSmallView small{};
};


extern "C" {

int gp_sparse_symbolic(int num_slots, const int* factor_slots, int num_factors, int ordering, int* perm_out, int* parent_out, int64_t* nnz_a_blocks, int64_t* nnz_l_blocks,
                       int* num_subtrees, int* top_columns) {
  if (num_slots <= 0 || num_factors < 0 || (num_factors > 0 && !factor_slots) || ordering < 0 || ordering > 4)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_symbolic: bad arguments (ordering: 0 = natural, 1 = nested dissection, 2 = minimum degree, 3 = minimum degree with slack, 4 = automatic)");
  gp::SparseSymbolic S;
  GP_TRY(gp::sparse_symbolic(num_slots, factor_slots, num_factors, ordering, &S));
  if (perm_out) memcpy(perm_out, S.perm.data(), sizeof(int) * (size_t)num_slots);
  if (parent_out) memcpy(parent_out, S.parent.data(), sizeof(int) * (size_t)num_slots);
  if (nnz_a_blocks) *nnz_a_blocks = S.nnzA;
  if (nnz_l_blocks) *nnz_l_blocks = S.colptr[num_slots];
  if (num_subtrees) *num_subtrees = S.num_subtrees;
  if (top_columns) *top_columns = num_slots - S.work_ptr[S.num_subtrees];
  return GP_OK;
}

// the schedule of the numeric phase (pure host code): number of launch levels (level 0 = the subtrees) and the critical path in columns, i.e. the
// sum over the levels of the longest work list of the level -- what one solve walks sequentially
int gp_sparse_symbolic_schedule(int num_slots, const int* factor_slots, int num_factors, int ordering, int* num_levels, int* critical_columns, int* num_lists) {
  if (num_slots <= 0 || num_factors < 0 || (num_factors > 0 && !factor_slots) || ordering < 0 || ordering > 4)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_symbolic_schedule: bad arguments");
  gp::SparseSymbolic S;
  GP_TRY(gp::sparse_symbolic(num_slots, factor_slots, num_factors, ordering, &S));
  const int levels = (int)S.level_ptr.size() - 1;
  if (num_levels) *num_levels = levels;
  if (critical_columns) *critical_columns = gp::critical_columns(S);
  if (num_lists) *num_lists = S.level_ptr[levels];
  return GP_OK;
}

// host-side check hook (no device needed): the work lists of the numeric phase -- per list its level, its number of columns, the block products its columns gather in
// all and the most a single column gathers; arrays of `capacity` ints (num_lists from gp_sparse_symbolic_schedule)
int gp_debug_sparse_work_lists(int num_slots, const int* factor_slots, int num_factors, int ordering, int capacity, int* level, int* columns, int* products, int* max_column_products) {
  if (num_slots <= 0 || num_factors < 0 || (num_factors > 0 && !factor_slots) || ordering < 0 || ordering > 4 || !level || !columns || !products || !max_column_products)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_sparse_work_lists: bad arguments");
  gp::SparseSymbolic S;
  GP_TRY(gp::sparse_symbolic(num_slots, factor_slots, num_factors, ordering, &S));
  const int levels = (int)S.level_ptr.size() - 1;
  if (S.level_ptr[levels] > capacity) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_sparse_work_lists: capacity too small");
  for (int l = 0; l < levels; l++)
    for (int w = S.level_ptr[l]; w < S.level_ptr[l + 1]; w++) {
      level[w] = l;
      columns[w] = S.work_ptr[w + 1] - S.work_ptr[w];
      int sum = 0, mx = 0;
      for (int q = S.work_ptr[w]; q < S.work_ptr[w + 1]; q++) {
        const int k = S.work_cols[q];
        const int n = S.upd_ptr[S.colptr[k + 1]] - S.upd_ptr[S.colptr[k]];
        sum += n;
        mx = std::max(mx, n);
      }
      products[w] = sum;
      max_column_products[w] = mx;
    }
  return GP_OK;
}

int gp_sparse_system_create(int num_slots, const int* factor_slots, int num_factors, int ordering, gp_stream_t stream, gp_sparse_system_t** out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_create: null out");
  *out = nullptr;
  if (num_slots <= 0 || num_factors < 0 || (num_factors > 0 && !factor_slots) || ordering < 0 || ordering > 4)
    return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_create: factor_slots = [num_factors][2] (target, source; < 0 = fixed), ordering 0 .. 4");
  auto s = std::make_unique<gp_sparse_system>();
  GP_TRY(gp::sparse_symbolic(num_slots, factor_slots, num_factors, ordering, &s->sym));
  const gp::SparseSymbolic& S = s->sym;
  const int P = num_slots, nnzL = S.colptr[P];
  s->num_factors = num_factors;
  s->n = 6 * P;
  s->stream = (hipStream_t)stream;
  // destination block -> ordered contribution list (factor order = summation order)
  std::map<int, std::vector<gp::SparseContribution>> lists;
  auto block_of = [&](int i, int k) {
    if (i == k) return S.colptr[k];
    const int* b = S.rowidx.data() + S.colptr[k] + 1;
    const int* e = S.rowidx.data() + S.colptr[k + 1];
    return (int)(std::lower_bound(b, e, i) - S.rowidx.data());
  };
  for (int p = 0; p < nnzL; p++) lists[p];  // every block of L is a destination: diagonal blocks carry b, fill blocks get an empty list and are written as zeros (no memset per build)
  for (int f = 0; f < num_factors; f++) {
    const int st = factor_slots[2 * f], ss = factor_slots[2 * f + 1];
    const int it = st >= 0 ? S.iperm[st] : -1, is = ss >= 0 ? S.iperm[ss] : -1;
    if (it >= 0) lists[S.colptr[it]].push_back({f, gp::STAKE_HT});
    if (is >= 0) lists[S.colptr[is]].push_back({f, gp::STAKE_HS});
    if (it >= 0 && is >= 0) {
      if (it > is) {
        lists[block_of(it, is)].push_back({f, gp::STAKE_HTS});    // row = target, col = source
      } else {
        lists[block_of(is, it)].push_back({f, gp::STAKE_HTS_T});  // row = source, col = target
      }
    }
  }
  std::vector<int> diag_of((size_t)nnzL, -1);
  for (int k = 0; k < P; k++) diag_of[S.colptr[k]] = k;
  for (auto& kv : lists) {
    gp::SparseDest d;
    d.block = kv.first;
    d.diag_col = diag_of[kv.first];
    d.begin = (int)s->contribs.size();
    d.count = (int)kv.second.size();
    s->contribs.insert(s->contribs.end(), kv.second.begin(), kv.second.end());
    s->dests.push_back(d);
  }
  // one device arena for all index arrays
  const std::vector<int>* arrs[] = {&S.colptr, &S.rowidx, &S.upd_ptr, &S.upd_a, &S.upd_b, &S.row_ptr, &S.row_blk, &S.row_col, &S.work_ptr, &S.work_cols, &S.perm, &S.upd_bpos};
  size_t total = 0;
  for (const auto* a : arrs) total += a->size() + 1;
  std::vector<int> packed;
  packed.reserve(total);
  size_t offs[12];
  for (int i = 0; i < 12; i++) {
    offs[i] = packed.size();
    packed.insert(packed.end(), arrs[i]->begin(), arrs[i]->end());
    packed.push_back(0);
  }
  int rc = GP_OK;
  if ((rc = s->d_int.alloc(sizeof(int) * packed.size())) || (rc = s->d_dests.alloc(sizeof(gp::SparseDest) * s->dests.size())) ||
      (rc = s->d_contribs.alloc(sizeof(gp::SparseContribution) * std::max<size_t>(s->contribs.size(), 1))) || (rc = s->L.alloc(sizeof(double) * 36 * (size_t)nnzL)) ||
      (rc = s->y.alloc(sizeof(double) * (size_t)s->n)) || (rc = s->x.alloc(sizeof(double) * (size_t)s->n)) || (rc = s->x_slots.alloc(sizeof(double) * (size_t)s->n)) ||
      (rc = s->c.alloc(sizeof(double))) || (rc = s->status.alloc(sizeof(int))) || (rc = s->prior.alloc(sizeof(double) * (size_t)s->n)) ||
      (rc = s->dinv.alloc(sizeof(double) * (size_t)s->n)) || (rc = s->diag0.alloc(sizeof(double) * (size_t)s->n)))
    return rc;
  GP_HIP(hipMemcpy(s->d_int.ptr, packed.data(), sizeof(int) * packed.size(), hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(s->d_dests.ptr, s->dests.data(), sizeof(gp::SparseDest) * s->dests.size(), hipMemcpyHostToDevice));
  if (!s->contribs.empty()) GP_HIP(hipMemcpy(s->d_contribs.ptr, s->contribs.data(), sizeof(gp::SparseContribution) * s->contribs.size(), hipMemcpyHostToDevice));
  const int* base = s->d_int.as<int>();
  s->view.colptr = base + offs[0];
  s->view.rowidx = base + offs[1];
  s->view.upd_ptr = base + offs[2];
  s->view.upd_a = base + offs[3];
  s->view.upd_b = base + offs[4];
  s->view.row_ptr = base + offs[5];
  s->view.row_blk = base + offs[6];
  s->view.row_col = base + offs[7];
  s->view.work_ptr = base + offs[8];
  s->view.work_cols = base + offs[9];
  s->d_perm = base + offs[10];
  s->view.upd_bpos = base + offs[11];
  s->view.L = s->L.as<double>();
  s->view.y = s->y.as<double>();
  s->view.x = s->x.as<double>();
  s->view.dinv = s->dinv.as<double>();
  s->view.diag0 = s->diag0.as<double>();
  s->view.status = s->status.as<int>();
  // the one-launch step (small graphs): its own copy of the index lists it walks, in the order of SparseSmallView's offsets, and the levels
  size_t small_words = 0;
  s->small_lds_bytes = gp::small_step_lds_bytes(S, &small_words);
  if (s->small_lds_bytes) {
    std::vector<int> blkcol((size_t)nnzL);
    for (int k = 0; k < P; k++)
      for (int q = S.colptr[k]; q < S.colptr[k + 1]; q++) blkcol[(size_t)q] = k;
    std::vector<int> list_maxnb(S.work_ptr.size() - 1, 0);
    for (size_t l = 0; l + 1 < S.work_ptr.size(); l++)
      for (int w = S.work_ptr[l]; w < S.work_ptr[l + 1]; w++) list_maxnb[l] = std::max(list_maxnb[l], S.colptr[S.work_cols[w] + 1] - S.colptr[S.work_cols[w]]);
    const std::vector<int>* sa[] = {&S.colptr, &S.rowidx, &S.upd_ptr, &S.upd_a, &S.upd_b, &S.row_ptr, &S.row_blk, &S.row_col, &S.work_ptr, &S.work_cols, &blkcol, &list_maxnb};
    std::vector<int> arena;
    arena.reserve(small_words);
    int off[12];
    for (int i = 0; i < 12; i++) {
      off[i] = (int)arena.size();
      arena.insert(arena.end(), sa[i]->begin(), sa[i]->end());
    }
    if (arena.size() & 1) arena.push_back(0);
    if (arena.size() > small_words) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_create: internal: the one-launch step's index arena outgrew its estimate");
    if ((rc = s->d_small_arena.alloc(sizeof(int) * std::max<size_t>(arena.size(), 1))) || (rc = s->d_level_ptr.alloc(sizeof(int) * S.level_ptr.size()))) return rc;
    GP_HIP(hipMemcpy(s->d_small_arena.ptr, arena.data(), sizeof(int) * arena.size(), hipMemcpyHostToDevice));
    GP_HIP(hipMemcpy(s->d_level_ptr.ptr, S.level_ptr.data(), sizeof(int) * S.level_ptr.size(), hipMemcpyHostToDevice));
    gp::SparseSmallView& V = s->small;
    V.L_global = s->L.as<double>();
    V.y_global = s->y.as<double>();
    V.diag0_global = s->diag0.as<double>();
    V.arena = s->d_small_arena.as<int>();
    V.level_ptr = s->d_level_ptr.as<int>();
    V.arena_words = (int)arena.size(), V.num_levels = (int)S.level_ptr.size() - 1, V.P = P, V.nnzL = nnzL;
    V.o_colptr = off[0], V.o_rowidx = off[1], V.o_upd_ptr = off[2], V.o_upd_a = off[3], V.o_upd_b = off[4], V.o_row_ptr = off[5], V.o_row_blk = off[6], V.o_row_col = off[7];
    V.o_work_ptr = off[8], V.o_work_cols = off[9], V.o_blkcol = off[10], V.o_list_maxnb = off[11];
    V.perm = s->d_perm;
    V.wave_columns = 1;
    V.x_slots = s->x_slots.as<double>();
    // (more than 64 KB of dynamic LDS per workgroup has to be asked for; a runtime that refuses leaves the multi-launch step in charge)
    // (the attribute belongs to the FUNCTION, not to this system: every system asks for the same ceiling, so that one created later never lowers what an earlier one needs)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gp::sparse_small_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess) {
      s->small_ok = true;
      s->one_launch = true;
    } else {
      (void)hipGetLastError();
    }
  }
  *out = s.release();
  return GP_OK;
}

// the one-launch step (sparse_small_step_kernel): 1 = gp_sparse_system_step uses it when the system qualifies (default), 0 = the multi-launch form; returns what the
// next step will run (1 / 0).  The two are bit-identical; the switch exists for the test that says so and for A/B timing.
// measurement hook: the one-launch step's thread 0 leaves shader-clock stamps in dev_buffer (64 uint64; SparseSmallView::trace); null = off
int gp_debug_sparse_step_trace(gp_sparse_system_t* s, unsigned long long* dev_buffer) {
  if (!s) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_debug_sparse_step_trace: null");
  s->small.trace = dev_buffer;
  return GP_OK;
}

int gp_sparse_system_set_one_launch(gp_sparse_system_t* s, int enable) {
  if (!s) return 0;
  s->one_launch = enable != 0 && s->small_ok;
  s->small.wave_columns = enable == 2 ? 0 : (enable == 3 ? 3 : 1);  // 3: a list per (lone) wave whatever the number of lists  // 2: the one-launch step's first form (teams of waves in lock step) -- kept for the bit-identity test and A/B timing
  return s->one_launch ? (s->small.wave_columns == 3 ? 3 : (s->small.wave_columns ? 1 : 2)) : 0;
}

int gp_sparse_system_destroy(gp_sparse_system_t* s) {
  if (!s) return GP_OK;
  (void)hipStreamSynchronize(s->stream);
  delete s;
  return GP_OK;
}

int gp_sparse_system_size(const gp_sparse_system_t* s) { return s ? s->n : 0; }

int gp_sparse_system_info(const gp_sparse_system_t* s, int64_t* nnz_a_blocks, int64_t* nnz_l_blocks, int64_t* block_products, int* num_subtrees, int* top_columns) {
  if (!s) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_info: null");
  const gp::SparseSymbolic& S = s->sym;
  if (nnz_a_blocks) *nnz_a_blocks = S.nnzA;
  if (nnz_l_blocks) *nnz_l_blocks = S.colptr[S.P];
  if (block_products) *block_products = (int64_t)S.upd_a.size();
  if (num_subtrees) *num_subtrees = S.num_subtrees;
  if (top_columns) *top_columns = S.P - S.work_ptr[S.num_subtrees];
  return GP_OK;
}

int gp_sparse_system_build(gp_sparse_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                           const double* prior_diag_host) {
  if (!s || (!records_dev && s->num_factors > 0) || !(lambda >= 0.0)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_build: bad arguments");
  const gp::SparseSymbolic& S = s->sym;
  // (no memsets: the destination list covers every block of L and every entry of y)
  hipLaunchKernelGGL(gp::sparse_assemble_kernel<false>, dim3((unsigned)s->dests.size()), dim3(64), 0, s->stream, s->d_dests.as<gp::SparseDest>(),
                     s->d_contribs.as<gp::SparseContribution>(), reinterpret_cast<const double*>(records_dev), s->L.as<double>(), s->y.as<double>(), [&] {
                       gp::SparseStepExtras e0{};
                       e0.diag0 = s->diag0.as<double>();
                       return e0;
                     }());
  hipLaunchKernelGGL(gp::sparse_sum_errors_kernel, dim3(1), dim3(256), 0, s->stream, reinterpret_cast<const double*>(records_dev), s->num_factors, s->c.as<double>());
  const double* prior = nullptr;
  std::vector<double> permuted;
  if (prior_diag_host) {
    permuted.resize((size_t)s->n);
    for (int k = 0; k < S.P; k++)
      for (int r = 0; r < 6; r++) permuted[6 * (size_t)k + r] = prior_diag_host[6 * (size_t)S.perm[k] + r];
    GP_HIP(hipMemcpyAsync(s->prior.ptr, permuted.data(), sizeof(double) * (size_t)s->n, hipMemcpyHostToDevice, s->stream));
    prior = s->prior.as<double>();
  }
  if (lambda > 0.0 || prior) {
    hipLaunchKernelGGL(gp::sparse_damp_kernel, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, s->L.as<double>(), s->view.colptr, s->n, lambda, diagonal_damping,
                       min_diagonal, max_diagonal, prior, s->diag0.as<double>());
  }
  GP_HIP(hipGetLastError());
  if (prior_diag_host) GP_HIP(hipStreamSynchronize(s->stream));  // `permuted` goes away
  s->built = true;
  return GP_OK;
}

// the system as a dense symmetric column-major [n][n] in SLOT order (for checkers), b [n] (slot order), c
int gp_sparse_system_download(const gp_sparse_system_t* s, double* A_host, double* b_host, double* c_host) {
  if (!s || !s->built) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_download: build the system first");
  const gp::SparseSymbolic& S = s->sym;
  const size_t n = (size_t)s->n;
  GP_HIP(hipStreamSynchronize(s->stream));
  if (A_host) {
    std::vector<double> L(36 * (size_t)S.colptr[S.P]);
    GP_HIP(hipMemcpy(L.data(), s->L.ptr, sizeof(double) * L.size(), hipMemcpyDeviceToHost));
    std::fill(A_host, A_host + n * n, 0.0);
    for (int k = 0; k < S.P; k++)
      for (int p = S.colptr[k]; p < S.colptr[k + 1]; p++) {
        const int si = S.perm[S.rowidx[p]], sk = S.perm[k];
        for (int c = 0; c < 6; c++)
          for (int r = 0; r < 6; r++) {
            const double v = L[36 * (size_t)p + 6 * c + r];  // A(6 i + r, 6 k + c)
            if (p == S.colptr[k] && r < c) continue;          // diagonal block: lower triangle holds the data
            A_host[(6 * (size_t)sk + c) * n + 6 * (size_t)si + r] = v;
            A_host[(6 * (size_t)si + r) * n + 6 * (size_t)sk + c] = v;
          }
      }
  }
  if (b_host) {
    std::vector<double> y(n);
    GP_HIP(hipMemcpy(y.data(), s->y.ptr, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int k = 0; k < S.P; k++)
      for (int r = 0; r < 6; r++) b_host[6 * (size_t)S.perm[k] + r] = y[6 * (size_t)k + r];
  }
  if (c_host) GP_HIP(hipMemcpy(c_host, s->c.ptr, sizeof(double), hipMemcpyDeviceToHost));
  return GP_OK;
}

// the numeric phase on the system's stream: factorisation + forward substitution level by level, backward substitution the same levels in reverse
static void launch_factor_and_substitutions(gp_sparse_system_t* s) {
  const gp::SparseSymbolic& S = s->sym;
  // one launch per level of the schedule (level 0: the subtrees; then the chains of separator columns, level by level), the backward
  // substitution the same levels in reverse
  const int levels = (int)S.level_ptr.size() - 1;
  for (int l = 0; l < levels; l++) {
    const int first = S.level_ptr[l], count = S.level_ptr[l + 1] - first;
    // 256 threads for the subtrees (columns of a few blocks), 1024 for the chains of separator columns (a dozen blocks each).  Measured on the 512-pose
    // band graph (profiles/r03_solver_time.txt): this form 1.34 ms; 512 threads + batches of eight products 1.45-1.48 ms with or without the next
    // batch's indices prefetched; 1024 threads + batches of eight 2.8 ms (the 128-register cap spills the batch)
    if (count > 0 && l == 0) hipLaunchKernelGGL(gp::sparse_factor_kernel<256>, dim3(count), dim3(256), 0, s->stream, s->view, first);
    if (count > 0 && l > 0) {
      constexpr bool staged = true;  // separator chains through the LDS-staged kernel (the per-entry gather for them as well: 1.45 vs 0.93 ms, profiles/r03_solver_time.txt)
      if (staged) hipLaunchKernelGGL(gp::sparse_factor_staged_kernel<1024>, dim3(count), dim3(1024), 0, s->stream, s->view, first);
      else hipLaunchKernelGGL(gp::sparse_factor_kernel<1024>, dim3(count), dim3(1024), 0, s->stream, s->view, first);
    }
  }
  for (int l = levels - 1; l >= 0; l--) {
    const int first = S.level_ptr[l], count = S.level_ptr[l + 1] - first;
    if (count > 0) hipLaunchKernelGGL(gp::sparse_backsolve_kernel, dim3(count), dim3(64), 0, s->stream, s->view, first);
  }
}

// SparseLinearSolver::solve(A, b): A x = b by block-sparse LL^T.  The assembled blocks are overwritten by the factor (build again
// before the next solve).  x in slot order.
int gp_sparse_system_solve(gp_sparse_system_t* s, double* x_host, double* x_dev_out) {
  if (!s || !s->built) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_solve: build the system first");
  const gp::SparseSymbolic& S = s->sym;
  GP_HIP(hipMemsetAsync(s->status.ptr, 0, sizeof(int), s->stream));
  launch_factor_and_substitutions(s);
  hipLaunchKernelGGL(gp::sparse_unpermute_kernel, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, (const double*)s->x.as<double>(), s->d_perm, S.P, s->x_slots.as<double>());
  GP_HIP(hipGetLastError());
  s->built = false;
  int h_status = 0;
  GP_HIP(hipMemcpyAsync(&h_status, s->status.ptr, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  if (x_dev_out) GP_HIP(hipMemcpyAsync(x_dev_out, s->x_slots.ptr, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->stream));
  if (x_host) GP_HIP(hipMemcpyAsync(x_host, s->x_slots.ptr, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost, s->stream));
  GP_HIP(hipStreamSynchronize(s->stream));
  if (h_status != 0) return gp::fail(GP_ERROR_INDETERMINATE, "gp_sparse_system_solve: the system is not positive definite (indeterminate linear system)");
  return GP_OK;
}

// One damped step of the optimizer's inner loop in ONE stream-ordered pass and ONE synchronisation: buildDampedSystem + solve (levenberg_marquardt_ext.cpp:146-161, 200-220),
// i.e. gp_sparse_system_build, gp_sparse_system_download(b, c) and gp_sparse_system_solve without the two waits and the four copies between them.  2 + 2 x levels launches:
// the assembly (damping applied as the diagonal is written; b, c stored where the host reads them; status cleared), the levels, x (slot order) + status to the host.
// Bit-identical to the three calls.  b_host / c_host are valid also when the system turns out indeterminate (the optimizer raises lambda and tries again).
// the step's device work, queued on the system's stream: nothing waits here
static int issue_step_impl(gp_sparse_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                           const double* prior_diag_host, const gp::LmPoseView* epi) {
  if (!s || (!records_dev && s->num_factors > 0) || !(lambda >= 0.0)) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_step: bad arguments");
  const gp::SparseSymbolic& S = s->sym;
  const size_t n = (size_t)s->n;
  GP_TRY(s->pinned.ensure(sizeof(double) * (2 * n + 2)));
  double* h = s->pinned.as<double>();
  gp::SparseStepExtras ex{};
  ex.lambda = lambda, ex.min_diag = min_diagonal, ex.max_diag = max_diagonal, ex.diagonal = diagonal_damping;
  if (prior_diag_host) {
    s->prior_staged.resize(n);  // (outlives the copy: the next issue waits for the step in flight first -- gp_sparse_system_finish_step)
    for (int k = 0; k < S.P; k++)
      for (int r = 0; r < 6; r++) s->prior_staged[6 * (size_t)k + r] = prior_diag_host[6 * (size_t)S.perm[k] + r];
    GP_HIP(hipMemcpyAsync(s->prior.ptr, s->prior_staged.data(), sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
    ex.prior_diag = s->prior.as<double>();
  }
  ex.perm = s->d_perm, ex.b_slots_host = h + n, ex.c_dev = s->c.as<double>(), ex.c_host = h + 2 * n, ex.status = s->status.as<int>();
  ex.num_factors = s->num_factors, ex.num_dests = (int)s->dests.size();
  ex.diag0 = s->diag0.as<double>();
  hipLaunchKernelGGL(gp::sparse_assemble_kernel<true>, dim3((unsigned)s->dests.size() + 1), dim3(64), 0, s->stream, s->d_dests.as<gp::SparseDest>(),
                     s->d_contribs.as<gp::SparseContribution>(), reinterpret_cast<const double*>(records_dev), s->L.as<double>(), s->y.as<double>(), ex);
  if (s->one_launch) {
    // small graph: the assembly as it is (one workgroup per block of L across the chip: a gather of 8-byte values, which ONE compute unit's vector memory path takes
    // 40 us for), then factorisation and both substitutions in ONE launch with every operand in the LDS of one compute unit (sparse_small_step_kernel)
    gp::SparseSmallView V = s->small;
    V.x_slots_host = h;
    V.status_host = h + 2 * n + 1;
    V.has_epi = epi ? 1 : 0;
    if (epi) V.epi = *epi;
    hipLaunchKernelGGL(gp::sparse_small_step_kernel, dim3(1), dim3(gp::kSmallThreads), s->small_lds_bytes, s->stream, V, s->status.as<int>());
  } else {
    launch_factor_and_substitutions(s);
    hipLaunchKernelGGL(gp::sparse_step_end_kernel, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, (const double*)s->x.as<double>(), s->d_perm, S.P, s->x_slots.as<double>(), h,
                       (const int*)s->status.as<int>(), h + 2 * n + 1);
  }
  GP_HIP(hipGetLastError());
  s->built = false;
  s->step_in_flight = true;
  return GP_OK;
}

int gp_sparse_system_issue_step(gp_sparse_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                                const double* prior_diag_host) {
  return issue_step_impl(s, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, prior_diag_host, nullptr);
}

// waits for the stream and hands over what the step's kernels left in the pinned buffer
static int finish_step(gp_sparse_system_t* s, double* x_host, double* b_host, double* c_host, bool wait) {
  if (!s || !s->step_in_flight) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_finish_step: no step was issued");
  const size_t n = (size_t)s->n;
  const double* h = s->pinned.as<double>();
  s->step_in_flight = false;
  if (wait) GP_HIP(hipStreamSynchronize(s->stream));
  if (b_host) memcpy(b_host, h + n, sizeof(double) * n);
  if (c_host) *c_host = h[2 * n];
  if (h[2 * n + 1] != 0.0) return gp::fail(GP_ERROR_INDETERMINATE, "gp_sparse_system_step: the system is not positive definite (indeterminate linear system)");
  if (x_host) memcpy(x_host, h, sizeof(double) * n);
  return GP_OK;
}

int gp_sparse_system_finish_step(gp_sparse_system_t* s, double* x_host, double* b_host, double* c_host) { return finish_step(s, x_host, b_host, c_host, true); }
// ... for a caller that has SEEN the stream pass the step (a completion word of work it queued behind the step on the same stream): no wait of its own
int gp_sparse_system_collect_step(gp_sparse_system_t* s, double* x_host, double* b_host, double* c_host) { return finish_step(s, x_host, b_host, c_host, false); }

int gp_sparse_system_step(gp_sparse_system_t* s, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                          const double* prior_diag_host, double* x_host, double* b_host, double* c_host) {
  GP_TRY(gp_sparse_system_issue_step(s, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, prior_diag_host));
  return gp_sparse_system_finish_step(s, x_host, b_host, c_host);
}

// where an issued step leaves its result ON THE DEVICE: x in slot order and the status word (!= 0: indeterminate, x is not to be used) -- for a consumer queued
// behind the step on the same stream (the device-side retract of gp_lm.hip)
int gp_sparse_system_device_solution(gp_sparse_system_t* s, const double** x_slots_dev, const int** status_dev) {
  if (!s) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_sparse_system_device_solution: null system");
  if (x_slots_dev) *x_slots_dev = s->x_slots.as<double>();
  if (status_dev) *status_dev = s->status.as<int>();
  return GP_OK;
}

}  // extern "C"

namespace gp {
int sparse_issue_step_with_poses(gp_sparse_system_t* sys, const gp_linearized6* records_dev, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                                 const LmPoseView& poses, bool* fused) {
  *fused = sys && sys->one_launch;
  return issue_step_impl(sys, records_dev, lambda, diagonal_damping, min_diagonal, max_diagonal, nullptr, *fused ? &poses : nullptr);
}
}  // namespace gp
