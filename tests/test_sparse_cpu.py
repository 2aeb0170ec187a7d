"""CPU tests of the symbolic phase of the block-sparse solver (gp_sparse_symbolic: pure host code of libgtsam_points_hip.so):
elimination order, elimination tree, fill and schedule, against a dense boolean elimination in numpy."""
import numpy as np
import pytest

import gtsam_points_amd as gpa


def _chain_graph(n, closures=(), fixed_first=False):
    slots = [(i, i + 1) for i in range(n - 1)] + list(closures)
    if fixed_first:
        slots = [(-1, 0)] + slots
    return slots


def _boolean_fill(n, slots, perm):
    """number of lower-triangle blocks (incl. diagonal) of the Cholesky factor of the permuted pattern, by dense elimination"""
    ip = np.empty(n, int)
    ip[perm] = np.arange(n)
    M = np.eye(n, dtype=bool)
    for a, b in slots:
        if a >= 0 and b >= 0:
            M[ip[a], ip[b]] = M[ip[b], ip[a]] = True
    nnz_a = int(np.tril(M).sum())
    parent = -np.ones(n, int)
    for k in range(n):
        rows = np.nonzero(M[k + 1 :, k])[0] + k + 1
        if len(rows):
            parent[k] = rows[0]
            M[np.ix_(rows, rows)] = True
    return nnz_a, int(np.tril(M).sum()), parent


def _graph(case):
    rng = np.random.default_rng(5)
    if case == "chain":
        return 512, _chain_graph(512, fixed_first=True)
    if case == "loops":
        n = 300
        return n, _chain_graph(n, closures=[(int(a), int(b)) for a, b in rng.integers(0, n, (40, 2)) if a != b])
    if case == "disconnected":
        return 40, [(i, i + 1) for i in range(0, 15)] + [(i, i + 1) for i in range(20, 39)]  # poses 16..19 isolated
    if case == "band":
        return 512, [(i, i + d) for i in range(512) for d in (1, 2, 7) if i + d < 512]
    w = 12
    return w * w, [(r * w + c, r * w + c + 1) for r in range(w) for c in range(w - 1)] + [(r * w + c, (r + 1) * w + c) for r in range(w - 1) for c in range(w)]


@pytest.mark.parametrize("case", ["chain", "loops", "disconnected", "grid", "band"])
def test_minimum_degree_never_fills_more_than_the_dissection(case):
    """ordering 2 (minimum degree by multiple elimination -- the fill-reducing class of ordering GTSAM hands the reference's solver) against
    ordering 1 (BFS nested dissection): its factor is never larger; ordering 4 (automatic) keeps whichever of 1 / 3 has the shorter critical
    path; the level schedule covers every column exactly once and walks far fewer columns sequentially than there are."""
    n, slots = _graph(case)
    nd, md, md1, auto = (gpa.sparse_symbolic(n, slots, o) for o in (1, 2, 3, 4))
    assert md["nnz_l_blocks"] <= nd["nnz_l_blocks"], (md["nnz_l_blocks"], nd["nnz_l_blocks"])
    assert auto["critical_columns"] == min(nd["critical_columns"], md1["critical_columns"])
    for s in (nd, md, md1, auto):
        assert sorted(s["perm"].tolist()) == list(range(n))
        assert 1 <= s["num_levels"] <= 64 and s["critical_columns"] <= n
    if case in ("chain", "band"):
        assert auto["critical_columns"] <= 64 + n // 8  # separators / rounds side by side: nowhere near one path of n columns


@pytest.mark.parametrize("ordering", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", ["chain", "loops", "disconnected", "grid", "band"])
def test_symbolic_matches_boolean_elimination(case, ordering):
    n, slots = _graph(case)
    s = gpa.sparse_symbolic(n, slots, ordering)
    perm = s["perm"]
    assert sorted(perm.tolist()) == list(range(n))
    if ordering == 0:
        assert perm.tolist() == list(range(n))
    nnz_a, nnz_l, parent = _boolean_fill(n, slots, perm)
    assert s["nnz_a_blocks"] == nnz_a and s["nnz_l_blocks"] == nnz_l
    assert np.array_equal(s["parent"], parent)
    assert all(p == -1 or p > k for k, p in enumerate(s["parent"]))
    assert s["num_subtrees"] >= 1 and 0 <= s["top_columns"] <= n


def test_nested_dissection_makes_chains_shallow_and_keeps_fill_low():
    n = 512
    slots = _chain_graph(n)
    nat, nd = gpa.sparse_symbolic(n, slots, 0), gpa.sparse_symbolic(n, slots, 1)

    def height(parent):
        h = np.zeros(len(parent), int)
        for k in range(len(parent)):  # children precede parents
            if parent[k] >= 0:
                h[parent[k]] = max(h[parent[k]], h[k] + 1)
        return int(h.max())

    assert height(nat["parent"]) == n - 1  # the natural order of a chain is one path
    assert height(nd["parent"]) <= 80       # pieces of <= 8 poses under ~log2(n / 8) separators
    assert nd["nnz_l_blocks"] <= 2.5 * nat["nnz_l_blocks"]
    assert nd["num_subtrees"] >= 32 and nd["top_columns"] <= 64


def test_bad_arguments():
    with pytest.raises(gpa.GPError):
        gpa.sparse_symbolic(4, [(0, 4)], 0)
    with pytest.raises(gpa.GPError):
        gpa.sparse_symbolic(4, [(1, 1)], 0)
    with pytest.raises(gpa.GPError):
        gpa.sparse_symbolic(4, [(0, 1)], 7)
