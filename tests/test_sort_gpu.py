"""The stable LSD radix sort behind the structure builds (gtsam_points_amd/csrc/gp_sort.hpp: one kernel per pass, offsets from two levels of published
counts) against numpy's stable argsort, through the tune library's test hook: sizes around the tile (4096) and group (32 tiles) boundaries, every pass count,
keys with few distinct values (every tile publishes into the same few digits), already sorted and reversed input."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(gpu, keys, bits):
    import torch

    from gtsam_points_amd import _capi

    lib = _capi.load_tune()
    n = len(keys)
    k = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    ko = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    vo = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    _capi.check(lib.gp_debug_sort_pairs(C.c_void_p(k.data_ptr()), n, bits, C.c_void_p(ko.data_ptr()), C.c_void_p(vo.data_ptr()), None), "gp_debug_sort_pairs")
    return ko.cpu().numpy().view(np.uint32)[:n], vo.cpu().numpy()[:n]


@pytest.mark.parametrize("n", [1, 63, 4095, 4096, 4097, 131071, 131072, 131073, 1_000_003, 4_200_000])
@pytest.mark.parametrize("bits", [7, 8, 9, 22, 25, 31])
def test_radix_sort_is_a_stable_argsort(gpu, n, bits):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** min(bits, 31), size=n, dtype=np.int64).astype(np.uint32)
    got_k, got_v = _sort(gpu, keys, bits)
    want_v = np.argsort(keys, kind="stable")
    assert np.array_equal(got_v, want_v)
    assert np.array_equal(got_k, keys[want_v])


@pytest.mark.parametrize("kind", ["few_values", "one_value", "sorted", "reversed", "top_bit_marker"])
def test_radix_sort_on_structured_keys(gpu, kind):
    n, bits = 700_001, 24
    rng = np.random.default_rng(5)
    if kind == "few_values":
        keys = rng.integers(0, 3, size=n).astype(np.uint32) * 65537
    elif kind == "one_value":
        keys = np.full(n, 0x00ABCDEF, dtype=np.uint32)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 2**bits, size=n)).astype(np.uint32)
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 2**bits, size=n))[::-1].astype(np.uint32).copy()
    else:  # the binning's skipped points: the all-ones key among ordinary keys
        keys = rng.integers(0, 2**bits - 1, size=n).astype(np.uint32)
        keys[rng.integers(0, n, size=1000)] = 2**bits - 1
    got_k, got_v = _sort(gpu, keys, bits)
    want_v = np.argsort(keys, kind="stable")
    assert np.array_equal(got_v, want_v) and np.array_equal(got_k, keys[want_v])


def _sort_ex(gpu, keys, bits, classes, stream=None):
    import torch

    from gtsam_points_amd import _capi

    lib = _capi.load_tune()
    n = len(keys)
    k = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    ko = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    vo = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    fault = C.c_int(-1)
    torch.cuda.synchronize()
    _capi.check(lib.gp_debug_sort_pairs_ex(C.c_void_p(k.data_ptr()), n, bits, C.c_void_p(ko.data_ptr()), C.c_void_p(vo.data_ptr()), classes, C.byref(fault), stream), "gp_debug_sort_pairs_ex")
    return ko.cpu().numpy().view(np.uint32)[:n], vo.cpu().numpy()[:n], fault.value


@pytest.mark.parametrize("n", [4097, 131073, 4_200_000])
def test_one_class_sort_is_the_same_sort(gpu, n):
    """the deadlock-free form of the sort (ONE ticket counter: a tile's predecessors have been drawn by running workgroups whatever order the device
    starts them in) is what a build falls back to; it must give the same stable argsort, also beyond what is resident at once (1026 tiles)"""
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**24, size=n, dtype=np.int64).astype(np.uint32)
    want_v = np.argsort(keys, kind="stable")
    for classes in (1, 32):
        got_k, got_v, fault = _sort_ex(gpu, keys, 24, classes)
        assert fault == 0
        assert np.array_equal(got_v, want_v) and np.array_equal(got_k, keys[want_v])
    _, _, fault = _sort_ex(gpu, keys, 24, -32)  # the test hook: tile 0 of every pass raises the fault word
    assert fault == 1


def test_sort_beside_a_kernel_that_holds_the_cus(gpu):
    """the sort's tiles wait for tiles with smaller indices; with the CUs held by another stream's kernel the device may start few of the sort's workgroups at a time
    (and not necessarily in blockIdx order).  The sort must end -- by itself, or by giving up with the fault word set -- and the one-class form must always sort."""
    import torch

    from gtsam_points_amd import _capi

    tune = _capi.load_tune()
    lib = _capi.load()
    n = 4_200_000
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 2**24, size=n, dtype=np.int64).astype(np.uint32)
    want_v = np.argsort(keys, kind="stable")
    hog, work = C.c_void_p(), C.c_void_p()
    _capi.check(lib.gp_stream_create(C.byref(hog)), "stream")
    _capi.check(lib.gp_stream_create(C.byref(work)), "stream")
    try:
        for classes in (32, 1):
            _capi.check(tune.gp_debug_occupy(30_000.0, 2048, hog), "occupy")  # 8 workgroups of 256 per CU for 30 ms: every wave slot the sort could use is contended
            got_k, got_v, fault = _sort_ex(gpu, keys, 24, classes, stream=work)
            _capi.check(lib.gp_stream_synchronize(hog), "sync")
            if classes == 1:
                assert fault == 0
            if fault == 0:
                assert np.array_equal(got_v, want_v) and np.array_equal(got_k, keys[want_v])
    finally:
        lib.gp_stream_destroy(hog)
        lib.gp_stream_destroy(work)
