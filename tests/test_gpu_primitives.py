"""GPU: exclusive scan and stable radix sort (bevfusion_amd/csrc/primitives.hip) vs numpy."""
import numpy as np
import pytest
import torch

from bevfusion_amd import _capi

pytestmark = pytest.mark.gpu


def _scan(x, dev):
    lib = _capi.load()
    n = x.shape[0]
    t = torch.from_numpy(x.astype(np.int32)).to(dev)
    out = torch.empty_like(t)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.bevamd_scan_workspace_bytes(n)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    rc = lib.bevamd_exclusive_scan_u32(_capi.ptr(t), _capi.ptr(out), n, _capi.ptr(total), _capi.ptr(ws), wsb,
                                       _capi.stream_ptr(dev))
    _capi.check(rc, "scan")
    return out.cpu().numpy().astype(np.int64), int(total.item())


@pytest.mark.parametrize("n", [0, 1, 63, 64, 1023, 1024, 1025, 2048, 5000, 131072, 131073, 1 << 20, 3000001])
def test_exclusive_scan(dev, n):
    rng = np.random.default_rng(n)
    x = rng.integers(0, 5, n).astype(np.int64)
    got, total = _scan(x, dev)
    exp = np.cumsum(x) - x
    assert np.array_equal(got, exp)
    assert total == int(x.sum())


@pytest.mark.parametrize("n", [0, 1, 63, 2047, 2048, 2049, 5000, 131072, 131073, 1 << 20, 3000001, 40000003])
def test_single_pass_scan(dev, n):
    """One launch: tiles chained by decoupled look-back (csrc/single_pass.h); in place, repeated on the same state buffer."""
    lib = _capi.load()
    rng = np.random.default_rng(n + 7)
    x = rng.integers(0, 5, n).astype(np.int64)
    if n > 100000:
        x[rng.integers(0, n, 100)] = 1 << 20      # sums beyond 16 bits inside single tiles
    sb = lib.bevamd_scan_single_pass_state_bytes(n)
    state = torch.full((max(sb, 8),), 0xAB, dtype=torch.uint8, device=dev)   # dirty: the entry zeroes it
    exp = np.cumsum(x) - x
    for rep in range(3):
        t = torch.from_numpy(x.astype(np.int32)).to(dev)
        total = torch.full((1,), -1, dtype=torch.int32, device=dev)
        rc = lib.bevamd_exclusive_scan_u32_single_pass(_capi.ptr(t), _capi.ptr(t), n, _capi.ptr(total), _capi.ptr(state), sb,
                                                       _capi.stream_ptr(dev))
        _capi.check(rc, "single-pass scan")
        assert np.array_equal(t.cpu().numpy().astype(np.int64), exp)
        assert int(total.item()) == int(x.sum())


def _sort(keys, vals, nbits, dev):
    lib = _capi.load()
    n = keys.shape[0]
    k = torch.from_numpy(keys.astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
    v = torch.from_numpy(vals.astype(np.uint32).view(np.int32)).to(dev)
    ko, vo = torch.empty_like(k), torch.empty_like(v)
    wsb = lib.bevamd_radix_sort_workspace_bytes(n)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    rc = lib.bevamd_radix_sort_pairs_u32(_capi.ptr(k), _capi.ptr(v), _capi.ptr(ko), _capi.ptr(vo), n, nbits,
                                         _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
    _capi.check(rc, "radix sort")
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n,nbits", [(1, 8), (64, 3), (4095, 17), (4096, 17), (4097, 20), (100000, 27),
                                     (1000003, 32), (2000000, 18), (50000, 1), (70000, 9)])
def test_radix_sort_is_a_stable_sort(dev, n, nbits):
    rng = np.random.default_rng(n + nbits)
    hi = (1 << nbits) - 1
    # few distinct keys on purpose: long runs of ties exercise stability
    keys = rng.integers(0, min(hi, 5000) + 1, n).astype(np.uint32)
    if nbits == 32:
        keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        keys[::3] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = _sort(keys, vals, nbits, dev)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ko, keys[order])
    assert np.array_equal(vo, vals[order])


def test_radix_sort_all_equal_and_already_sorted(dev):
    n = 300000
    ko, vo = _sort(np.full(n, 7, np.uint32), np.arange(n, dtype=np.uint32), 8, dev)
    assert np.all(ko == 7) and np.array_equal(vo, np.arange(n, dtype=np.uint32))
    keys = np.arange(n, dtype=np.uint32)
    ko, vo = _sort(keys, keys[::-1].copy(), 19, dev)
    assert np.array_equal(ko, keys) and np.array_equal(vo, keys[::-1])


@pytest.mark.parametrize("counts,nbits", [([5000, 0, 1, 1024, 3000], 13), ([1], 8), ([0, 0, 70000], 27),
                                          ([2048] * 64, 9), ([310000] * 8, 27), ([1023, 1025], 32),
                                          # the one-sweep passes' lanes: 8 uneven segments, 9 with empties, 64 ragged ones, one
                                          # segment far larger than the rest, a single large one (one lane, > 2048 tiles)
                                          ([300000, 12, 250000, 0, 310000, 99999, 1025, 400000], 27),
                                          ([90000, 12, 70000, 0, 110000, 9999, 1025, 100000], 27),
                                          ([0, 70000, 0, 0, 1, 2, 3, 1024, 500000], 20),
                                          ([(37 * i * i) % 9000 for i in range(64)], 18),
                                          ([10] * 7 + [2000000] + [10] * 8, 27), ([3000000], 27)])
def test_segmented_radix_sort_sorts_every_segment_on_its_own(dev, counts, nbits):
    import ctypes

    lib = _capi.load()
    rng = np.random.default_rng(len(counts) + nbits)
    n = sum(counts)
    keys = rng.integers(0, min((1 << nbits) - 1, 3000) + 1, n).astype(np.uint32)
    if nbits == 32:
        keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum(counts)])
    vals = np.concatenate([np.arange(c, dtype=np.uint32) for c in counts]) if n else np.zeros(0, np.uint32)
    k = torch.from_numpy(keys.view(np.int32)).to(dev)
    v = torch.from_numpy(vals.view(np.int32)).to(dev)
    ko, vo = torch.empty_like(k), torch.empty_like(v)
    cn = (ctypes.c_int * len(counts))(*counts)
    wsb = lib.bevamd_radix_sort_segmented_workspace_bytes(cn, len(counts))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    rc = lib.bevamd_radix_sort_pairs_u32_segmented(_capi.ptr(k), _capi.ptr(v), _capi.ptr(ko), _capi.ptr(vo), cn, len(counts),
                                                   nbits, _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
    _capi.check(rc, "segmented radix sort")
    ko, vo = ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)
    for s in range(len(counts)):
        seg = slice(off[s], off[s + 1])
        order = np.argsort(keys[seg], kind="stable")
        assert np.array_equal(ko[seg], keys[seg][order]) and np.array_equal(vo[seg], vals[seg][order])
