"""Host logic of the one-sweep segmented sort (csrc/primitives.hip: sort_segs_init): how the tiles of the segments are dealt to the
lanes.  A look-back chain never crosses a segment, a lane hands its tiles out in ticket order — so every lane must own WHOLE
segments, in order, and the lanes must cover every tile exactly once.  Runs without a GPU (the library loads, nothing is launched)."""
import ctypes

import numpy as np
import pytest

from bevfusion_amd import _capi

TILE = 1024   # RS_TILE


def lanes_of(counts):
    lib = _capi.load()
    n = len(counts)
    cn = (ctypes.c_int * n)(*counts)
    tiles = (ctypes.c_uint * (n + 1))()
    lanes = (ctypes.c_uint * 9)()
    k = lib.bevamd_radix_sort_segmented_lanes(cn, n, tiles, lanes)
    return k, list(tiles), list(lanes)


@pytest.mark.parametrize("counts", [[310000] * 8, [1], [0, 0, 70000], [2048] * 64, [300000, 12, 250000, 0, 310000, 99999, 1025, 400000],
                                    [0, 70000, 0, 0, 1, 2, 3, 1024, 500000], [(37 * i * i) % 9000 for i in range(64)],
                                    [10] * 7 + [2000000] + [10] * 8, [0] * 8, [5] * 9, [1025] * 8])
def test_lanes_own_whole_segments_in_order(counts):
    k, tiles, lanes = lanes_of(counts)
    n = len(counts)
    assert k == (8 if n >= 8 else 1)
    assert tiles[0] == 0 and all(tiles[s + 1] - tiles[s] == -(-counts[s] // TILE) for s in range(n))
    total = tiles[n]
    assert lanes[0] == 0 and lanes[8] == total
    assert all(lanes[x] <= lanes[x + 1] for x in range(8))
    if k == 1:
        assert all(v == total for v in lanes[1:])
        return
    assert all(v in tiles for v in lanes)          # every boundary is a segment boundary
    if total:
        biggest = max(tiles[s + 1] - tiles[s] for s in range(n))
        share = -(-total // 8)
        assert max(lanes[x + 1] - lanes[x] for x in range(8)) <= share + biggest   # balanced up to one segment


def test_lanes_reject_bad_arguments():
    lib = _capi.load()
    tiles = (ctypes.c_uint * 3)()
    lanes = (ctypes.c_uint * 9)()
    assert lib.bevamd_radix_sort_segmented_lanes((ctypes.c_int * 2)(5, -1), 2, tiles, lanes) < 0
    assert lib.bevamd_radix_sort_segmented_lanes(None, 2, tiles, lanes) < 0
