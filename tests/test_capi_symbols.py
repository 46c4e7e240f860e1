"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import glob
import os
import re

import pytest

from bevfusion_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="bevfusion_amd.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef BEVAMD_PROFILING.*?#endif", "", txt, flags=re.S)   # hooks of rejected experiments: profiling builds only
    return set(re.findall(r"\b(bevamd_\w+)\s*\(", txt))


def test_header_declares_something():
    assert len(_declared()) >= 10
    assert sorted(os.path.basename(h) for h in glob.glob(os.path.join(ROOT, "include", "*.h"))) == ["bevfusion_amd.h", "bevfusion_amd_ext.h"]


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    missing = [n for n in sorted(_declared()) if not hasattr(lib, n)]
    assert not missing, f"declared in include/bevfusion_amd.h but not exported: {missing}"
    ext = ctypes.CDLL(_capi.EXT_LIB_PATH)
    missing = [n for n in sorted(_declared("bevfusion_amd_ext.h")) if not hasattr(ext, n)]
    assert not missing, f"declared in include/bevfusion_amd_ext.h but not exported: {missing}"


def test_hot_library_carries_only_the_hot_path():
    """The exports of the reference's pybind modules OUTSIDE SURVEY.md §8's path (sparse max pooling, dynamic scatter) live in the
    optional ext library: the hot library does not define them, the one namespace `_capi.load()` hands out still resolves them."""
    out = os.popen(f"nm -D --defined-only {_capi.LIB_PATH}").read()
    assert "bevamd_bev_pool_forward_cells" in out
    assert not [n for n in _declared("bevfusion_amd_ext.h") if f" {n}\n" in out]
    lib = _capi.load()
    assert lib.bevamd_dynamic_scatter_workspace_bytes(1000) > 0 and lib.bevamd_scan_workspace_bytes(1000) > 0


def test_binding_table_matches_header():
    assert set(_capi.exported_names()) == _declared()
    assert set(_capi.ext_exported_names()) == _declared("bevfusion_amd_ext.h")
    _capi.load()  # resolves every symbol with its argtypes


def test_error_string_api():
    assert isinstance(_capi.last_error(), str)


def test_workspace_queries_run_without_gpu():
    lib = _capi.load()
    assert lib.bevamd_bev_pool_prepare_workspace_bytes(1000, 1, 1, 10, 10) > 2 * 4 * 1000
    assert lib.bevamd_radix_sort_workspace_bytes(1 << 20) >= (1 << 20) // 4096 * 256 * 4
    assert lib.bevamd_scan_workspace_bytes(1 << 20) > 0


def test_invalid_arguments_are_rejected_before_any_gpu_work():
    lib = _capi.load()
    rc = lib.bevamd_bev_pool_forward(None, None, None, None, None, 10, 0, 1, 1, 1, 1, 1, None)
    assert rc == 1
    assert "bad sizes" in _capi.last_error()


def test_hip_entry_points_refuse_cpu_tensors():
    """The HIP ops have no CPU path: the extension mirror, the plan and QuickCumsumCuda raise on host tensors.  (`bev_pool()`
    itself routes HOST tensors to the reference's device-agnostic torch QuickCumsum — BASELINE configs[0],
    tests/test_cpu_plumbing.py — which is a separate algorithm for host data, never a fallback for GPU tensors.)"""
    import torch

    from bevfusion_amd.bev_pool import BevPoolPlan, QuickCumsumCuda, bev_pool_ext

    x, c = torch.zeros(4, 8), torch.zeros(4, 4, dtype=torch.int32)
    iv = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        bev_pool_ext.bev_pool_forward(x, c, iv, iv, 1, 1, 2, 2)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        BevPoolPlan.from_coords(c, 1, 1, 2, 2)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        QuickCumsumCuda.apply(x, c, torch.zeros(4, dtype=torch.long), 1, 1, 2, 2)


def test_host_side_tables_of_the_slab_kernels():
    """Pure host logic of the staged-rows convolution family: variant codes (LDS-filter, 1xxxxxx register-filter, 2xxxxxx
    persistent, 4xxxxxx filter-stationary: 32 channels, 64-row blocks with baked slots), their block sizes, the 16-bit slot bound of
    a grid, the metadata sizes."""
    lib = _capi.load()
    for cin in (32, 64, 128):
        codes = (ctypes.c_int * 64)()
        n = lib.bevamd_spconv_slab_variants(cin, codes, 64)
        got = [codes[i] for i in range(n)]
        assert n >= 3 and len(set(got)) == n
        regw = [v for v in got if 1000000 <= v < 2000000]
        pers = [v for v in got if 2000000 <= v < 3000000]
        fstat = [v for v in got if v >= 4000000]
        # every PLAIN shape is built in both flavours; shapes with kernel flags (ID >= 4: baked slot metadata) or 64-row blocks
        # (baked metadata of the filter-stationary format) have no persistent twin
        plain = [v for v in regw if v % 10 < 4 and (v // 100 % 10) * 16 * (v // 1000 % 10) != 64]
        assert regw and sorted(v + 1000000 for v in plain) == sorted(pers)
        assert bool(fstat) == (cin == 32) and all(lib.bevamd_spconv_slab_block_rows(cin, v) == 64 for v in fstat)
        for v in got:
            code = lib.bevamd_spconv_slab_block_rows(cin, v)
            rows, fmt = code & 0xFFFF, code >> 16                                        # upper half: slot format (1 = baked 128-byte rows)
            assert rows in (128, 256) or v in fstat or (rows == 64 and v in regw)
            assert fmt == (1 if (1000000 <= v < 2000000 and v % 10 >= 8) else 0)
            if 1000000 <= v < 3000000:                                                   # RW * 16 * MT from the code itself
                assert rows == (v // 100 % 10) * 16 * (v // 1000 % 10)
                twin = lib.bevamd_spconv_slab_block_rows(cin, v % 1000000 + 2000000)
                assert twin == (rows if v % 1000000 + 1000000 in plain else 0)            # the persistent twin: same blocks
        assert lib.bevamd_spconv_slab_block_rows(cin, 0) == lib.bevamd_spconv_slab_block_rows(cin, got[0])
        assert lib.bevamd_spconv_slab_block_rows(cin, 1999999) == 0            # not built
    assert lib.bevamd_spconv_slab_block_rows(48, 0) == 0
    # the narrow-row kernels (cin <= 16): 256- | 128-row blocks; 31xxxxx = the 16 -> 32 layer with both output tiles in one wave
    for cin in (5, 8, 16):
        assert [lib.bevamd_spconv_slab_block_rows(cin, v) for v in (0, 3000256, 3000128, 3000064)] == [256, 256, 128, 0]
        assert [lib.bevamd_spconv_slab_block_rows(cin, v) for v in (3100256, 3100128)] == ([256, 128] if cin > 8 else [0, 0])
        n = lib.bevamd_spconv_slab_variants(cin, codes, 64)
        assert [codes[i] for i in range(n)] == [3000256, 3000128] + ([3100256, 3100128] if cin > 8 else [])
    shape = (ctypes.c_int * 3)(720, 720, 21)
    assert lib.bevamd_spconv_slab_grid_ok(shape, 256) == 1                     # 256 + 722 * 21 + 2 rows fit 16-bit slots
    wide = (ctypes.c_int * 3)(720, 4000, 21)
    assert lib.bevamd_spconv_slab_grid_ok(wide, 128) == 0
    assert lib.bevamd_spconv_slab_block_rows(64, 4000112) == 0                 # the filter-stationary kernels exist for 32 channels only
    assert lib.bevamd_spconv_slab_hdr_bytes(1000, 128) == 8 * 3 * 8
    assert lib.bevamd_spconv_slab_slot_bytes(1000, 128) == 8 * 27 * 128 * 2
    assert lib.bevamd_spconv_slab_slot_bytes(1000, 128 | 1 << 16) == 8 * 27 * 128 * 2   # a format code does not change the sizes
    assert lib.bevamd_spconv_slab_hdr_bytes(1000, 128 | 1 << 16) == 8 * 3 * 8 and lib.bevamd_spconv_slab_grid_ok(shape, 256 | 1 << 16) == 1
    assert lib.bevamd_spconv_slab_ablation_mask() == 0                         # shipped builds compile nothing out


def test_batch_entry_points_validate_their_host_arrays():
    """Segment tables of the batched voxelizer / segmented sort are host arrays: sizes and limits are checked before any
    GPU work (workspace queries return 0 for a table they would reject)."""
    lib = _capi.load()
    ok = (ctypes.c_int * 3)(1000, 0, 5000)
    assert lib.bevamd_radix_sort_segmented_workspace_bytes(ok, 3) >= 6 * 512 * 4
    assert lib.bevamd_voxelize_mean_batch_workspace_bytes(ok, 3) >= 5 * 6000 * 4
    neg = (ctypes.c_int * 2)(10, -1)
    assert lib.bevamd_radix_sort_segmented_workspace_bytes(neg, 2) == 0
    assert lib.bevamd_voxelize_mean_batch_workspace_bytes(neg, 2) == 0
    many = (ctypes.c_int * 65)(*([1] * 65))
    assert lib.bevamd_radix_sort_segmented_workspace_bytes(many, 65) == 0      # at most 64 segments per launch
    assert lib.bevamd_voxelize_mean_batch_workspace_bytes(None, 1) == 0
    vs, cr = (ctypes.c_float * 3)(0.5, 0.5, 0.5), (ctypes.c_float * 6)(0, 0, 0, 10, 10, 2)
    rc = lib.bevamd_voxelize_mean_batch(None, None, 1, 5, vs, cr, 10, 100, 1, None, None, None, None, None, None, 0, None)
    assert rc != 0 and "host arrays" in _capi.last_error()
    one = (ctypes.c_int * 1)(4)
    ptrs = (ctypes.c_void_p * 1)(None)
    rc = lib.bevamd_voxelize_mean_batch(ptrs, one, 1, 2, vs, cr, 10, 100, 1, None, None, None, None, None, None, 0, None)
    assert rc != 0 and "num_features" in _capi.last_error()


def test_column_pooling_tile_leaves_room_for_two_workgroups_per_compute_unit():
    """Pass 1 of the fused column pooling sizes its dynamic LDS from the shape (context rows of 4 image columns + a depth tile of DH
    bins + the run metadata of the tile).  Two workgroups share a compute unit's 160 KiB only if one takes at most 80 KiB: a third
    metadata array once added 720 bytes too many and halved the occupancy (round 5, EXPERIMENTS C.8) — this is the guard that
    needs no GPU (tests/test_gpu_fused_columns.py asks the runtime's occupancy calculator)."""
    lib = _capi.load()
    fh, c, dh = 32, 80, 60                                            # the flagship tile: 118 depth bins in two tiles of 60
    want = (fh * 4 * c + (dh // 4) * fh * 20 + 9 * dh) * 4            # context + depth (pitch 20 floats) + 2 words per column + 1 per bin
    got = lib.bevamd_bev_pool_fused_columns_lds_bytes(c, 118, fh, 88)
    assert got == want == 81520 and 2 * got <= 160 * 1024
    assert lib.bevamd_bev_pool_fused_columns_lds_bytes(c, 118, 33, 88) == 0      # unsupported: more rows than mask bits
    assert lib.bevamd_bev_pool_fused_columns_lds_bytes(c, 8, 4, 8) == (4 * 4 * c + 2 * 4 * 20 + 9 * 8) * 4   # a tile shorter than DH
