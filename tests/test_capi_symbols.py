"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import glob
import os
import re

import pytest

from bevfusion_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(bevamd_\w+)\s*\(", txt))
    return names


def test_header_declares_something():
    assert len(_declared()) >= 10


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in sorted(_declared()) if not hasattr(lib, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_binding_table_matches_header():
    assert set(_capi.exported_names()) == _declared()
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


def test_product_op_refuses_cpu_tensors():
    import torch

    from bevfusion_amd.bev_pool import bev_pool

    with pytest.raises(RuntimeError, match="GPU tensor"):
        bev_pool(torch.zeros(4, 8), torch.zeros(4, 4, dtype=torch.long), 1, 1, 2, 2)
