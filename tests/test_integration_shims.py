"""CPU: the pybind11 shims under integration/ (the reference's three extension modules re-bound over the C ABI) are COMPILED AND
LINKED against the real torch / pybind11 libraries and libbevfusion_amd.so by the committed recipe (integration/build_shims.py,
the same one `__graft_entry__.build()` runs), import as Python modules and export the reference's function names
(bev_pool_cpu.cpp:89-94, voxelization.cpp:6-11, all.cc:21-51).  Calling them needs a GPU: tests/test_gpu_shims.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EXPORTS = {
    "bev_pool_ext": ["bev_pool_forward", "bev_pool_backward"],
    "voxel_layer": ["hard_voxelize", "dynamic_voxelize"],
    "sparse_conv_ext": ["get_indice_pairs_3d", "indice_conv_fp32", "indice_conv_half", "fused_indice_conv_fp32",
                        "fused_indice_conv_half", "indice_conv_backward_fp32", "indice_conv_backward_half"],
}


@pytest.mark.parametrize("name", sorted(EXPORTS))
def test_shim_links_imports_and_exports_the_reference_names(name):
    from integration import build_shims

    path = build_shims.build_one(name)          # no-op when up to date
    assert os.path.exists(path)
    mod = build_shims.load_shim(name)
    for fn in EXPORTS[name]:
        assert callable(getattr(mod, fn)), fn
    # the error behaviour of the reference's CHECK macros: a host tensor is rejected with RuntimeError, before any GPU work
    import torch

    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        if name == "bev_pool_ext":
            mod.bev_pool_forward(x, x.int(), x.int(), x.int(), 1, 1, 1, 1)
        elif name == "voxel_layer":
            mod.dynamic_voxelize(x, x.int(), [1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 1.0, 1.0, 1.0], 3)
        else:
            mod.indice_conv_fp32(x, x, x.int(), x.int(), 4, 0, 0)


def test_reference_python_wrappers_are_staged_outside_history():
    from integration import build_shims

    root = build_shims.stage_reference_python()
    if root is None:
        pytest.skip("/root/reference is not present and nothing was staged")
    assert os.path.exists(os.path.join(root, "ref_bev_pool", "bev_pool.py"))
    ignore = open(os.path.join(ROOT, ".gitignore")).read()
    assert "integration/_build/" in ignore                      # no reference source enters the repository
