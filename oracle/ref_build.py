"""ORACLE — test infrastructure only.

Builds the REFERENCE's own native extensions for the hot path from the sources where they lie under
/root/reference, into oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun):

  voxel_layer      mmdet3d/ops/voxel/src/*        CPU + hipified GPU paths
  sparse_conv_ext  mmdet3d/ops/spconv/{src,include}  CPU functors + hipified GPU functors
  bev_pool_ext     mmdet3d/ops/bev_pool/src/*     hipified GPU kernel only (no CPU path exists)
  iou3d_cuda       mmdet3d/ops/iou3d/src/*        hipified GPU kernels only (rotated BEV overlap / IoU / NMS)

No reference source is copied into the repository: sources are staged in a temp dir (hipify writes
`*_hip.*` files next to its inputs and /root/reference is read-only) and only the compiled .so is kept.
They are used to (a) pin the restated oracle and (b) generate golden vectors; the product never
loads them.

    python -m oracle.ref_build            # build everything that is missing
"""
import importlib.util
import os
import shutil
import sys
import tempfile

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

EXTS = {
    "voxel_layer": dict(
        dir="mmdet3d/ops/voxel",
        sources=["src/voxelization.cpp", "src/scatter_points_cpu.cpp", "src/scatter_points_cuda.cu",
                 "src/voxelization_cpu.cpp", "src/voxelization_cuda.cu"],
        include=None,
    ),
    "sparse_conv_ext": dict(
        dir="mmdet3d/ops/spconv",
        sources=["src/all.cc", "src/reordering_cpu.cc", "src/reordering_cuda.cu", "src/indice_cpu.cc",
                 "src/indice_cuda.cu", "src/maxpool_cpu.cc", "src/maxpool_cuda.cu"],
        include="include",
    ),
    "bev_pool_ext": dict(
        dir="mmdet3d/ops/bev_pool",
        sources=["src/bev_pool_cpu.cpp", "src/bev_pool_cuda.cu"],
        include=None,
    ),
    "iou3d_cuda": dict(
        dir="mmdet3d/ops/iou3d",
        sources=["src/iou3d.cpp", "src/iou3d_kernel.cu"],
        include=None,
    ),
}


def so_path(name):
    return os.path.join(OUT, name, name + ".so")


def available(name):
    return os.path.exists(so_path(name))


def build_one(name, force=False, verbose=False):
    if available(name) and not force:
        return so_path(name)
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"{REF_ROOT} not present: cannot build oracle/_ref/{name}")
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils.cpp_extension import load

    spec = EXTS[name]
    stage = tempfile.mkdtemp(prefix=f"bevamd_ref_{name}_")
    try:
        src_root = os.path.join(REF_ROOT, spec["dir"])
        shutil.copytree(os.path.join(src_root, "src"), os.path.join(stage, "src"))
        inc = []
        if spec["include"]:
            shutil.copytree(os.path.join(src_root, spec["include"]), os.path.join(stage, spec["include"]))
            inc = [os.path.join(stage, spec["include"])]
        bdir = os.path.join(OUT, name)
        os.makedirs(bdir, exist_ok=True)
        flags = ["-w", "-std=c++17", "-DWITH_ROCM", "-DWITH_CUDA"]
        load(name=name, sources=[os.path.join(stage, s) for s in spec["sources"]], build_directory=bdir,
             with_cuda=True, extra_cflags=flags, extra_cuda_cflags=flags, extra_include_paths=inc,
             verbose=verbose, is_python_module=False)
    finally:
        shutil.rmtree(stage, ignore_errors=True)
    if not available(name):
        raise RuntimeError(f"build of {name} produced no {so_path(name)}")
    # keep only the shared object (objects / ninja files are build scratch)
    for f in os.listdir(os.path.join(OUT, name)):
        if not f.endswith(".so"):
            p = os.path.join(OUT, name, f)
            shutil.rmtree(p, ignore_errors=True) if os.path.isdir(p) else os.remove(p)
    return so_path(name)


def build_all(force=False, verbose=False):
    built = {}
    for name in EXTS:
        try:
            built[name] = build_one(name, force=force, verbose=verbose)
        except Exception as e:  # one failing checker must not hide the others
            built[name] = e
            print(f"[oracle.ref_build] {name}: FAILED: {e}", file=sys.stderr)
    return built


def load_ref(name):
    """Import a built reference extension as a Python module (pybind11 module named `name`)."""
    import torch  # noqa: F401  (loads libtorch / libamdhip64 first)

    path = so_path(name)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    res = build_all(force="--force" in sys.argv, verbose="-v" in sys.argv)
    for k, v in res.items():
        print(k, "->", v)
