"""Build the three pybind11 shims of this directory against torch and libbevfusion_amd.so — the modules a maintainer of the
reference would import as `bev_pool_ext`, `voxel_layer` and `sparse_conv_ext` — into integration/_build/<name>/<name>.so
(git-ignored; it travels to the GPU box with the snapshot like the library itself), and, when /root/reference is present, stage
the reference's OWN Python wrappers of the three ops (ops/bev_pool/bev_pool.py, ops/voxel/voxelize.py,
ops/spconv/{ops,functional,structure}.py) next to them under integration/_build/refpy/ (git-ignored as well: no reference source
enters the repository) so that tests/test_gpu_shims.py can run the reference's Python over the drop-in modules.

    python -m integration.build_shims [--force]

Plain g++ (no ninja, no hipify: the shims are host C++ that only calls the C ABI); the HIP stream of the current torch stream
comes from c10_hip."""
import importlib.util
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_build")
REF_OPS = "/root/reference/mmdet3d/ops"
SHIMS = {"bev_pool_ext": "bev_pool_ext_shim.cpp", "voxel_layer": "voxel_layer_shim.cpp", "sparse_conv_ext": "sparse_conv_ext_shim.cpp"}
# reference wrapper files -> package layout under _build/refpy (each package gets an empty __init__.py; the extension
# module of the package is injected by the test before the wrapper is imported)
REF_PY = {"bev_pool": ["bev_pool.py"], "voxel": ["voxelize.py"], "spconv": ["ops.py", "functional.py", "structure.py"]}


def so_path(name):
    return os.path.join(OUT, name, name + ".so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_one(name, force=False, verbose=False):
    import torch
    from torch.utils import cpp_extension

    src = os.path.join(HERE, SHIMS[name])
    lib = os.path.join(ROOT, "bevfusion_amd", "lib", "libbevfusion_amd.so")
    if not os.path.exists(lib):
        raise RuntimeError(f"{lib} missing: build the library first (python -m bevfusion_amd.build)")
    out = so_path(name)
    deps = [src, os.path.join(HERE, "shim_common.h"), os.path.join(ROOT, "include", "bevfusion_amd.h"), lib]
    if not force and not _newer(out, deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = cpp_extension.include_paths() + [sysconfig.get_paths()["include"], os.path.join(ROOT, "include"), HERE]
    try:
        import pybind11

        inc.append(pybind11.get_include())
    except ImportError:
        pass
    if os.path.isdir("/opt/rocm/include"):
        inc.append("/opt/rocm/include")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", f"-DTORCH_EXTENSION_NAME={name}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for d in inc:
        cmd += ["-I", d]
    cmd += [src, "-o", out, f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
            f"-L{os.path.dirname(lib)}", "-lbevfusion_amd", "-lbevfusion_amd_ext",   # ext: sparse max pooling / dynamic scatter exports
            f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN/../../../bevfusion_amd/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on {src}:\n{r.stderr[-4000:]}")
    return out


def stage_reference_python(force=False):
    """Copy the reference's Python wrappers into _build/refpy/<pkg>/ (build scratch, git-ignored).  Returns the directory, or
    None when /root/reference is not there (the GPU box: it uses what the CPU container staged)."""
    dst_root = os.path.join(OUT, "refpy")
    if not os.path.isdir(REF_OPS):
        return dst_root if os.path.isdir(dst_root) else None
    for pkg, files in REF_PY.items():
        d = os.path.join(dst_root, "ref_" + pkg)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "__init__.py"), "w") as f:
            f.write("")
        for fn in files:
            src = os.path.join(REF_OPS, pkg, fn)
            if force or _newer(os.path.join(d, fn), [src]):
                shutil.copyfile(src, os.path.join(d, fn))
    return dst_root


def build_all(force=False, verbose=False):
    built = {name: build_one(name, force=force, verbose=verbose) for name in SHIMS}
    built["refpy"] = stage_reference_python(force=force)
    return built


def load_shim(name):
    """Import a built shim as the Python module `name` (what `from . import bev_pool_ext` finds in the reference)."""
    import torch  # noqa: F401  (libtorch / libc10_hip first)

    path = so_path(name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run python -m integration.build_shims")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    for k, v in build_all(force="--force" in sys.argv, verbose="-v" in sys.argv).items():
        print(k, "->", v)
