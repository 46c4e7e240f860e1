"""Build the gfx950 shared library `bevfusion_amd/lib/libbevfusion_amd.so`.

hipcc cross-compiles without a GPU.  The library has a plain C ABI
(include/bevfusion_amd.h) and links only against the HIP runtime: no torch,
no pybind.  Objects are rebuilt only when their sources (the unit and the headers its depfile lists) are newer.

    python -m bevfusion_amd.build [--force] [--verbose] [--profiling]

`--profiling` (or BEVAMD_PROFILING=1) adds -DBEVAMD_PROFILING: the ablation instantiations of the tiled convolution that
tools/sweep_spconv.py --ablate times (kernels with parts compiled out, wrong results by design) — never in the shipped build.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libbevfusion_amd.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-result",
    "-fno-gpu-rdc",
]


EXT_CSRC = os.path.join(CSRC, "ext")
EXT_LIB = os.path.join(LIBDIR, "libbevfusion_amd_ext.so")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _ext_sources():
    """csrc/ext/: exports of the reference's pybind modules OUTSIDE the hot path (sparse max pooling, dynamic scatter), built into
    their own library so that libbevfusion_amd.so carries only what the path runs."""
    if not os.path.isdir(EXT_CSRC):
        return []
    return sorted(os.path.join(EXT_CSRC, f) for f in os.listdir(EXT_CSRC) if f.endswith(".hip"))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(obj, src):
    """Prerequisites of `obj`: what hipcc's own depfile of the last build lists (the headers this unit really includes), every
    header of csrc/ when there is none yet."""
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep):
        return [src] + _headers()
    with open(dep) as fh:
        words = fh.read().replace("\\\n", " ").split()
    files = [w for w in words[1:] if w.startswith(CSRC)]
    if any(not os.path.exists(f) for f in files):
        return None          # a header was removed or renamed: stale
    return [src] + files


def _compile_one(src, force, verbose, extra=()):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    deps = _deps(obj, src)
    if not force and deps is not None and not _newer(obj, deps):
        return obj, False
    cmd = [HIPCC] + CFLAGS + list(extra) + ["-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj, True


def build(force=False, verbose=False, profiling=None):
    os.makedirs(OBJDIR, exist_ok=True)
    if profiling is None:
        profiling = os.environ.get("BEVAMD_PROFILING", "0") == "1"
    extra = ["-DBEVAMD_PROFILING"] if profiling else []
    stamp = os.path.join(OBJDIR, ".profiling")
    if os.path.exists(stamp) != bool(profiling):      # flavour changed: every object is stale
        force = True
        if profiling:
            open(stamp, "w").close()
        elif os.path.exists(stamp):
            os.remove(stamp)
    srcs = _sources()
    if not srcs:
        raise RuntimeError("no .hip sources found in " + CSRC)
    ext_srcs = _ext_sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs) + len(ext_srcs))) as ex:
        results = list(ex.map(lambda s: _compile_one(s, force, verbose, extra), srcs))
        ext_results = list(ex.map(lambda s: _compile_one(s, force, verbose, list(extra) + [f"-I{CSRC}"]), ext_srcs))
    objs = [o for o, _ in results]
    manifest = os.path.join(OBJDIR, ".linked")           # the object list of the last link: a unit that moved or vanished relinks
    stale = not os.path.exists(manifest) or open(manifest).read().split() != [os.path.basename(o) for o in objs]
    if any(changed for _, changed in results) or not os.path.exists(LIB) or force or stale:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(manifest, "w") as fh:
            fh.write("\n".join(os.path.basename(o) for o in objs))
    ext_objs = [o for o, _ in ext_results]
    if ext_objs and (any(changed for _, changed in ext_results) or not os.path.exists(EXT_LIB) or force):
        # the optional library resolves the shared primitives (scan, sort, error string) from the main one, found next to it
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", EXT_LIB] + ext_objs + [f"-L{LIBDIR}", "-lbevfusion_amd",
                                                                                            "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link of the ext library failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv,
                 profiling=True if "--profiling" in sys.argv else None)
    print(path)
