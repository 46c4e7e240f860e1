"""CPU: register / scratch budget of the kernels the default inference step launches, read from the gfx950 code objects hipcc
produced (bevfusion_amd/lib/obj/*.o -> .hip_fatbin -> clang-offload-bundler -> llvm-readelf --notes).  No GPU needed: a kernel
that starts spilling, grows a scratch frame or outgrows the registers its occupancy was tuned for shows up here, in the build
check, not as an unexplained slowdown on the GPU box (round 5: 720 bytes of LDS too many halved a kernel's occupancy unnoticed
until it was timed — tests/test_capi_symbols.py guards that one; this file guards the registers)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "bevfusion_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]

# kernel (demangled prefix up to the argument list) -> (max registers (VGPR + AGPR as the metadata counts them), why)
BUDGET = {
    # camera stages
    "void bevamd::bev_pool_fwd_cells_vec_kernel<HIP_vector_type<float, 4u>, 4, 8, 0, true, 1>": (72, "7 waves per SIMD"),
    "void bevamd::bev_pool_fwd_cells_vec_kernel<bevamd::U4, 8, 4, 0, true, 1>": (64, "8 waves per SIMD"),
    "void bevamd::bev_fused_cols_kernel<false>": (128, "two 5-wave workgroups per CU and room to spare"),
    "void bevamd::bev_fused_reduce_kernel<2>": (64, "8 waves per SIMD"),
    # level 1: narrow-row kernels (two 4-wave workgroups per CU by LDS: 2 waves per SIMD)
    "void bevamd::slab::spconv_slabs_kernel<1, 8, 1, 4, 4, 1, 384>": (256, "2 waves per SIMD"),
    "void bevamd::slab::spconv_slabs_kernel<1, 16, 1, 4, 4, 1, 384>": (256, "2 waves per SIMD"),
    "void bevamd::slab::spconv_slabs_kernel<1, 16, 2, 2, 4, 1, 256>": (256, "2 waves per SIMD (both output tiles in one wave: 112 filter registers)"),
    # levels 2-4
    "void bevamd::slab::spconv_slabf2_kernel<1, 128>": (256, "round 6 default for 32 channels: a wave pair splits the input channels, 108 filter registers each, 2 waves per SIMD"),
    "void bevamd::slab::spconv_slabf_kernel<1, 112>": (512, "one wave per SIMD: the whole 27 x 32 x 32 filter in registers"),
    "void bevamd::slab::spconv_slabr_kernel<1, 64, 64, 4, 4, 2, 2, 168, 8>": (256, "2 waves per SIMD"),
    "void bevamd::slab::spconv_slabr_kernel<1, 64, 128, 8, 4, 2, 2, 184, 0>": (256, "2 waves per SIMD"),
    "void bevamd::tile::spconv_stream_kernel<1, 32, 4, 2, 4, 1, 0>": (256, "2 waves per SIMD"),
    "void bevamd::tile::spconv_stream_kernel<1, 64, 8, 2, 4, 2, 0>": (256, "2 waves per SIMD"),
    "void bevamd::tile::spconv_stream_kernel<1, 128, 8, 1, 8, 4, 0>": (256, "2 waves per SIMD"),
}


def _kernels():
    import tempfile

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
            b = os.path.basename(o)[:-2]
            fb, co = os.path.join(tmp, b + ".fb"), os.path.join(tmp, b + ".co")
            if subprocess.run([TOOLS[0], f"--dump-section=.hip_fatbin={fb}", o], capture_output=True).returncode or not os.path.exists(fb):
                continue                                              # a host-only object
            r = subprocess.run([TOOLS[1], "--unbundle", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--output={co}"], capture_output=True)
            if r.returncode or not os.path.exists(co):
                continue
            notes = subprocess.run([TOOLS[2], "--notes", co], capture_output=True, text=True).stdout
            names = []
            for blk in notes.split("    .args:")[1:]:                 # one block per kernel of the amdhsa.kernels list
                def field(k, blk=blk):
                    m = re.search(r"\.%s:\s+(\S+)" % k, blk)
                    return m.group(1) if m else None
                if field("name"):
                    names.append((field("name"), int(field("vgpr_count") or 0), int(field("private_segment_fixed_size") or 0),
                                  int(field("vgpr_spill_count") or 0)))
            if names:
                dem = subprocess.run(["c++filt"] + [n[0] for n in names], capture_output=True, text=True).stdout.splitlines()
                for (_, vg, scratch, spill), d in zip(names, dem):
                    out[re.sub(r"\(.*", "", d)] = dict(regs=vg, scratch=scratch, vgpr_spills=spill, obj=b)
    return out


@pytest.mark.skipif(not glob.glob(os.path.join(OBJ, "*.o")) or not all(os.path.exists(t) for t in TOOLS) or not shutil.which("c++filt"),
                    reason="needs the in-tree objects of bevfusion_amd.build and the ROCm llvm tools")
def test_default_step_kernels_keep_their_register_budget_and_never_touch_scratch():
    k = _kernels()
    assert len(k) > 300, len(k)                                       # the library's kernels were found at all
    missing = [n for n in BUDGET if n not in k]
    assert not missing, missing
    for name, (regs, why) in BUDGET.items():
        got = k[name]
        assert got["scratch"] == 0 and got["vgpr_spills"] == 0, (name, got)
        assert got["regs"] <= regs, (name, got, why)
