"""Registers / scratch / LDS of the kernels in an object file (or every in-tree object): python tools/kernel_regs.py [obj.o ...] [--grep substr]"""
import glob, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
def kernels(o):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fb, co = os.path.join(tmp, "a.fb"), os.path.join(tmp, "a.co")
        if subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", o], capture_output=True).returncode or not os.path.exists(fb):
            return out
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if not os.path.exists(co):
            return out
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        rows = []
        for blk in notes.split("    .args:")[1:]:
            f = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, None])[1]
            if f("name"):
                rows.append((f("name"), f("vgpr_count"), f("agpr_count"), f("sgpr_count"), f("private_segment_fixed_size"), f("vgpr_spill_count"), f("group_segment_fixed_size")))
        dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
        for r, d in zip(rows, dem):
            out.append((re.sub(r"\(.*", "", d),) + r[1:])
    return out
if __name__ == "__main__":
    args = sys.argv[1:]; pat = None
    if "--grep" in args:
        i = args.index("--grep"); pat = args[i + 1]; del args[i:i + 2]
    objs = args or sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bevfusion_amd", "lib", "obj", "*.o")))
    for o in objs:
        for k in kernels(o):
            if pat is None or pat in k[0]:
                print(f"{k[0][:110]:110s} vgpr {k[1]:>4} agpr {k[2]:>4} sgpr {k[3]:>4} scratch {k[4]:>5} spill {k[5]:>3} lds {k[6]}")
