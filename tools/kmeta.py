"""Register / scratch / LDS use of the kernels in a built libmorb_hip*.so (no GPU needed): python tools/kmeta.py [lib] [name-substring ...]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "morb_slam_amd", "libmorb_hip.so")
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
blob = open(fat, "rb").read()
starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
for i, st in enumerate(starts):
    en = starts[i + 1] if i + 1 < len(starts) else len(blob)
    bun, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"c{i}.o")
    open(bun, "wb").write(blob[st:en])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + bun, "--output=" + co])
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if args and not any(a in name for a in args):
            continue
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
