"""A/B builds of the HIP library (developer tool): python tools/ab_build.py <name> <TU.hip> [-Dflag ...]
compiles ONE translation unit with the extra flags, links it with the product objects of the other units and writes
morb_slam_amd/libmorb_hip_<name>.so (select it with MORB_HIP_LIB=...; git-ignored)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morb_slam_amd import build as b
name, tu, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build_hip()                                   # product objects up to date
objdir = os.path.join(b.CSRC, "_obj")
vobj = os.path.join(b.CSRC, "_obj_ab"); os.makedirs(vobj, exist_ok=True)
obj = os.path.join(vobj, f"{name}.{tu}.o")
base = [f for f in b.HIPCC_FLAGS if f != "-shared"]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + base + flags + ["-c", "-o", obj, os.path.join(b.CSRC, tu)])
others = [o for o in sorted(glob.glob(os.path.join(objdir, "*.hip.o"))) if os.path.basename(o) != tu + ".o"]
out = os.path.join(ROOT, "morb_slam_amd", f"libmorb_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others)
print(out)
