"""Build libmorb_hip.so in-tree with hipcc for gfx950 (explicit command, no JIT cache: the .so travels to the
GPU box with the repo snapshot)."""
import glob
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_DIR)
CSRC = os.path.join(_DIR, "csrc")
OUT = os.path.join(_DIR, "libmorb_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # one rounding convention shared with the oracle: no mul+add contraction (DESIGN.md "FP conventions")
               "-ffp-contract=off",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False, extra_flags=(), out=OUT):
    """One hipcc -c per translation unit (in parallel, objects under csrc/_obj/), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "_obj" + ("_" + "".join(c for c in "".join(extra_flags) if c.isalnum()) if extra_flags else ""))
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + flags + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        res = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(out) or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_test_native():
    nat = os.path.join(ROOT, "tests", "native")
    out = os.path.join(nat, "libqt_host.so")
    src = os.path.join(nat, "qt_host.cc")
    if _stale(out, [src, os.path.join(CSRC, "quadtree.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src])
    return out
