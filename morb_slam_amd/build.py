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


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not force and not _stale(OUT, deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_test_native():
    nat = os.path.join(ROOT, "tests", "native")
    out = os.path.join(nat, "libqt_host.so")
    src = os.path.join(nat, "qt_host.cc")
    if _stale(out, [src, os.path.join(CSRC, "quadtree.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src])
    return out
