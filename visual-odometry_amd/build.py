"""Builds libvofe.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "pyramid.hip", "lk.hip", "corners.hip", "bookkeeping.hip", "framestate.hip"]
# -ffp-contract=off: the float steps must round exactly like the CPU oracle (no fused multiply-add)
# -dot4-insts: without v_dot2c_i32_i16 (accumulator tied to the destination) the compiler selects the three-source
# v_dot2_i32_i16, which costs the same issue time and needs no v_mov to keep the seed of every bilinear sample alive
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
         "-Xclang", "-target-feature", "-Xclang", "-dot4-insts"]


def build_lib(force=False, verbose=False):
    csrc = os.path.join(_HERE, "csrc")
    out = os.path.join(_HERE, "libvofe.so")
    deps = [os.path.join(csrc, s) for s in SOURCES] + [os.path.join(csrc, "vofe_internal.h"),
                                                       os.path.join(_HERE, "..", "include", "vofe.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", out] + [os.path.join(csrc, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
