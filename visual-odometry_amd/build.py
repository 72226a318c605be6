"""Builds libvofe.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "pyramid.hip", "lk.hip", "lk_wide.hip", "corners.hip", "bookkeeping.hip", "framestate.hip"]
# -ffp-contract=off: the float steps must round exactly like the CPU oracle (no fused multiply-add)
# -dot4-insts: without v_dot2c_i32_i16 (accumulator tied to the destination) the compiler selects the three-source
# v_dot2_i32_i16, which costs the same issue time and needs no v_mov to keep the seed of every bilinear sample alive
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
         "-Xclang", "-target-feature", "-Xclang", "-dot4-insts"]


def build_lib(force=False, verbose=False, extra_flags=(), out_name="libvofe.so"):
    """extra_flags / out_name: A/B builds of a tuning macro (e.g. -DLK_X2_RS_PAD=8 -> libvofe_pad8.so, loaded through VOFE_LIB);
    such a build passes the same no-scratch and hazard gates as the shipped library."""
    csrc = os.path.join(_HERE, "csrc")
    out = os.path.join(_HERE, out_name)
    deps = [os.path.join(csrc, s) for s in SOURCES] + [os.path.join(csrc, "vofe_internal.h"),
                                                       os.path.join(_HERE, "..", "include", "vofe.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # The same compile reports every kernel's resources; a kernel that spills registers to scratch memory is refused: none
    # of the shipped kernels needs scratch, and the one experimental build that did (the two-keypoint tracker forced to 128
    # VGPRs) faulted the GPU (profiles/r02_overlap_experiments.txt).
    tmp = out                    # (compiled in place: the offload bundle embeds an id derived from the output name)
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-Rpass-analysis=kernel-resource-usage", "-o", tmp] + [os.path.join(csrc, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    report = scratch_report(r.stderr)
    other = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln and ln.strip()
             and "'-dot4-insts' is not a recognized feature" not in ln       # the host pass ignores the device feature flag
             and not ln.lstrip().startswith("|") and not (ln.split("|")[0].strip().isdigit() and "|" in ln)]
    if other:
        print("\n".join(other))
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    bad = {k: v for k, v in report.items() if v}
    if bad:
        os.remove(tmp)
        raise RuntimeError("kernels that spill to scratch memory (bytes per lane): %r" % bad)
    if not report:
        os.remove(tmp)
        raise RuntimeError("the compiler reported no kernel resources: cannot check for scratch use")
    sites = sdwa_hazard_sites(hipcc, csrc, extra_flags)
    if sites:
        os.remove(tmp)
        raise RuntimeError("lk.hip: a VALU instruction reads a register directly after the inline-assembly SDWA shift wrote "
                           "half of it (no hardware interlock on gfx950); rebuild with -DLK_SDWA_NOP or reorder: %r" % (sites[:3],))
    return out


def sdwa_hazard_sites(hipcc, csrc, extra_flags=()):
    """The one hazard the compiler cannot guard for us: lk.hip's inline-assembly SDWA shift writes half a register, and on
    the gfx940 family a directly following VALU reader of that register is not interlocked (LLVM adds the wait state only
    for SDWA instructions it selected itself).  The tracker source is compiled once more, device side only, to assembly,
    and every partial-write SDWA instruction's successor is checked (tools/isa_audit.py has the same check as a tool)."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(_HERE, "..", "tools", "isa_audit.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "lk.s")
        flags = [f for f in FLAGS if f not in ("-fPIC", "-shared")]
        subprocess.run([hipcc] + flags + list(extra_flags) + ["--cuda-device-only", "-S", "-o", asm, os.path.join(csrc, "lk.hip")],
                       check=True, stderr=subprocess.DEVNULL)
        n, sites = audit.hazard_sites(open(asm).read())
    if n == 0:
        raise RuntimeError("lk.hip: no partial-write SDWA instruction found in the assembly: the hazard check looked at nothing")
    return sites


def scratch_report(stderr_text):
    """{kernel name: scratch bytes per lane} from -Rpass-analysis=kernel-resource-usage remarks."""
    rep, name = {}, None
    for ln in stderr_text.splitlines():
        if "Function Name:" in ln:
            name = ln.split("Function Name:")[1].split("[")[0].strip()
        elif "ScratchSize [bytes/lane]:" in ln and name is not None:
            rep[name] = max(rep.get(name, 0), int(ln.split("ScratchSize [bytes/lane]:")[1].split("[")[0].strip()))
    return rep


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
