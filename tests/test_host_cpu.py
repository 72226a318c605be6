"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/vofe.h
declares (no compute calls without a GPU), the product path fails loudly without a GPU, and the
FrameState / preset mirrors keep the reference's contract."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vofe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vofe_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from visual_odometry_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libvofe.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(L, name), "libvofe.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared
    L.vofe_version.restype = ctypes.c_int
    assert L.vofe_version() == 100
    L.vofe_last_error.restype = ctypes.c_char_p
    assert isinstance(L.vofe_last_error(), bytes)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from visual_odometry_amd import _lib
    L = _lib.load()
    ctx = ctypes.c_void_p()
    rc = L.vofe_create(0, None, ctypes.byref(ctx))
    assert rc != 0 and not ctx.value
    assert b"no HIP device" in L.vofe_last_error() or b"HIP error" in L.vofe_last_error()
    from visual_odometry_amd.frontend import FrontEnd
    with pytest.raises(RuntimeError):
        FrontEnd(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "visual-odometry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "vofe_cpu_" not in src, f


def test_framestate_contract():
    from visual_odometry_amd import FrameState
    ok = dict(landmarks_image=np.zeros((2, 5)), landmarks_world=np.zeros((3, 5)),
              cand_landmarks_image_current=np.zeros((2, 4)), cand_landmarks_image_first=np.zeros((2, 4)),
              cand_landmarks_transform=np.zeros((12, 4)))
    s = FrameState(**ok)
    assert "5 landmarks" in str(s) and "4 candidate" in str(s)
    FrameState(**{**ok, "cand_landmarks_transform": np.zeros((16, 4))})
    for key, bad in (("landmarks_image", np.zeros((3, 5))), ("landmarks_world", np.zeros((3, 6))),
                     ("cand_landmarks_image_current", np.zeros((2, 3))), ("cand_landmarks_image_first", np.zeros(4)),
                     ("cand_landmarks_transform", np.zeros((13, 4)))):
        with pytest.raises(ValueError):
            FrameState(**{**ok, key: bad})


def test_presets_match_reference_tables():
    from visual_odometry_amd import TRACKING_PRESETS, BOOTSTRAP_PRESETS, Constants
    assert TRACKING_PRESETS["parking"]["winSize"] == (11, 11) and TRACKING_PRESETS["kitti"]["winSize"] == (21, 21)
    assert all(p["maxLevel"] == 3 and p["minEigThreshold"] == 0.001 for p in TRACKING_PRESETS.values())
    assert BOOTSTRAP_PRESETS["malaga"]["winSize"] == (41, 41) and BOOTSTRAP_PRESETS["malaga"]["maxCorners"] == 400
    assert all(p["maxLevel"] == 0 for p in BOOTSTRAP_PRESETS.values())
    assert Constants.THRESHOLD_NEW_KEYPOINTS == 15


def test_synthetic_is_seeded():
    from visual_odometry_amd import synthetic
    a = synthetic.sequence(40, 50, 7, 2)
    b = synthetic.sequence(40, 50, 7, 2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert a[0].dtype == np.uint8 and a[0].shape == (40, 50) and a[0].std() > 20
    p = synthetic.pad_points(None, 100, 40, 50, 1)
    assert p.shape == (100, 2) and p.dtype == np.float32


def test_c_host_example_compiles_as_c99_and_links():
    """include/vofe.h is plain C (gcc -std=c99 -pedantic -Werror) and every entry point the C host program uses resolves
    against libvofe.so; the program itself runs in the GPU suite."""
    import __graft_entry__ as g
    exe = g.build_example()
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


def test_build_refuses_kernels_that_spill_to_scratch():
    """build.py reads the compiler's per-kernel resource remarks and refuses a library in which any kernel uses scratch
    memory (the parser, on the remark format of this toolchain)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vofe_build", os.path.join(ROOT, "visual-odometry_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    text = """lk.hip:1050:1: remark: Function Name: _ZN12_GLOBAL__N_118lk_track_kernel_x2ILi2EEEvNS_6LKArgsE [-Rpass-analysis=kernel-resource-usage]
lk.hip:1050:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]
lk.hip:1050:1: remark:     ScratchSize [bytes/lane]: 128 [-Rpass-analysis=kernel-resource-usage]
lk.hip:431:1: remark: Function Name: other_kernel [-Rpass-analysis=kernel-resource-usage]
lk.hip:431:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
"""
    rep = b.scratch_report(text)
    assert rep == {"_ZN12_GLOBAL__N_118lk_track_kernel_x2ILi2EEEvNS_6LKArgsE": 128, "other_kernel": 0}
    assert b.scratch_report("") == {}


def test_isa_audit_finds_the_partial_write_hazard_and_the_wide_reload():
    """tools/isa_audit.py is the build gate for the inline-assembly SDWA shift (build.py) and the aid of the spill study
    (DESIGN.md section 8): it must flag a reader directly behind a partial register write, accept one behind a wait
    state, and flag a spill reload that runs under a wider EXEC region than its store."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    sdwa = "\tv_ashrrev_i32_sdwa v5, v1, v2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
    n, sites = audit.hazard_sites(sdwa + "\tv_dot2_i32_i16 v7, v5, v9, v7\n")
    assert n == 1 and len(sites) == 1
    n, sites = audit.hazard_sites(sdwa + "\tv_dot2c_i32_i16_e32 v5, v8, v9\n")          # destination is read too
    assert len(sites) == 1
    n, sites = audit.hazard_sites(sdwa + "\tv_add_u32_e32 v6, v3, v4\n\tv_dot2_i32_i16 v7, v5, v9, v7\n")
    assert n == 1 and not sites
    n, sites = audit.hazard_sites(sdwa + "\ts_nop 0\n\tv_dot2_i32_i16 v7, v5, v9, v7\n")
    assert not sites
    n, sites = audit.hazard_sites(sdwa + "\tv_pk_add_f32 v[10:11], v[4:5], v[6:7]\n")     # register range
    assert len(sites) == 1
    kern = ("\nk:\n\ts_and_saveexec_b64 s[4:5], vcc\n\tscratch_store_dword off, v3, off offset:8\n\ts_or_b64 exec, exec, s[4:5]\n"
            "\tscratch_load_dword v3, off, off offset:8\n\t.end_amdhsa_kernel\n")
    out, bad = audit.spill_report(kern, "k")
    assert bad == 1 and len(out) == 2
    kern_ok = kern.replace("\ts_or_b64 exec, exec, s[4:5]\n\tscratch_load", "\tscratch_load") 
    assert audit.spill_report(kern_ok, "k")[1] == 0
