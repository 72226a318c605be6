"""ctypes binding of libvofe.so (C ABI in include/vofe.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOFE_LIB", os.path.join(_HERE, "libvofe.so"))

# every symbol include/vofe.h declares (tests check the built library exports all of them)
SYMBOLS = (
    "vofe_last_error", "vofe_version", "vofe_create", "vofe_destroy", "vofe_synchronize", "vofe_reload_env",
    "vofe_pyramid_create", "vofe_pyramid_free", "vofe_pyramid_build", "vofe_pyramid_levels",
    "vofe_pyramid_level_size", "vofe_pyramid_download", "vofe_pyramid_download_padded",
    "vofe_lk_track", "vofe_track_fb", "vofe_lk_set_counters", "vofe_profile_begin", "vofe_profile_end", "vofe_corner_score", "vofe_gftt", "vofe_gftt_host", "vofe_gftt_batch", "vofe_gftt_noalloc", "vofe_new_feature_mask", "vofe_new_feature_mask2",
    "vofe_candidate_filter", "vofe_pair_filter", "vofe_triangulate_candidates", "vofe_candidate_commit",
    "vofe_framestate_create", "vofe_framestate_destroy", "vofe_framestate_reset", "vofe_framestate_frame_buffer",
    "vofe_framestate_track", "vofe_framestate_update", "vofe_framestate_update_async", "vofe_framestate_wait", "vofe_framestate_counts", "vofe_framestate_landmarks_world",
    "vofe_framestate_download",
)


class FrameStateParams(C.Structure):
    """vofe_framestate_params of include/vofe.h, field for field."""
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("win_w", C.c_int), ("win_h", C.c_int), ("max_level", C.c_int),
                ("crit_type", C.c_int), ("max_count", C.c_int), ("eps", C.c_double), ("min_eig_thr", C.c_double),
                ("fb_thresh", C.c_float), ("max_corners", C.c_int), ("quality_level", C.c_double), ("min_distance", C.c_double),
                ("thr_new_keypoints", C.c_double), ("thr_triangulation", C.c_double), ("thr_candidates_min", C.c_double),
                ("thr_candidates_max", C.c_double), ("thr_alpha_rad", C.c_double), ("cap_landmarks", C.c_int),
                ("cap_candidates", C.c_int), ("first_is_f64", C.c_int), ("K", C.c_double * 9)]

_lib = None


class error(Exception):
    """Raised where cv2 would raise cv2.error (bad arguments) and on HIP failures."""

    def __init__(self, code, msg):
        super().__init__("vofe error %d: %s" % (code, msg))
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvofe.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, d, f, sz = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t
    ip = C.POINTER(C.c_int)
    L.vofe_last_error.restype = C.c_char_p
    L.vofe_last_error.argtypes = []
    L.vofe_version.argtypes = []
    L.vofe_create.argtypes = [i, vp, C.POINTER(vp)]
    L.vofe_destroy.argtypes = [vp]
    L.vofe_synchronize.argtypes = [vp]
    L.vofe_reload_env.argtypes = [vp]
    L.vofe_pyramid_create.argtypes = [vp, i, i, i, i, i, i, i, C.POINTER(vp)]
    L.vofe_pyramid_free.argtypes = [vp, vp]
    L.vofe_pyramid_build.argtypes = [vp, vp, i, i, vp, i, sz]
    L.vofe_pyramid_levels.argtypes = [vp]
    L.vofe_pyramid_level_size.argtypes = [vp, i, ip, ip]
    L.vofe_pyramid_download.argtypes = [vp, vp, i, i, vp, vp]
    L.vofe_pyramid_download_padded.argtypes = [vp, vp, i, i, i, i, vp]
    L.vofe_lk_track.argtypes = [vp, vp, i, vp, i, vp, vp, i, vp, vp, vp, i, i, i, d, i, d]
    L.vofe_track_fb.argtypes = [vp, vp, i, vp, i, vp, vp, i, vp, vp, vp, i, i, i, d, d, f]
    L.vofe_lk_set_counters.argtypes = [vp, vp]
    L.vofe_profile_begin.argtypes = [vp]
    L.vofe_profile_end.argtypes = [vp, i, vp, vp, ip]
    L.vofe_corner_score.argtypes = [vp, vp, i, i, i, i, sz, i, i, i, d, vp]
    L.vofe_gftt.argtypes = [vp, vp, i, i, i, i, d, d, vp, i, i, i, d, vp, i, ip]
    L.vofe_gftt_noalloc.argtypes = [vp, vp, i, i, i, i, d, d, vp, i, i, i, d, vp, i, ip]
    L.vofe_gftt_host.argtypes = [vp, vp, i, i, i, i, d, d, vp, i, i, i, d, vp, vp, i, ip]
    L.vofe_new_feature_mask.argtypes = [vp, vp, i, vp, i, d, vp]
    L.vofe_new_feature_mask2.argtypes = [vp, vp, i, vp, i, vp, i, d, vp]
    L.vofe_candidate_filter.argtypes = [vp, vp, vp, vp, i, vp, i, i, d, d, d, vp, vp, vp, vp, vp, ip]
    L.vofe_pair_filter.argtypes = [vp, vp, vp, vp, i, d, vp, vp, vp, ip]
    L.vofe_triangulate_candidates.argtypes = [vp, vp, vp, vp, vp, i, vp, i, vp, i, d, vp, vp]
    L.vofe_candidate_commit.argtypes = [vp, vp, vp, i, vp, i, i, i, vp, vp, vp, vp, i, vp, vp, i, i, vp, vp, vp, vp]
    L.vofe_gftt_batch.argtypes = [vp, vp, i, i, i, i, sz, i, d, d, i, i, i, d, vp, i, ip]
    L.vofe_framestate_create.argtypes = [i, C.POINTER(FrameStateParams), C.POINTER(vp)]
    L.vofe_framestate_destroy.argtypes = [vp]
    L.vofe_framestate_reset.argtypes = [vp, vp, i, vp, vp, i, vp, vp, vp, i]
    L.vofe_framestate_frame_buffer.argtypes = [vp, C.POINTER(vp)]
    L.vofe_framestate_track.argtypes = [vp, vp, i, C.POINTER(vp), C.POINTER(vp), ip, ip, ip]
    L.vofe_framestate_update.argtypes = [vp, vp, vp, ip]
    L.vofe_framestate_update_async.argtypes = [vp, vp, vp]
    L.vofe_framestate_wait.argtypes = [vp, ip]
    L.vofe_framestate_counts.argtypes = [vp, ip, ip]
    L.vofe_framestate_landmarks_world.argtypes = [vp, C.POINTER(vp), ip]
    L.vofe_framestate_download.argtypes = [vp, vp, vp, vp, vp]
    for name in SYMBOLS:
        if name != "vofe_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise error(rc, load().vofe_last_error().decode("utf-8", "replace"))
