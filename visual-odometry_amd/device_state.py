"""Device-resident FrameState: the image-point arrays of the reference's FrameState
(framestate.py:13-78) -- landmarks Pi, candidates Ci / Fi / Ti -- stay in HBM across frames, and one frame of the
reference's front end (continuous.py:86-92 landmark tracking, :375-495 _handle_candidates) is two device phases around
the CPU pose estimation, with one host wait each:

    tracked, status = st.track(img_current)         # phase 1: upload, ONE pyramid, ONE launch for landmarks + candidates,
                                                    #          candidate compaction + distance band, detection (second stream)
    ... P3P-RANSAC / refinement on the CPU (reference continuous.py:166-272) -> pose, inlier mask ...
    st.update(pose, keep)                           # phase 2: de-duplication of the detections, triangulation of matured
                                                    #          candidates, promotion to landmarks, new candidates appended

Point sets never travel: phase 1 returns the K landmark positions / status flags PnP needs (through pinned host
memory the kernels write directly), phase 2 uploads the pose and the surviving landmarks in one copy command and
returns the promoted landmarks (image points + 3-D points) the same way.  The world points Xi live on the host (only
the CPU pose estimation reads them).  Results equal the reference method evaluated with the oracle, array for array.

This class is the numpy-shaped face of the native object `vofe_framestate` (include/vofe.h, csrc/framestate.hip), which
owns the streams, the buffers and the launch order; a frame is two C calls.  Arrays cross this boundary in the
reference's layout ((2,K) / (3,K) / (12,M) column-per-point) and are transposed to the C ABI's point-per-row layout here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import error
from .frontend import _DEFAULT_CRITERIA


class DeviceFrameState:
    def __init__(self, height, width, K, winSize=(21, 21), maxLevel=3, criteria=_DEFAULT_CRITERIA, minEigThreshold=1e-3,
                 fb_thresh=1.0, max_corners=1000, quality_level=0.01, min_distance=10, threshold_new_keypoints=15,
                 threshold_triangulation=15, threshold_candidates_min=2, threshold_candidates_max=100, threshold_alpha_deg=1,
                 cap_landmarks=8192, cap_candidates=8192, first_dtype=np.float64, frontend=None, device=0):
        self.L = _lib.load()
        if frontend is not None:                  # only its device matters: the native object has its own contexts
            device = frontend.device.index
        self.h, self.w = int(height), int(width)
        self.fdt = np.dtype(first_dtype)
        if self.fdt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise error(-1, "first observations must be float32 or float64")
        p = _lib.FrameStateParams()
        p.height, p.width = self.h, self.w
        p.win_w, p.win_h, p.max_level = int(winSize[0]), int(winSize[1]), int(maxLevel)
        p.crit_type, p.max_count, p.eps = int(criteria[0]), int(criteria[1]), float(criteria[2])
        p.min_eig_thr, p.fb_thresh = float(minEigThreshold), float(fb_thresh)
        p.max_corners, p.quality_level, p.min_distance = int(max_corners), float(quality_level), float(min_distance)
        p.thr_new_keypoints, p.thr_triangulation = float(threshold_new_keypoints), float(threshold_triangulation)
        p.thr_candidates_min, p.thr_candidates_max = float(threshold_candidates_min), float(threshold_candidates_max)
        p.thr_alpha_rad = float(np.deg2rad(threshold_alpha_deg))
        p.cap_landmarks, p.cap_candidates = int(cap_landmarks), int(cap_candidates)
        p.first_is_f64 = int(self.fdt == np.float64)
        p.K[:] = [float(v) for v in np.asarray(K, dtype=np.float64).reshape(9)]
        self._h = C.c_void_p()
        _lib.check(self.L.vofe_framestate_create(int(device), C.byref(p), C.byref(self._h)))
        buf = C.c_void_p()
        _lib.check(self.L.vofe_framestate_frame_buffer(self._h, C.byref(buf)))
        # the pinned staging frame: a caller (an image decoder) may write the next frame here and hand this array to track()
        self.frame_buffer = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(self.h, self.w))
        self._n_tracked = 0

    # ------------------------------------------------------------------------------------------------ state in / out
    @property
    def K(self):
        k = C.c_int()
        _lib.check(self.L.vofe_framestate_counts(self._h, C.byref(k), None))
        return k.value

    @property
    def M(self):
        m = C.c_int()
        _lib.check(self.L.vofe_framestate_counts(self._h, None, C.byref(m)))
        return m.value

    @property
    def landmarks_world(self):
        """Xi, (3,K) float64 (a copy)."""
        ptr, k = C.c_void_p(), C.c_int()
        _lib.check(self.L.vofe_framestate_landmarks_world(self._h, C.byref(ptr), C.byref(k)))
        if k.value == 0:
            return np.empty((3, 0))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(k.value, 3)).T.copy()

    def _frame(self, img):
        a = np.asarray(img)
        if a.dtype != np.uint8 or a.shape != (self.h, self.w):
            raise error(-1, "need a uint8 frame of %dx%d" % (self.w, self.h))
        if a.strides[1] != 1 or a.strides[0] < self.w:
            a = np.ascontiguousarray(a)
        return a

    def reset(self, img, landmarks_image, landmarks_world, cand_current=None, cand_first=None, cand_transform=None):
        """Start from a FrameState (reference main.py:82-90 builds the first one) on its frame `img`."""
        a = self._frame(img)
        P = np.ascontiguousarray(np.asarray(landmarks_image, dtype=np.float32).T.reshape(-1, 2))
        X = np.ascontiguousarray(np.asarray(landmarks_world, dtype=np.float64).reshape(3, -1).T)
        Cc = np.zeros((0, 2), np.float32) if cand_current is None else np.ascontiguousarray(np.asarray(cand_current, dtype=np.float32).T.reshape(-1, 2))
        Fc = np.zeros((0, 2), self.fdt) if cand_first is None else np.ascontiguousarray(np.asarray(cand_first, dtype=self.fdt).T.reshape(-1, 2))
        Tc = np.zeros((0, 12)) if cand_transform is None else np.ascontiguousarray(np.asarray(cand_transform, dtype=np.float64)[:12].T)
        if X.shape[0] != P.shape[0] or Fc.shape[0] != Cc.shape[0] or Tc.shape[0] != Cc.shape[0]:
            raise error(-1, "DeviceFrameState.reset: inconsistent state")
        _lib.check(self.L.vofe_framestate_reset(self._h, a.ctypes.data, int(a.strides[0]), P.ctypes.data, X.ctypes.data, P.shape[0],
                                                Cc.ctypes.data, Fc.ctypes.data, Tc.ctypes.data, Cc.shape[0]))

    def download(self):
        """The FrameState arrays as the reference holds them: (Pi (2,K) f32, Xi (3,K) f64, Ci (2,M) f32, Fi (2,M), Ti (12,M) f64)."""
        k, m = self.K, self.M                     # raises between track() and update()
        X = self.landmarks_world
        P, Cc = np.empty((k, 2), np.float32), np.empty((m, 2), np.float32)
        Fc, Tc = np.empty((m, 2), self.fdt), np.empty((m, 12), np.float64)
        _lib.check(self.L.vofe_framestate_download(self._h, P.ctypes.data, Cc.ctypes.data, Fc.ctypes.data, Tc.ctypes.data))
        return P.T, X, Cc.T, Fc.T, Tc.T

    # ------------------------------------------------------------------------------------------------ phase 1
    def track(self, img_current):
        """Reference continuous.py:86-87 (landmarks) and :380-405 (detection, candidate tracking, status compaction,
        travel-distance band), all asynchronous, one wait.  Returns (tracked (2,K) f32, status (K,) u8) of the landmarks."""
        a = img_current if img_current is self.frame_buffer else self._frame(img_current)
        xy, st, k = C.c_void_p(), C.c_void_p(), C.c_int()
        _lib.check(self.L.vofe_framestate_track(self._h, a.ctypes.data, int(a.strides[0]), C.byref(xy), C.byref(st), C.byref(k),
                                                None, None))
        n = self._n_tracked = k.value
        if n == 0:
            return np.empty((2, 0), np.float32), np.empty((0,), np.uint8)
        tracked = np.ctypeslib.as_array(C.cast(xy, C.POINTER(C.c_float)), shape=(n, 2)).copy()
        status = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint8)), shape=(n,)).copy()
        return tracked.T, status

    # ------------------------------------------------------------------------------------------------ phase 2
    def update(self, pose, keep, wait=True):
        """Reference continuous.py:90-92 + the inlier selection of :95-101 (keep = status == 1 and kept by the pose
        estimation, a (K,) mask) and :408-495: de-duplication, triangulation, promotion, new candidates.
        Returns the number of promoted candidates; with wait=False the work is only queued (returns None) and is
        collected by wait() or by whichever call needs its outcome next -- track() overlaps it with the copy, upload
        and pyramid of its frame."""
        keep = np.ascontiguousarray(np.asarray(keep).reshape(-1) != 0).view(np.uint8)
        if keep.shape[0] != self._n_tracked:
            raise error(-1, "keep mask must have one entry per landmark")
        pose = np.ascontiguousarray(np.asarray(pose, dtype=np.float64)[:3, :4])
        rc = self.L.vofe_framestate_update_async(self._h, pose.ctypes.data, keep.ctypes.data)
        if rc:
            msg = self.L.vofe_last_error().decode("utf-8", "replace")
            if msg.startswith("zero-size array"):
                raise ValueError(msg)                  # what np.min(cdist(...), axis=1) raises in the reference
            raise error(rc, msg)
        return self.wait() if wait else None

    def wait(self):
        """Collect a queued update; returns the number of candidates it promoted."""
        n_prom = C.c_int()
        _lib.check(self.L.vofe_framestate_wait(self._h, C.byref(n_prom)))
        return n_prom.value

    def close(self):
        if self._h:
            self.L.vofe_framestate_destroy(self._h)
            self._h = C.c_void_p()
            self.frame_buffer = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
