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

Host side only: built from the C-ABI entry points vofe_pyramid_build / vofe_track_fb / vofe_candidate_filter /
vofe_gftt / vofe_new_feature_mask2 / vofe_triangulate_candidates / vofe_candidate_commit.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import error
from .frontend import FrontEnd, _DEFAULT_CRITERIA
from .sequence import SequenceTracker


def _mapped(nbytes):
    """Pinned host memory the device addresses directly: (tensor, device pointer)."""
    t = torch.zeros(int(nbytes), dtype=torch.uint8).pin_memory()
    dev = C.c_void_p()
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
    if hip.hipHostGetDevicePointer(C.byref(dev), C.c_void_p(t.data_ptr()), 0) != 0 or not dev.value:
        raise RuntimeError("hipHostGetDevicePointer failed for the pinned result buffer")
    return t, dev.value


class DeviceFrameState:
    def __init__(self, height, width, K, winSize=(21, 21), maxLevel=3, criteria=_DEFAULT_CRITERIA, minEigThreshold=1e-3,
                 fb_thresh=1.0, max_corners=1000, quality_level=0.01, min_distance=10, threshold_new_keypoints=15,
                 threshold_triangulation=15, threshold_candidates_min=2, threshold_candidates_max=100, threshold_alpha_deg=1,
                 cap_landmarks=8192, cap_candidates=8192, first_dtype=np.float64, frontend=None, device=0):
        self.fe = frontend if frontend is not None else FrontEnd(device)
        self._own_fe = frontend is None
        self.h, self.w = int(height), int(width)
        self.trk = SequenceTracker(self.h, self.w, winSize, maxLevel, criteria, minEigThreshold, fb_thresh, frontend=self.fe)
        self.det = (int(max_corners), float(quality_level), float(min_distance))
        self.thr = dict(new=float(threshold_new_keypoints), tri=float(threshold_triangulation), cmin=float(threshold_candidates_min),
                        cmax=float(threshold_candidates_max), alpha=float(np.deg2rad(threshold_alpha_deg)))
        self.capP, self.capC = int(cap_landmarks), int(cap_candidates)
        self.fdt = np.dtype(first_dtype)
        if self.fdt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise error(-1, "first observations must be float32 or float64")
        dev = self.fe.device
        ncap = self.capP + self.capC
        self.d_track = torch.zeros((ncap, 2), dtype=torch.float32, device=dev)         # [landmarks | candidates] of the last frame
        self.d_C = torch.zeros((self.capC, 2), dtype=torch.float32, device=dev)
        self.d_F = torch.zeros((self.capC, 2), dtype=torch.float64 if self.fdt == np.float64 else torch.float32, device=dev)
        self.d_T = torch.zeros((self.capC, 12), dtype=torch.float64, device=dev)
        self.d_tri = torch.zeros(self.capC, dtype=torch.uint8, device=dev)
        self.d_prom = torch.zeros(self.capC, dtype=torch.uint8, device=dev)
        self.d_xyz = torch.zeros((self.capC, 3), dtype=torch.float64, device=dev)
        self.d_feat = torch.zeros((max(self.det[0], 1), 2), dtype=torch.float32, device=dev)
        self.d_fmask = torch.zeros(max(self.det[0], 1), dtype=torch.uint8, device=dev)
        # phase-2 parameters, one H2D copy: pose (12 f64) | K (9 f64) | pad | surviving landmarks (capP, 2) f32
        self._par_host = torch.zeros(24 * 8 + self.capP * 8, dtype=torch.uint8).pin_memory()
        self.d_par = torch.zeros(24 * 8 + self.capP * 8, dtype=torch.uint8, device=dev)
        self.d_P = self.d_par[24 * 8:].view(torch.float32).view(self.capP, 2)
        self._par_host.numpy()[96:168].view(np.float64)[:] = np.asarray(K, dtype=np.float64).reshape(9)
        # results the kernels write straight into pinned host memory
        self._res, self._res_dev = _mapped(ncap * 13 + 128 + self.capC * (8 + 24))
        self._o_cnt = (ncap * 13 + 63) // 64 * 64    # int32 x 8: [kept candidates, ., ., ., promoted, candidates, landmarks, overflow]
        self._o_newP = self._o_cnt + 64
        self._o_newX = self._o_newP + self.capC * 8
        self.K = self.M = 0
        self.landmarks_world = np.empty((3, 0))
        self._phase = 0
        self._det_fe, self._det_stream = None, None

    # ------------------------------------------------------------------------------------------------ state in / out
    def reset(self, img, landmarks_image, landmarks_world, cand_current=None, cand_first=None, cand_transform=None):
        """Start from a FrameState (reference main.py:82-90 builds the first one) on its frame `img`."""
        P = np.ascontiguousarray(np.asarray(landmarks_image, dtype=np.float32).T.reshape(-1, 2))
        self.landmarks_world = np.array(landmarks_world, dtype=np.float64).reshape(3, -1)
        Cc = np.zeros((0, 2), np.float32) if cand_current is None else np.ascontiguousarray(np.asarray(cand_current, dtype=np.float32).T.reshape(-1, 2))
        Fc = np.zeros((0, 2), self.fdt) if cand_first is None else np.ascontiguousarray(np.asarray(cand_first, dtype=self.fdt).T.reshape(-1, 2))
        Tc = np.zeros((0, 12)) if cand_transform is None else np.ascontiguousarray(np.asarray(cand_transform, dtype=np.float64)[:12].T)
        self.K, self.M = P.shape[0], Cc.shape[0]
        if self.K > self.capP or self.M > self.capC or self.landmarks_world.shape[1] != self.K or Fc.shape[0] != self.M or Tc.shape[0] != self.M:
            raise error(-1, "DeviceFrameState.reset: inconsistent or over-capacity state")
        self.trk.reset(img)
        dev = self.fe.device
        if self.K:
            self.d_track[: self.K].copy_(torch.from_numpy(P))
        if self.M:
            self.d_track[self.K: self.K + self.M].copy_(torch.from_numpy(Cc))
            self.d_C[: self.M].copy_(torch.from_numpy(Cc))
            self.d_F[: self.M].copy_(torch.from_numpy(Fc))
            self.d_T[: self.M].copy_(torch.from_numpy(Tc))
        torch.cuda.current_stream(dev).synchronize()
        self._phase = 0

    def download(self):
        """The FrameState arrays as the reference holds them: (Pi (2,K) f32, Xi (3,K) f64, Ci (2,M) f32, Fi (2,M), Ti (12,M) f64)."""
        if self._phase != 0:
            raise error(-1, "DeviceFrameState.download between track() and update()")
        return (self.d_track[: self.K].cpu().numpy().T, self.landmarks_world, self.d_C[: self.M].cpu().numpy().T,
                self.d_F[: self.M].cpu().numpy().T, self.d_T[: self.M].cpu().numpy().T)

    # ------------------------------------------------------------------------------------------------ phase 1
    def track(self, img_current):
        """Reference continuous.py:86-87 (landmarks) and :380-405 (detection, candidate tracking, status compaction,
        travel-distance band), all asynchronous, one wait.  Returns (tracked (2,K) f32, status (K,) u8) of the landmarks."""
        if self._phase != 0:
            raise error(-1, "DeviceFrameState.track called twice without update()")
        fe, L, trk = self.fe, self.fe.L, self.trk
        if self._det_fe is None:
            self._det_stream = torch.cuda.Stream(device=fe.device, priority=-1)   # the detector is a chain of short launches: let it go first
            self._det_fe = FrontEnd(fe.device.index, stream=self._det_stream)
        prev_slot = trk.frame_index & 1
        next_slot = trk._push(img_current)                     # upload + ONE pyramid build
        K, M = self.K, self.M
        n = K + M
        b = self._res_dev
        o_next, o_err, o_st = b, b + n * 8, b + n * 12
        if n:
            _lib.check(L.vofe_track_fb(fe.ctx, trk.pyr._h, prev_slot, trk.pyr._h, next_slot, self.d_track.data_ptr(), None, 1,
                                       o_next, o_st, o_err, n, trk.criteria[0], trk.criteria[1], trk.criteria[2], trk.min_eig,
                                       trk.fb_thresh))
        # candidates: status == 1, travel-distance band, triangulation flag (continuous.py:391-405, 446-448); the kernel
        # reads the tracker's outputs from the pinned buffer and compacts into the device arrays
        _lib.check(L.vofe_candidate_filter(fe.ctx, o_next + K * 8, o_st + K, self.d_F.data_ptr(), int(self.fdt == np.float64),
                                           self.d_T.data_ptr(), 12, M, self.thr["cmin"], self.thr["cmax"], self.thr["tri"],
                                           self.d_C.data_ptr(), self.d_F.data_ptr(), self.d_T.data_ptr(), self.d_tri.data_ptr(),
                                           b + self._o_cnt, None))
        # detection on the second stream while the tracker works (continuous.py:380-385)
        img = trk._img[trk.frame_index & 1]
        img.record_stream(self._det_stream)
        self._det_stream.wait_event(trk._uploaded)
        nf = C.c_int(0)
        dfe = self._det_fe
        with torch.cuda.stream(self._det_stream):
            _lib.check(dfe.L.vofe_gftt(dfe.ctx, img.data_ptr(), self.h, self.w, img.stride(0), self.det[0], self.det[1], self.det[2],
                                       None, 3, 3, 0, 0.04, self.d_feat.data_ptr(), self.d_feat.shape[0], C.byref(nf)))
        fe.synchronize()
        self._nf = nf.value
        host = self._res.numpy()
        self._m1 = int(host[self._o_cnt: self._o_cnt + 4].view(np.int32)[0]) if M else 0
        tracked = host[: K * 8].view(np.float32).reshape(K, 2).copy()
        status = host[n * 12: n * 12 + K].copy()
        self._tracked = tracked
        self._phase = 1
        return tracked.T, status

    # ------------------------------------------------------------------------------------------------ phase 2
    def update(self, pose, keep):
        """Reference continuous.py:90-92 + the inlier selection of :95-101 (keep = status == 1 and kept by the pose
        estimation, a (K,) mask) and :408-495: de-duplication, triangulation, promotion, new candidates.
        Returns the number of promoted candidates."""
        if self._phase != 1:
            raise error(-1, "DeviceFrameState.update before track()")
        fe, L = self.fe, self.fe.L
        keep = np.asarray(keep).astype(bool).reshape(-1)
        if keep.shape[0] != self.K:
            raise error(-1, "keep mask must have one entry per landmark")
        P1 = self._tracked[keep]
        K1, m1, nf = P1.shape[0], self._m1, self._nf
        self.landmarks_world = self.landmarks_world[:, keep]
        if nf and m1 + K1 == 0:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")   # np.min(cdist(...), axis=1)
        ph = self._par_host.numpy()
        ph[:96].view(np.float64)[:] = np.asarray(pose, dtype=np.float64)[:3, :4].reshape(12)
        ph[192: 192 + K1 * 8].view(np.float32)[:] = P1.reshape(-1)
        nbytes = 192 + K1 * 8
        self.d_par[:nbytes].copy_(self._par_host[:nbytes], non_blocking=True)
        d_pose, d_K = self.d_par.data_ptr(), self.d_par.data_ptr() + 96
        if nf:
            _lib.check(L.vofe_new_feature_mask2(fe.ctx, self.d_feat.data_ptr(), nf, self.d_C.data_ptr(), m1, self.d_P.data_ptr(), K1,
                                                self.thr["new"], self.d_fmask.data_ptr()))
        if m1:
            _lib.check(L.vofe_triangulate_candidates(fe.ctx, d_K, d_pose, self.d_C.data_ptr(), self.d_F.data_ptr(),
                                                     int(self.fdt == np.float64), self.d_T.data_ptr(), 12, self.d_tri.data_ptr(), m1,
                                                     self.thr["alpha"], self.d_prom.data_ptr(), self.d_xyz.data_ptr()))
        b = self._res_dev
        _lib.check(L.vofe_candidate_commit(fe.ctx, self.d_C.data_ptr(), self.d_F.data_ptr(), int(self.fdt == np.float64),
                                           self.d_T.data_ptr(), 12, m1, self.capC, self.d_prom.data_ptr(), self.d_xyz.data_ptr(),
                                           self.d_feat.data_ptr(), self.d_fmask.data_ptr(), nf, d_pose, self.d_P.data_ptr(), K1,
                                           self.capP, self.d_track.data_ptr(), b + self._o_newP, b + self._o_newX,
                                           b + self._o_cnt + 16))
        fe.synchronize()
        host = self._res.numpy()
        n_prom, m_out, k_out, overflow = (int(x) for x in host[self._o_cnt + 16: self._o_cnt + 32].view(np.int32))
        if overflow:
            raise error(-4, "DeviceFrameState: landmark / candidate capacity exceeded (%d / %d)" % (self.capP, self.capC))
        if n_prom:
            newX = host[self._o_newX: self._o_newX + n_prom * 24].view(np.float64).reshape(n_prom, 3)
            self.landmarks_world = np.hstack([self.landmarks_world, newX.T])
        self.K, self.M = k_out, m_out
        self._phase = 0
        return n_prom

    def close(self):
        if self.trk is not None:
            self.trk.close()
            self.trk = None
        if self._det_fe is not None:
            self._det_fe.close()
            self._det_fe = None
        if self._own_fe and self.fe is not None:
            self.fe.close()
        self.fe = None
