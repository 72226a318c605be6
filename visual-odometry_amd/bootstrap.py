"""The front-end half of the reference's bootstrap, Initialization.get_features_and_k (initialization.py:39-70): corners
of the first image, then per further image ONE forward single-level LK call (no backward check), `status == 1`, the
"moved far enough from its position in the first image" rule, and the inlier mask of the essential-matrix RANSAC.  The
point pairs (good_old in image 0, good_new in the latest image) stay on the device between the steps; the RANSAC itself
(cv2.findEssentialMat, initialization.py:67-68) and everything after the loop stay on the reference's CPU path: the caller
runs them on the arrays `track` returns and hands the mask back through `keep`."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import error
from .frontend import FrontEnd, Pyramid, default_frontend

_CRITERIA = (3, 30, 0.01)                      # initialization.py:49-50


class BootstrapTracker:
    def __init__(self, height, width, maxCorners=1000, qualityLevel=0.01, minDistance=10, winSize=(11, 11),
                 dist_threshold_move=0, maxLevel=0, criteria=_CRITERIA, minEigThreshold=1e-4, frontend=None, **_ignored):
        self.fe = frontend if frontend is not None else default_frontend()
        if not isinstance(self.fe, FrontEnd):
            raise error(-1, "frontend must be a FrontEnd")
        self.h, self.w = int(height), int(width)
        self.max_corners, self.quality, self.min_distance = int(maxCorners), float(qualityLevel), float(minDistance)
        self.win, self.max_level, self.criteria = (int(winSize[0]), int(winSize[1])), int(maxLevel), tuple(criteria)
        self.min_eig, self.dist_move = float(minEigThreshold), float(dist_threshold_move)
        if self.max_corners <= 0:
            raise error(-1, "BootstrapTracker needs maxCorners > 0 (the capacity of its point arrays)")
        dev = self.fe.device
        cap = self.max_corners
        self.pyr = Pyramid(self.fe, 2, self.h, self.w, self.win, self.max_level, True)
        self._frames = torch.empty((2, self.h, self.w), dtype=torch.uint8, device=dev)
        self._old = torch.empty((cap, 2), dtype=torch.float32, device=dev)      # positions in image 0
        self._new = torch.empty((cap, 2), dtype=torch.float32, device=dev)      # positions in the latest image
        self._next = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        self._status = torch.empty(cap, dtype=torch.uint8, device=dev)
        self._err = torch.empty(cap, dtype=torch.float32, device=dev)
        self._cnt = torch.empty(1, dtype=torch.int32, device=dev)
        self.n = 0
        self._slot = -1

    def _push(self, img):
        slot = (self._slot + 1) & 1
        dst = self._frames[slot:slot + 1]
        if isinstance(img, torch.Tensor):
            d = self.fe.to_device(img)
            if tuple(d.shape) != (self.h, self.w):
                raise error(-1, "image size differs from the tracker's")
            with torch.cuda.stream(self.fe.stream):
                dst[0].copy_(d)
        else:
            a = np.asarray(img)
            if a.shape != (self.h, self.w):
                raise error(-1, "image size differs from the tracker's")
            self.fe.upload_frames(a, out=dst, tag="boot")
        self.pyr.build(dst, slot)
        self._slot = slot

    def _pairs(self):
        n = self.n
        with torch.cuda.stream(self.fe.stream):
            both = torch.stack([self._old[:n], self._new[:n]]).cpu().numpy()   # one download
        self.fe._stage_busy = False                                            # (that download waited for the stream)
        return both[0].reshape(-1, 1, 2), both[1].reshape(-1, 1, 2)

    def detect(self, img0):
        """initialization.py:41-44: corners of the first image; good_old = good_new = corners.  Returns the count."""
        self._slot = -1
        self._push(img0)
        n = C.c_int(0)
        L = self.fe.L
        _lib.check(L.vofe_gftt(self.fe.ctx, self._frames[self._slot].data_ptr(), self.h, self.w, self.w, self.max_corners,
                               self.quality, self.min_distance, None, 3, 3, 0, 0.04, self._old.data_ptr(), self.max_corners,
                               C.byref(n)))
        if n.value == 0:
            raise error(-1, "no corner in the first image")       # the reference fails here as well (np.float32(None).reshape)
        self.n = n.value
        self.fe._stage_busy = False                   # vofe_gftt returned after a wait for the stream
        with torch.cuda.stream(self.fe.stream):
            self._new[:self.n].copy_(self._old[:self.n])
        return self.n

    def track(self, img_next):
        """initialization.py:51-64 for the next image: forward LK from the latest positions, status == 1, drop the pairs
        that moved less than dist_threshold_move since image 0.  Returns (good_old, good_new), (N,1,2) float32 each."""
        if self._slot < 0:
            raise error(-1, "BootstrapTracker.track before detect")
        prev = self._slot
        self._push(img_next)
        n = self.n
        if n == 0:
            # cv2 raises on an empty prevPts here (initialization.py:51): so does the twin
            raise error(-1, "calcOpticalFlowPyrLK: prevPts is empty")
        L, ctx = self.fe.L, self.fe.ctx
        _lib.check(L.vofe_lk_track(ctx, self.pyr._h, prev, self.pyr._h, self._slot, self._new.data_ptr(), None, 1,
                                   self._next.data_ptr(), self._status.data_ptr(), self._err.data_ptr(), n,
                                   int(self.criteria[0]), int(self.criteria[1]), float(self.criteria[2]), 0, self.min_eig))
        m = C.c_int(0)
        _lib.check(L.vofe_pair_filter(ctx, self._next.data_ptr(), self._status.data_ptr(), self._old.data_ptr(), n,
                                      self.dist_move, self._new.data_ptr(), self._old.data_ptr(), self._cnt.data_ptr(),
                                      C.byref(m)))
        self.n = m.value
        return self._pairs()

    def keep(self, mask):
        """initialization.py:69-70: good_old[mask.flatten() == 1], good_new[mask.flatten() == 1] (mask: the inliers of the
        caller's essential-matrix RANSAC over the pairs `track` returned).  Returns the new count."""
        mk = np.ascontiguousarray(np.asarray(mask).reshape(-1) == 1).astype(np.uint8)
        if mk.shape[0] != self.n:
            raise error(-1, "mask length %d != number of pairs %d" % (mk.shape[0], self.n))
        if self.n == 0:
            return 0
        with torch.cuda.stream(self.fe.stream):
            d_mask = self._status[:self.n].copy_(torch.from_numpy(mk))
        m = C.c_int(0)
        _lib.check(self.fe.L.vofe_pair_filter(self.fe.ctx, self._new.data_ptr(), d_mask.data_ptr(), self._old.data_ptr(), self.n,
                                              0.0, self._new.data_ptr(), self._old.data_ptr(), self._cnt.data_ptr(), C.byref(m)))
        self.n = m.value
        return self.n

    def pairs(self):
        return self._pairs()


def bootstrap_correspondences(imgs, params, essential_mask, frontend=None):
    """The loop of initialization.py:41-70.  `essential_mask(good_new, good_old) -> mask (N,) or (N,1)` is the caller's
    cv2.findEssentialMat(...)[1] (CPU, out of scope here).  Returns (good_old, good_new) after the last image."""
    imgs = list(imgs)
    h, w = (imgs[0].shape[-2], imgs[0].shape[-1])
    bt = BootstrapTracker(h, w, frontend=frontend, **params)
    bt.detect(imgs[0])
    for img in imgs[1:]:
        good_old, good_new = bt.track(img)
        bt.keep(essential_mask(good_new, good_old))
    return bt.pairs()
