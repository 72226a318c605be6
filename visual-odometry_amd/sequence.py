"""Frame-level tracker with a cached pyramid (first "next" row).

The reference calls cv2.calcOpticalFlowPyrLK four times per frame (continuous.py:145-151 inside
_track_keypoints, called for landmarks at continuous.py:180 and for candidates at continuous.py:330),
and every call rebuilds the pyramids of both images.  Frames arrive in order, so the pyramid of the
previous frame is still valid: SequenceTracker keeps a two-slot pyramid ring, builds ONE pyramid per
new frame and tracks all point sets of the frame (landmarks and candidates) forward + backward with
the forward-backward check in ONE launch.  Results per set are the same triple, bit for bit, that
FrontEnd.track_keypoints (= ContinuousVO._track_keypoints) returns.

Host side only: it is built from the C-ABI entry points vofe_pyramid_build / vofe_track_fb / vofe_gftt.
"""
import numpy as np
import torch

from . import _lib
from ._lib import error
from .frontend import FrontEnd, Pyramid, _DEFAULT_CRITERIA


class SequenceTracker:
    def __init__(self, height, width, winSize=(21, 21), maxLevel=3, criteria=_DEFAULT_CRITERIA,
                 minEigThreshold=1e-3, fb_thresh=1.0, frontend=None, device=0):
        self.fe = frontend if frontend is not None else FrontEnd(device)
        self._own_fe = frontend is None
        self.h, self.w = int(height), int(width)
        self.criteria = (int(criteria[0]), int(criteria[1]), float(criteria[2]))
        self.min_eig = float(minEigThreshold)
        self.fb_thresh = float(fb_thresh)
        self.pyr = Pyramid(self.fe, 2, self.h, self.w, winSize, maxLevel, True)
        self.frame_index = -1                     # index of the frame held in slot frame_index & 1
        self._img = [None, None]                  # device images of the two slots (for detect())
        self._stage, self._stage_k, self._copy_stream, self._prefetched = None, 0, None, None
        self._det_fe, self._det_stream = None, None            # side context for track_and_detect()
        self._ring = None                          # device copies of the two newest host frames

    # -- frames ----------------------------------------------------------------------------------
    def prefetch(self, img_next):
        """Image ingest overlap: start the upload of the NEXT frame (pinned staging buffer,
        asynchronous copy on a side stream) while the caller still works on the current one -- e.g. right after
        track() returns, before the CPU-side PnP/triangulation of reference continuous.py.  The following
        track(img_next, ...) / reset(img_next) call picks the device copy up if it is handed the same array."""
        a = np.asarray(img_next)
        if a.dtype != np.uint8 or a.shape != (self.h, self.w):
            raise error(-1, "prefetch: need a uint8 frame of %dx%d" % (self.w, self.h))
        if self._stage is None:
            self._stage = [torch.empty((self.h, self.w), dtype=torch.uint8).pin_memory() for _ in range(2)]
            self._stage_ev = [None, None]
            self._copy_stream = torch.cuda.Stream(device=self.fe.device)
        k = self._stage_k = 1 - self._stage_k
        if self._stage_ev[k] is not None:
            self._stage_ev[k].synchronize()                   # the previous upload out of this staging buffer has left it
        self._stage[k].numpy()[...] = a                       # host copy into pinned memory
        with torch.cuda.stream(self._copy_stream):
            d = self._stage[k].to(self.fe.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._stage_ev[k] = ev
        self._prefetched = (a, d, ev)

    def _push(self, img):
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is img:
            torch.cuda.current_stream(self.fe.device).wait_event(pre[2])
            d = pre[1]
            # the frame was allocated on the copy stream: tell the caching allocator which streams consume it, so the
            # block is not handed to a later side-stream copy while their kernels still read it
            d.record_stream(torch.cuda.current_stream(self.fe.device))
            if self._det_stream is not None:
                d.record_stream(self._det_stream)
            self._uploaded = pre[2]
        elif isinstance(img, torch.Tensor):
            d = self.fe.to_device(img)
            self._uploaded = torch.cuda.Event()                # the frame is resident once this has passed
            self._uploaded.record(self.fe.stream)
        else:
            # host frame: through the front end's pinned staging buffer into this tracker's own two-frame ring (the
            # frame of either slot stays valid for detect() until the slot is reused)
            if self._ring is None:
                self._ring = torch.empty((2, 1, self.h, self.w), dtype=torch.uint8, device=self.fe.device)
            a = np.asarray(img)
            if a.shape != (self.h, self.w):
                raise error(-1, "frame size %s does not match the tracker geometry %s" % (tuple(a.shape), (self.h, self.w)))
            d = self.fe.upload_frames(a, out=self._ring[(self.frame_index + 1) & 1])[0]
            self._uploaded = torch.cuda.Event()
            self._uploaded.record(self.fe.stream)
        if tuple(d.shape) != (self.h, self.w):
            raise error(-1, "frame size %s does not match the tracker geometry %s" % (tuple(d.shape), (self.h, self.w)))
        self.frame_index += 1
        slot = self.frame_index & 1
        self.pyr.build(d.reshape(1, self.h, self.w), slot)
        self._img[slot] = d
        return slot

    def reset(self, img):
        """Start a sequence with its first frame (no tracking)."""
        self.frame_index = -1
        self._push(img)

    def track(self, img_current, *point_sets):
        """Track every (2,N) point set of the previous frame into img_current.

        Returns one (tracked (2,N) f32, status (N,) u8, err (N,) f32) triple per set, identical to
        track_keypoints(points, img_prev, img_current); empty sets give the empty triple of
        continuous.py:131-132.  img_current becomes the previous frame of the next call."""
        return self._collect(self._launch(img_current, point_sets))

    def _launch(self, img_current, point_sets):
        if self.frame_index < 0:
            raise error(-1, "SequenceTracker.track before reset()")
        prev_slot = self.frame_index & 1
        next_slot = self._push(img_current)
        sets = [np.asarray(p) for p in point_sets]
        flat = [np.ascontiguousarray(p.T.reshape(-1, 2).astype(np.float32)) for p in sets if p.size]
        n = sum(f.shape[0] for f in flat)
        io = None
        if n:
            fe = self.fe
            io = fe._points_io(np.concatenate(flat, 0))
            _lib.check(fe.L.vofe_track_fb(fe.ctx, self.pyr._h, prev_slot, self.pyr._h, next_slot, io["prev"], None, 1, io["next"],
                                          io["status"], io["err"], n, self.criteria[0], self.criteria[1], self.criteria[2],
                                          self.min_eig, self.fb_thresh))
        return sets, io, n

    def _collect(self, launched):
        sets, io, n = launched
        if n:
            nxt, st, err = self.fe._points_done(io)
        out, o = [], 0
        for p in sets:
            if p.size == 0:
                out.append((np.empty((2, 0)), np.empty((0,), dtype=np.uint8), np.empty((0,), dtype=np.float32)))
                continue
            k = p.T.reshape(-1, 2).shape[0]
            out.append((nxt[o:o + k].T, st[o:o + k], err[o:o + k]))
            o += k
        return out

    def track_and_detect(self, img_current, point_sets, maxCorners, qualityLevel, minDistance, mask=None,
                         useHarrisDetector=False, k=0.04):
        """One frame of the reference's front end with its two independent halves overlapped on the GPU: the tracking of
        every point set into img_current (continuous.py:86-87, 389-390) runs on the tracker's stream while
        goodFeaturesToTrack(img_current) (continuous.py:380-385) runs on a second context and stream.  Neither reads what
        the other writes, so the results are those of track() followed by detect(), bit for bit; the frame costs about
        the longer of the two instead of their sum.  Returns (list of triples, corners)."""
        if self._det_fe is None:
            self._det_stream = torch.cuda.Stream(device=self.fe.device)
            self._det_fe = FrontEnd(self.fe.device.index, stream=self._det_stream)
        launched = self._launch(img_current, point_sets)       # upload, pyramid, LK launch: asynchronous
        img = self._img[self.frame_index & 1]
        img.record_stream(self._det_stream)                   # consumed on the detector's stream as well
        self._det_stream.wait_event(self._uploaded)            # only the upload: not the pyramid or the tracker launch
        with torch.cuda.stream(self._det_stream):
            corners = self._det_fe.goodFeaturesToTrack(img, maxCorners, qualityLevel, minDistance, mask=mask,
                                                       useHarrisDetector=useHarrisDetector, k=k)
        return self._collect(launched), corners

    def detect(self, maxCorners, qualityLevel, minDistance, mask=None, useHarrisDetector=False, k=0.04):
        """cv2.goodFeaturesToTrack on the current frame (continuous.py:380-385) without another upload."""
        if self.frame_index < 0:
            raise error(-1, "SequenceTracker.detect before reset()")
        return self.fe.goodFeaturesToTrack(self._img[self.frame_index & 1], maxCorners, qualityLevel, minDistance,
                                           mask=mask, useHarrisDetector=useHarrisDetector, k=k)

    def close(self):
        if self.pyr is not None:
            self.pyr.free()
            self.pyr = None
        if self._det_fe is not None:
            self._det_fe.close()
            self._det_fe = None
        if self._own_fe and self.fe is not None:
            self.fe.close()
        self.fe = None


class MultiSequenceTracker:
    """B independent sequences of one geometry on one GPU (BASELINE configs[3]; SURVEY 8e: sequences are the only
    unit the path shards over).  Per frame index: ONE pyramid build of the B new frames and ONE tracker launch over the
    points of all sequences (a per-point image index selects the sequence), instead of B builds and B launches; the
    detector runs batched too.  Results per sequence are those of B SequenceTrackers, bit for bit."""

    def __init__(self, n_sequences, height, width, winSize=(21, 21), maxLevel=3, criteria=_DEFAULT_CRITERIA,
                 minEigThreshold=1e-3, fb_thresh=1.0, frontend=None, device=0):
        self.fe = frontend if frontend is not None else FrontEnd(device)
        self._own_fe = frontend is None
        self.B, self.h, self.w = int(n_sequences), int(height), int(width)
        self.criteria = (int(criteria[0]), int(criteria[1]), float(criteria[2]))
        self.min_eig = float(minEigThreshold)
        self.fb_thresh = float(fb_thresh)
        self.pyr = Pyramid(self.fe, 2 * self.B, self.h, self.w, winSize, maxLevel, True)   # two rings of B slots
        self.frame_index = -1
        self._img = [None, None]
        self._idx_cache = (None, None)            # (counts, device int32 image index per point) of the last call

    def _push(self, frames):
        if isinstance(frames, torch.Tensor):
            d = frames.to(self.fe.device)
        else:
            a = np.ascontiguousarray(np.asarray(frames))
            if a.dtype != np.uint8:
                raise error(-1, "frames must be 8-bit")
            d = torch.from_numpy(a).to(self.fe.device)
        if tuple(d.shape) != (self.B, self.h, self.w) or d.dtype != torch.uint8:
            raise error(-1, "need uint8 frames of shape (%d, %d, %d)" % (self.B, self.h, self.w))
        self.frame_index += 1
        ring = self.frame_index & 1
        self.pyr.build(d, ring * self.B)
        self._img[ring] = d
        return ring

    def reset(self, frames):
        """First frame of every sequence: (B,H,W) uint8 (numpy or CUDA tensor)."""
        self.frame_index = -1
        self._push(frames)

    def track(self, frames, point_sets):
        """frames: the next frame of every sequence, (B,H,W); point_sets: B arrays (2,N_b) (N_b may be 0).
        Returns B (tracked (2,N_b) f32, status (N_b,) u8, err (N_b,) f32) triples."""
        if self.frame_index < 0:
            raise error(-1, "MultiSequenceTracker.track before reset()")
        if len(point_sets) != self.B:
            raise error(-1, "need one point set per sequence")
        prev_ring = self.frame_index & 1
        next_ring = self._push(frames)
        sets = [np.asarray(p) for p in point_sets]
        flat = [np.ascontiguousarray(p.T.reshape(-1, 2).astype(np.float32)) if p.size else np.empty((0, 2), np.float32)
                for p in sets]
        counts = [f.shape[0] for f in flat]
        n = sum(counts)
        if n:
            fe = self.fe
            io = fe._points_io(np.concatenate(flat, 0))
            if self._idx_cache[0] != counts:          # the per-point image index changes only when the counts do
                self._idx_cache = (counts, torch.from_numpy(np.repeat(np.arange(self.B, dtype=np.int32), counts)).to(fe.device))
            idx = self._idx_cache[1]
            _lib.check(fe.L.vofe_track_fb(fe.ctx, self.pyr._h, prev_ring * self.B, self.pyr._h, next_ring * self.B, io["prev"],
                                          idx.data_ptr(), self.B, io["next"], io["status"], io["err"], n, self.criteria[0],
                                          self.criteria[1], self.criteria[2], self.min_eig, self.fb_thresh))
            nxt, st, err = fe._points_done(io)
        out, o = [], 0
        for k in counts:
            if k == 0:
                out.append((np.empty((2, 0)), np.empty((0,), dtype=np.uint8), np.empty((0,), dtype=np.float32)))
                continue
            out.append((nxt[o:o + k].T, st[o:o + k], err[o:o + k]))
            o += k
        return out

    def detect(self, maxCorners, qualityLevel, minDistance, useHarrisDetector=False, k=0.04):
        """cv2.goodFeaturesToTrack on the current frame of every sequence in one batched call: B arrays (N_b,1,2) / None."""
        if self.frame_index < 0:
            raise error(-1, "MultiSequenceTracker.detect before reset()")
        out, cnt = self.fe.goodFeaturesToTrack_batch(self._img[self.frame_index & 1], maxCorners, qualityLevel, minDistance,
                                                     useHarrisDetector=useHarrisDetector, k=k)
        host = out.cpu().numpy()
        return [host[b, :c].reshape(-1, 1, 2).copy() if c else None for b, c in enumerate(cnt)]

    def close(self):
        if self.pyr is not None:
            self.pyr.free()
            self.pyr = None
        if self._own_fe and self.fe is not None:
            self.fe.close()
        self.fe = None
