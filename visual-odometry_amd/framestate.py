"""FrameState container with the reference's field names, shapes and validation errors
(reference framestate.py:13-67), so that the CPU back end (PnP, triangulation) stays drop-in."""
from dataclasses import dataclass

import numpy as np


def _need(cond, msg):
    if not cond:
        raise ValueError(msg)


@dataclass
class FrameState:
    landmarks_image: np.ndarray               # (2, K)
    landmarks_world: np.ndarray               # (3, K)
    cand_landmarks_image_current: np.ndarray  # (2, M)
    cand_landmarks_image_first: np.ndarray    # (2, M)
    cand_landmarks_transform: np.ndarray      # (12, M) or (16, M)

    def __post_init__(self):
        P, X = self.landmarks_image, self.landmarks_world
        C, F, T = (self.cand_landmarks_image_current, self.cand_landmarks_image_first,
                   self.cand_landmarks_transform)
        _need(P.ndim == 2 and P.shape[0] == 2, f"'landmarks_image' must have dimensions 2 x K. Got shape {P.shape}.")
        _need(X.ndim == 2 and X.shape[0] == 3, f"'landmarks_world' must have dimensions 3 x K. Got shape {X.shape}.")
        _need(P.shape[1] == X.shape[1],
              f"'landmarks_image' and 'landmarks_world' must have shapes (2 x K) and (3 x K). Got {P.shape} and {X.shape}.")
        _need(C.ndim == 2 and C.shape[0] == 2,
              f"'cand_landmarks_image_current' must have dimensions 2 x M. Got shape {C.shape}.")
        _need(F.ndim == 2 and F.shape[0] == 2,
              f"'cand_landmarks_image_first' must have dimensions 2 x M. Got shape {F.shape}.")
        _need(T.ndim == 2 and T.shape[0] in (12, 16),
              f"'cand_landmarks_transform' must have dimensions (12 x M) or (16 x M). Got shape {T.shape}.")
        _need(C.shape[1] == F.shape[1] == T.shape[1],
              "'cand_landmarks_image_current', 'cand_landmarks_image_first', and 'cand_landmarks_transform' must all "
              f"refer to the same number of landmarks (M). Got shapes {C.shape}, {F.shape}, and {T.shape}.")

    def __str__(self):
        return (f"FrameState with {self.landmarks_image.shape[1]} landmarks and "
                f"{self.cand_landmarks_image_current.shape[1]} candidate landmarks")
