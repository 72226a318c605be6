"""Per-dataset front-end parameters of the reference: the continuous path reads constants.py:29-77
(+ TrackingParams continuous.py:22-29), the bootstrap path its own table initialization.py:185-189."""


class Constants:                                   # reference constants.py:5-10
    THRESHOLD_NEW_KEYPOINTS = 15
    THRESHOLD_PIXEL_DIST_TRIANGULATION = 15
    THRESHOLD_PIXEL_DIST_CANDIDATES_MIN = 2
    THRESHOLD_PIXEL_DIST_CANDIDATES_MAX = 100
    THRESHOLD_CANDIDATES_ALPHA = 1


_LK = dict(maxLevel=3, criteria=(3, 30, 0.01), minEigThreshold=0.001)
TRACKING_PRESETS = {
    "kitti": dict(maxCorners=1000, qualityLevel=0.01, minDistance=10, winSize=(21, 21), **_LK),
    "parking": dict(maxCorners=1000, qualityLevel=0.01, minDistance=10, winSize=(11, 11), **_LK),
    "malaga": dict(maxCorners=1000, qualityLevel=0.01, minDistance=10, winSize=(21, 21), **_LK),
}
_BOOT = dict(maxLevel=0, criteria=(3, 30, 0.01), minEigThreshold=1e-4)
BOOTSTRAP_PRESETS = {
    # dist_threshold_move: initialization.py:186-188; n_images: how many frames the bootstrap chains over (:180-182)
    "kitti": dict(maxCorners=1000, qualityLevel=0.01, minDistance=10, winSize=(11, 11), dist_threshold_move=2, n_images=3, **_BOOT),
    "parking": dict(maxCorners=1000, qualityLevel=0.01, minDistance=10, winSize=(11, 11), dist_threshold_move=0, n_images=7, **_BOOT),
    "malaga": dict(maxCorners=400, qualityLevel=0.01, minDistance=10, winSize=(41, 41), dist_threshold_move=0, n_images=5, **_BOOT),
}
