"""MI355X-native feature front end (Shi-Tomasi/Harris corners, Gaussian pyramid, pyramidal LK)
for the hot path of DiaHidvegi/visual-odometry.  The compute lives in libvofe.so (HIP, gfx950);
this package is the cv2-shaped host mirror.  Importing it needs neither a GPU nor the library;
constructing a FrontEnd needs both and fails loudly otherwise."""
from ._lib import error, LIB_PATH  # noqa: F401
from .framestate import FrameState  # noqa: F401
from .presets import TRACKING_PRESETS, BOOTSTRAP_PRESETS, Constants  # noqa: F401

TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_EPS = 2
OPTFLOW_USE_INITIAL_FLOW = 4
OPTFLOW_LK_GET_MIN_EIGENVALS = 8


def __getattr__(name):
    # torch-dependent names are resolved lazily so that CPU-only tooling can import the package
    if name in ("FrontEnd", "Pyramid", "goodFeaturesToTrack", "calcOpticalFlowPyrLK", "track_keypoints",
                "default_frontend"):
        from . import frontend
        return getattr(frontend, name)
    if name in ("SequenceTracker", "MultiSequenceTracker"):
        from . import sequence
        return getattr(sequence, name)
    if name in ("BootstrapTracker", "bootstrap_correspondences"):
        from . import bootstrap
        return getattr(bootstrap, name)
    if name == "DeviceFrameState":
        from . import device_state
        return device_state.DeviceFrameState
    raise AttributeError(name)
