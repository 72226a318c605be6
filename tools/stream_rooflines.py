"""The streaming kernels' rooflines alone (what bench.py reports as roofline_streaming), on 128 and 2048 KITTI frames per
launch and on a 4K pair; HIP-event bracketed through vofe_profile_begin / vofe_profile_end.
usage: python3 tools/stream_rooflines.py [frames ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                            # noqa: E402
from visual_odometry_amd import synthetic, _lib                         # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid              # noqa: E402

fe = FrontEnd(0)
H, W, seed, _, _, _ = synthetic.CONFIGS["kitti"]
fr = synthetic.sequence(H, W, seed, 8)
d = torch.from_numpy(np.stack(fr)).cuda()
for frames in [int(a) for a in sys.argv[1:]] or [128, 2048]:
    imgs = d[[k % 8 for k in range(frames)]].contiguous()
    for e in bench.streaming_rooflines(torch, fe, _lib, Pyramid, imgs, H, W):
        print("%5d frames  %-26s %9.2f us  %7.1f GB/s  %5.1f %% of 8 TB/s" % (frames, e["kernel"], e["avg_launch_ms"] * 1e3,
                                                                            e.get("achieved", 0), 100 * e.get("frac", 0)))
    del imgs
fe.close()
