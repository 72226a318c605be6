"""goodFeaturesToTrack per frame of the bench's synthetic KITTI sequence (why bench.py's gftt_us_per_call, which cycles
through the frames, is above scratch/gftt_bench.py's single-frame figure)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
fr = synthetic.sequence(H, W, seed, 8)
d = torch.from_numpy(np.stack(fr)).cuda()
for t in range(8):
    for _ in range(5): c = fe.goodFeaturesToTrack(d[t], n, 0.01, 7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): c = fe.goodFeaturesToTrack(d[t], n, 0.01, 7)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("frame %d: %.1f us/call, %d corners" % (t, dt * 1e6, len(c)), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200): c = fe.goodFeaturesToTrack(d[i % 8], n, 0.01, 7)
torch.cuda.synchronize(); print("cycling through the frames: %.1f us/call" % ((time.perf_counter() - t0) / 200 * 1e6))
os.environ["VOFE_GFTT_DEBUG"] = "1"
for t in (0, 3): fe.goodFeaturesToTrack(d[t], n, 0.01, 7)
fe.close()
