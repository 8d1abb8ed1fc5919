"""One-frame C-ABI latency of a 1080p sigma-2 frame in graph mode (median / min of 200 calls), for A/B runs of measurement
builds (AMDAT_LIB=<tag>).  Usage: python tools/latency_one.py [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1234
img, K, _ = synth.scene_c2(seed=seed)
t = torch.from_numpy(img).cuda()
det = AprilTagDetector(1920, 1080, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]), max_batch=1)
prep = det.prepare(t)
for _ in range(10):
    det.run_prepared(prep)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); det.run_prepared(prep); ts.append(time.perf_counter() - t0)
print("median %.3f min %.3f ms" % (float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3))
det.close()
