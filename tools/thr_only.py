"""Runs only the threshold pass on a 160-frame 1080p batch (for rocprofv3 --pmc runs).
Usage: python tools/thr_only.py [decimate] [encoding]   (encoding: mono8 | rgb8 | bgr8 | rgba8 | bgra8; colour frames carry the gray
value in every channel)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import synth, capi
if os.environ.get("AMDAT_LIB"):   # measurement variant (isaac_ros_apriltag_amd.build.build_amd_variant)
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
dec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
enc = sys.argv[2] if len(sys.argv) > 2 else "mono8"
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(8)])
big = torch.from_numpy(frames).cuda().repeat(20, 1, 1).contiguous()
if enc != "mono8":
    big = big.unsqueeze(-1).expand(-1, -1, -1, capi.ENC_CHANNELS[enc]).contiguous()
det = AprilTagDetector(1920, 1080, decimate=dec, max_batch=160, max_points=4096, hash_slots=256, max_clusters=256, max_quads=64, max_detections=16)
det.set_profiling(True)
ms = []
for _ in range(10):
    det.threshold_only(big, encoding=enc)
    ms.append(det.stage_ms()["threshold"])
print("threshold (%s) ms per 160-frame launch:" % enc, [round(m, 4) for m in ms])
det.close()
