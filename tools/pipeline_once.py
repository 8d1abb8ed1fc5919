"""Runs the full pipeline a few times on B sigma-2 1080p frames (for rocprofv3 kernel-trace / --pmc passes, which
serialise kernels and are slow on a full bench).  Usage: python tools/pipeline_once.py [B] [reps] [distinct]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):   # measurement variant built by isaac_ros_apriltag_amd.build.build_amd_variant
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 8
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(distinct)])
t = torch.from_numpy(frames).cuda().repeat((B + distinct - 1) // distinct, 1, 1)[:B].contiguous()
det = AprilTagDetector(1920, 1080, max_batch=B)
prep = det.prepare(t)
for _ in range(reps):
    det.run_prepared(prep)
det.set_profiling(True)
det.run_prepared(prep)
print("B=%d stages:" % B, {k: round(v, 3) for k, v in det.stage_ms().items()})
print("counts:", det.mean_counts(B), "flags:", sum(1 for f in det.frame_flags(B) if f), "bytes:", det.device_bytes())
det.close()
