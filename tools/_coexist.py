import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
from isaac_ros_apriltag_amd.detector import AprilTagDetector
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(8)])
t = torch.from_numpy(frames).cuda()
def med(det, prep, n=100):
    for _ in range(10): det.run_prepared(prep)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); det.run_prepared(prep); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)
for B in (1, 8):
    small = AprilTagDetector(1920, 1080, max_batch=B)
    ps = small.prepare(t[:B].contiguous())
    print("B=%d alone" % B, med(small, ps))
    big = AprilTagDetector(1920, 1080, max_batch=256)
    print("B=%d with an idle 64-frame handle (prioritised side streams) alive" % B, med(small, ps))
    pb = big.prepare(t.repeat(32, 1, 1).contiguous())
    big.run_prepared(pb); big.run_prepared(pb)
    print("B=%d after that handle ran twice" % B, med(small, ps))
    fresh = AprilTagDetector(1920, 1080, max_batch=B)
    pf = fresh.prepare(t[:B].contiguous())
    fresh.run_prepared(pf)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); fresh.run_prepared(pf); ts.append(time.perf_counter() - t0)
    print("B=%d FRESH handle created next to it, 1 warm-up, 40 calls: median %.4f first5 %s" % (B, float(np.median(ts)) * 1e3, [round(x * 1e3, 3) for x in ts[:5]]))
    fresh.close()
    big.close()
    print("B=%d after it was destroyed" % B, med(small, ps))
    other = AprilTagDetector(1920, 1080, max_batch=8)
    print("B=%d with an idle 8-frame handle (plain streams) alive" % B, med(small, ps))
    other.close(); small.close()
