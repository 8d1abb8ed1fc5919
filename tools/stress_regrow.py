"""Hammers the capacity-regrowth paths of a handle (candidate list grown inside a call, pair table grown before the next one, point
buffers grown and the submission repeated) in one process, handle after handle, as the GPU suite does between other tests.
Usage (GPU box): python tools/stress_regrow.py [iterations] [graph|nograph]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
w, h = 960, 720
img = np.full((h, w), 220, dtype=np.uint8)
for y0 in range(0, h - 5, 6):
    img[y0:y0 + 5] = np.where((np.arange(w) % 6) < 5, 30, 220).astype(np.uint8)[None, :]
img[h - h % 6:] = 220
yy = np.mgrid[0:480, 0:640][0]
stripes = np.where(yy % 2 == 0, 40, 215).astype(np.uint8)
K = synth.default_K(w, h)
t = torch.from_numpy(img).cuda()
ts = torch.from_numpy(np.ascontiguousarray(stripes)).cuda()
other = torch.from_numpy(synth.scene_c1()[0]).cuda()
ref = None
late = 0
for it in range(iters):
    det = AprilTagDetector(w, h, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]), max_batch=1)
    if mode == "nograph":
        det.set_profiling(True)
    trace = os.environ.get("STRESS_TRACE")
    def step(msg):
        if trace:
            print("it %d: %s" % (it, msg), flush=True)
    b0 = det.device_bytes()
    step("call 1")
    r1 = det.detect_batch_ex(t, max_dets=64)[0]
    b1 = det.device_bytes()
    f1 = det.frame_flags(1)
    step("call 2 (bytes %d -> %d, flags %s)" % (b0, b1, f1))
    r2 = det.detect_batch_ex(t, max_dets=64)[0]
    b2 = det.device_bytes()
    f2 = det.frame_flags(1)
    step("done (bytes %d, flags %s)" % (b2, f2))
    if not (b1 > b0 and b2 > b1 and f2 == [0]):
        print("UNEXPECTED it %d: bytes %d %d %d flags %s %s counts %s" % (it, b0, b1, b2, f1, f2, det.debug(0, capi.DBG_COUNTS)), flush=True)
    cnt = tuple(int(v) for v in det.debug(0, capi.DBG_COUNTS)[:5])
    if ref is None:
        ref = cnt
    if cnt != ref:
        print("UNEXPECTED it %d: counts %s vs %s" % (it, cnt, ref), flush=True)
    late += det.late_waits()
    det.close()
    d2 = AprilTagDetector(640, 480, max_batch=1)     # point buffers grow, submission repeated
    if mode == "nograph":
        d2.set_profiling(True)
    step("stripes")
    d2.detect_batch_ex(ts, max_dets=64)
    if d2.frame_flags(1) != [0]:
        print("UNEXPECTED it %d: stripes flags %s" % (it, d2.frame_flags(1)), flush=True)
    late += d2.late_waits()
    d2.close()
    d3 = AprilTagDetector(640, 480, decimate=2, max_batch=1)   # an ordinary handle in between
    step("plain")
    d3.detect_batch_ex(other, max_dets=64)
    late += d3.late_waits()
    d3.close()
print("stress_regrow: %d iterations (%s) ok, counts %s, late stream waits %d" % (iters, mode, ref, late))
