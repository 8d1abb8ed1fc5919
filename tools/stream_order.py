import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(8)])
t = torch.from_numpy(frames).cuda()
K = (1500.0, 1500.0, 960.0, 540.0)
def med(det, prep, n=40, warm=1):
    for _ in range(warm): det.run_prepared(prep)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); det.run_prepared(prep); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)
mode = sys.argv[1]
keep = []
if mode == "dummy_handle":
    keep.append(AprilTagDetector(64, 64, intrinsics=K, max_batch=1))
if mode == "dummy_handle_run":
    d = AprilTagDetector(64, 64, intrinsics=K, max_batch=1); keep.append(d)
    p = d.prepare(torch.zeros(1, 64, 64, dtype=torch.uint8, device="cuda")); d.run_prepared(p); d.run_prepared(p)
if mode == "streams":
    keep += [torch.cuda.Stream() for _ in range(4)]
if mode == "small_first_closed":
    d = AprilTagDetector(1920, 1080, intrinsics=K, max_batch=1)
    p = d.prepare(t[:1].contiguous()); d.run_prepared(p); d.run_prepared(p); d.close()
if mode.startswith("hip_"):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n = 4
    ss = [C.c_void_p() for _ in range(n)]
    for x in ss: assert hip.hipStreamCreateWithFlags(C.byref(x), 1) == 0
    if "work" in mode:
        buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
        for x in ss:
            assert hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), 0, 1 << 20, x) == 0
            assert hip.hipStreamSynchronize(x) == 0
    if "destroy" in mode:
        for x in ss: assert hip.hipStreamDestroy(x) == 0
    keep.append(ss)
bigB = 64 if mode != "big256" else 256
if mode != "nobig":
    big = AprilTagDetector(1920, 1080, intrinsics=K, max_batch=bigB)
    if mode != "big_idle":
        pb = big.prepare(t.repeat(bigB // 8, 1, 1).contiguous(), max_dets=64, intrinsics=[K] * bigB)
        for _ in range(3): big.run_prepared(pb)
torch.cuda.synchronize()
for B in (1, 8):
    small = AprilTagDetector(1920, 1080, intrinsics=K, max_batch=B)
    ps = small.prepare(t[:B].contiguous(), max_dets=64, intrinsics=[K] * B)
    print("%-20s B=%d fresh small handle: %.4f ms, graph_replay %s" % (mode, B, med(small, ps), small.graph_replay()))
    small.close()
