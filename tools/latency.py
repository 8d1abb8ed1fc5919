"""Single-frame latency (B = 1): the C-ABI call on a device-resident frame with its per-stage times, and the node shell
fed a host image (sensor_msgs/Image stand-in: H2D copy + detection + message assembly) -- 720p (the size the reference's
published node numbers use, README.md:65-70) and 1080p, clean and sigma-2 frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import synth
from isaac_ros_apriltag_amd import node as nd
from isaac_ros_apriltag_amd.detector import AprilTagDetector


def scene(w, h, sigma, seed=1234):
    img, K, _ = synth._grid_scene(w, h, [("tag36h11", i) for i in range(10)], 5, 2, seed, 96 * h / 1080, 192 * h / 1080, 30, 25, sigma)
    return img, K


rows = []
for (w, h) in ((1280, 720), (1920, 1080)):
    for sigma in (0.0, 2.0):
        img, K = scene(w, h, sigma)
        t = torch.from_numpy(img).cuda()
        det = AprilTagDetector(w, h, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]), max_batch=1)
        prep = det.prepare(t)
        for _ in range(5):
            det.run_prepared(prep)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter(); det.run_prepared(prep); ts.append(time.perf_counter() - t0)
        det.set_profiling(True); det.run_prepared(prep)
        st = det.stage_ms()
        ndet = len(det.unpack(prep)[0])
        det.close()
        n = nd.AprilTagNode(backends="CUDA")
        K9 = [K[0, 0], 0, K[0, 2], 0, K[1, 1], K[1, 2], 0, 0, 1]
        for _ in range(5):
            n.on_frame(img.ctypes.data, False, "mono8", w, h, w, K9)
        tn = []
        for _ in range(50):
            t0 = time.perf_counter(); d, _ = n.on_frame(img.ctypes.data, False, "mono8", w, h, w, K9); tn.append(time.perf_counter() - t0)
        n.close()
        row = {"size": "%dx%d" % (w, h), "sigma": sigma, "tags": ndet, "c_abi_ms_median": round(float(np.median(ts)) * 1e3, 3),
               "c_abi_ms_min": round(float(np.min(ts)) * 1e3, 3), "node_shell_host_image_ms_median": round(float(np.median(tn)) * 1e3, 3),
               "node_fps": round(1.0 / float(np.median(tn)), 1), "stage_ms": {k: round(v, 3) for k, v in st.items()}}
        rows.append(row)
        print(row)
import json
print(json.dumps(rows))
