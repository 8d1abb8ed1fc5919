"""Throughput of BASELINE.json configs 3 and 5 (and config 2 with decimate 2), same protocol as bench.py:
device-resident frames, one blocking C-ABI call per step, parity gate against the CPU oracle on one frame."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from isaac_ros_apriltag_amd import synth
from isaac_ros_apriltag_amd.detector import AprilTagDetector
import parity_util as pu
from oracle import pyoracle as po


def rate(name, frames, K, families, decimate, tag_size, B, steps=5):
    h, w = frames[0].shape
    t = torch.from_numpy(np.stack(frames)).cuda()
    batch = t.repeat((B + len(frames) - 1) // len(frames), 1, 1)[:B].contiguous()
    det = AprilTagDetector(w, h, families=families, decimate=decimate, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]),
                           tag_size=tag_size, max_batch=B)
    prep = det.prepare(batch, max_dets=128)
    det.run_prepared(prep); det.run_prepared(prep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): det.run_prepared(prep)
    dt = (time.perf_counter() - t0) / steps
    out = det.unpack(prep)
    o, _ = po.detect(frames[0], families=families, params=pu.oracle_params(K, decimate, tag_size), max_det=1024)
    ok = not pu.compare_detections(out[0], o[:128])
    det.set_profiling(True); det.run_prepared(prep)
    st = {k: round(v, 3) for k, v in det.stage_ms().items()}
    print("%-34s B=%d: %.2f ms/step, %.0f frames/s, dets/frame %.1f, parity %s, flags %s\n    stages %s" %
          (name, B, dt * 1e3, B / dt, np.mean([len(x) for x in out]), "pass" if ok else "FAIL", sum(det.frame_flags(B)), st))
    det.close()


c2 = [synth.scene_c2(seed=1234 + i)[0] for i in range(8)]
K2 = synth.default_K(1920, 1080)
rate("config 2, decimate 2 (sigma 2)", c2, K2, ("tag36h11",), 2, 0.22, 64)
c5 = [synth.scene_c5(seed=1234 + i)[0] for i in range(8)]
rate("config 5, two families (sigma 2)", c5, K2, ("tag36h11", "tag25h9"), 1, 0.22, 64)
img3, K3, _, size3 = synth.scene_c3()
rate("config 3, 4K 100 tags, decimate 2", [img3, synth.scene_c3(seed=78)[0]], K3, ("tag36h11",), 2, size3, 32)
