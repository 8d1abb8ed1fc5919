"""Single-frame latency and per-stage times (B=1 and B=64) for clean and noisy 1080p frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import synth
from isaac_ros_apriltag_amd.detector import AprilTagDetector
for sigma in (0.0, 2.0):
    frames = np.stack([synth.scene_c2(seed=1234 + i, sigma=sigma)[0] for i in range(4)])
    for B in (1, 8, 16, 64):
        t = torch.from_numpy(frames).cuda().repeat((B + 3) // 4, 1, 1)[:B].contiguous()
        det = AprilTagDetector(1920, 1080, max_batch=B)
        prep = det.prepare(t)
        for _ in range(3): det.run_prepared(prep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): det.run_prepared(prep)
        dt = (time.perf_counter() - t0) / 20
        det.set_profiling(True); det.run_prepared(prep)
        st = det.stage_ms()
        print("sigma %.0f B=%d: %.3f ms per call (%.0f fps); stages: %s" % (sigma, B, dt * 1e3, B / dt, {k: round(v, 3) for k, v in st.items()}))
        det.close()
