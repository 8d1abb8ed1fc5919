"""Quick GPU bring-up script (run through gpurun): device arithmetic self-check + stage parity on a
handful of scenes + a small timing.  The pytest -m gpu suite is the formal version of this."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from isaac_ros_apriltag_amd import capi, synth  # noqa: E402
if os.environ.get("AMDAT_LIB"):   # measurement variant (isaac_ros_apriltag_amd.build.build_amd_variant)
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector  # noqa: E402
import parity_util as pu  # noqa: E402


def math_check():
    rng = np.random.default_rng(0)
    a = np.abs(rng.standard_normal(100000)) * 10 ** rng.uniform(-6, 8, 100000)
    b = rng.standard_normal(100000) * 10 ** rng.uniform(-3, 3, 100000)
    ok = True
    r = capi.debug_math(0, a, b); ok &= np.array_equal(r, np.sqrt(a)); print("sqrt f64 exact:", np.array_equal(r, np.sqrt(a)))
    r = capi.debug_math(1, a, b); ok &= np.array_equal(r, a / b); print("div f64 exact:", np.array_equal(r, a / b))
    af, bf = a.astype(np.float32), b.astype(np.float32)
    r = capi.debug_math(2, a, b); e = np.sqrt(af).astype(np.float64); ok &= np.array_equal(r, e); print("sqrt f32 exact:", np.array_equal(r, e))
    r = capi.debug_math(3, a, b); e = (af / bf).astype(np.float64); ok &= np.array_equal(r, e); print("div f32 exact:", np.array_equal(r, e))
    g = np.arange(1 << 18, dtype=np.float64)
    r = capi.debug_math(5, g, g); ok &= np.array_equal(r, np.sqrt(g)); print("sqrt_u18 exact:", np.array_equal(r, np.sqrt(g)))
    return ok


def run_scene(name, img, K, families=("tag36h11",), decimate=1):
    h, w = img.shape
    det = AprilTagDetector(w, h, families=families, decimate=decimate, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]),
                           max_batch=1)
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    t0 = time.time()
    g = det.detect_batch_ex(t, max_dets=256)[0]
    dt = time.time() - t0
    errs, odets = pu.compare_stages(det, 0, img, families, K, decimate, verbose=True)
    errs += pu.compare_detections(g, odets)
    # the same frame through the throughput launch set (k_cc_local<4>, k_fit_small, per-wave prefilter, chunked select)
    det.set_submission_path("throughput")
    g2 = det.detect_batch_ex(t, max_dets=256)[0]
    e2, _ = pu.compare_stages(det, 0, img, families, K, decimate)
    errs += ["throughput set: " + e for e in e2 + pu.compare_detections(g2, odets)]
    det.set_submission_path("auto")
    print("%-28s dec %d: gpu dets %d, first-call %.1f ms, %s" % (name, decimate, len(g), dt * 1e3, "PARITY OK" if not errs else "MISMATCH"))
    for e in errs[:8]:
        print("      ", e)
    det.set_profiling(True)
    det.detect_batch_ex(t, max_dets=256)
    print("      stage ms:", {k: round(v, 3) for k, v in det.stage_ms().items()})
    # batched throughput: 16 copies of the frame
    det.close()
    B = 16
    detb = AprilTagDetector(w, h, families=families, decimate=decimate, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]),
                            max_batch=B)
    tb = t.unsqueeze(0).repeat(B, 1, 1).contiguous()
    detb.detect_batch_ex(tb, max_dets=64)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        rb = detb.detect_batch_ex(tb, max_dets=64)
    dt = (time.time() - t0) / 3
    same = all(len(r) == len(g) for r in rb)
    detb.set_profiling(True); detb.detect_batch_ex(tb, max_dets=64)
    print("      batch %d: %.2f ms/batch = %.0f fps (all frames same count: %s)" % (B, dt * 1e3, B / dt, same))
    print("      batch stage ms:", {k: round(v, 3) for k, v in detb.stage_ms().items()})
    detb.close()
    return not errs


def main():
    ok = math_check()
    img, K, _ = synth.scene_c1(); ok &= run_scene("c1", img, K); ok &= run_scene("c1", img, K, decimate=2)
    img, K, _ = synth.scene_pol_golden(); ok &= run_scene("pol_golden", img, K)
    img, K, _ = synth.scene_c2(sigma=0); ok &= run_scene("c2 clean", img, K)
    img, K, _ = synth.scene_c2(); ok &= run_scene("c2 sigma2", img, K); ok &= run_scene("c2 sigma2", img, K, decimate=2)
    img, K, _ = synth.scene_c5(); ok &= run_scene("c5 two families", img, K, families=("tag36h11", "tag25h9"))
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, size=(480, 644), dtype=np.uint8)
    ok &= run_scene("uniform noise 644x480", noise, synth.default_K(644, 480))
    odd = np.ascontiguousarray(synth.scene_c1()[0][:477, :635])
    ok &= run_scene("c1 cropped 635x477", odd, synth.default_K(635, 477))
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
