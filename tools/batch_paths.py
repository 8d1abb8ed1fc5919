"""Small-batch submissions of 1080p sigma-2 frames through the C ABI: median / min call time per batch size on the AUTO path
(graph replay where the library uses it), then the per-stage HIP-event times of one profiled submission.  For A/B runs of
measurement builds (AMDAT_LIB=<tag>) of the launch-set switch.  Usage: python tools/batch_paths.py "1 2 4 8 16 32" [calls]"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
Bs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1 8").split()]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(max(Bs))])
tall = torch.from_numpy(frames).cuda()
for B in Bs:
    det = AprilTagDetector(1920, 1080, max_batch=B, **({"no_graph_replay": 1} if os.environ.get("NOGRAPH") else {}))
    prep = det.prepare(tall[:B].contiguous())
    for _ in range(10):
        det.run_prepared(prep)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter(); det.run_prepared(prep); ts.append(time.perf_counter() - t0)
    det.set_profiling(True)
    det.run_prepared(prep); det.run_prepared(prep)
    st = {k: round(v, 3) for k, v in det.stage_ms().items()}
    print(json.dumps({"lib": os.environ.get("AMDAT_LIB", "default") + ("/nograph" if os.environ.get("NOGRAPH") else ""), "B": B, "ms_median": round(float(np.median(ts)) * 1e3, 4),
                      "ms_min": round(float(np.min(ts)) * 1e3, 4), "fps": round(B / float(np.median(ts)), 1),
                      "path": det.last_submission_path() if hasattr(det, "last_submission_path") else None, "stages": st}))
    det.close()
