"""bench.py on a measurement build of the library: AMDAT_LIB=<tag> python tools/bench_variant.py [bench.py's flags] (prints value, step, fit, points, sweep)."""
import os, sys, io, json, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
import bench
_keep = []
if os.environ.get("SMALL_FIRST"):   # a one-frame handle created (and used) before the bench's 256-frame handle
    import numpy as np, torch
    from isaac_ros_apriltag_amd.detector import AprilTagDetector
    from isaac_ros_apriltag_amd import synth
    d = AprilTagDetector(1920, 1080, max_batch=1)
    p = d.prepare(torch.from_numpy(synth.scene_c2(seed=5)[0][None]).cuda())
    d.run_prepared(p); d.run_prepared(p)
    _keep.append(d)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
st = d["stage_ms_per_step"]
print(os.environ.get("AMDAT_LIB", "default") + ("/small-first" if _keep else ""), d["value"], d["ms_per_step"], "fit", st["fit_quads"], "points", st["points"], "cc_local", st["cc_local"],
      "sweep", {k: v["ms_median"] for k, v in d["extra"]["batch_sweep"].items()}, d["parity_gate"])
