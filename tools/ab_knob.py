"""In-process A/B of a launch-order knob (amdAprilTagsDebugSetTuning) on the B = 256 sigma-2 pipeline: round-robin over the values,
wall-clock step time (blocking C-ABI call, profiling off) and the fit stage's event time (profiling on), medians.
Usage (GPU box): python tools/ab_knob.py <knob> <v0,v1,...> [rounds] [B] [distinct]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
knob = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
distinct = int(sys.argv[5]) if len(sys.argv) > 5 else 64
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    frames = np.stack(list(ex.map(lambda i: synth.scene_c2(seed=1234 + i)[0], range(distinct))))
t = torch.from_numpy(frames).cuda().repeat((B + distinct - 1) // distinct, 1, 1)[:B].contiguous()
det = AprilTagDetector(1920, 1080, max_batch=B)
prep = det.prepare(t)
for _ in range(3):
    det.run_prepared(prep)
ref = None
step = {v: [] for v in vals}; fit = {v: [] for v in vals}; tot = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        det.set_tuning(knob, v)
        det.set_profiling(False)
        det.run_prepared(prep)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); det.run_prepared(prep); ts.append((time.perf_counter() - t0) * 1e3)
        step[v].append(min(ts))
        out = [(d["id"], d["p"].tobytes()) for f in det.unpack(prep) for d in f]
        if ref is None: ref = out
        assert out == ref, "results changed with knob value %d" % v
        det.set_profiling(True)
        det.run_prepared(prep)
        sm = det.stage_ms()
        fit[v].append(sm["fit_quads"]); tot[v].append(sum(sm.values()))
for v in vals:
    print("knob %d = %d: step ms median %.3f min %.3f | fit stage median %.3f | stage sum median %.3f" %
          (knob, v, np.median(step[v]), np.min(step[v]), np.median(fit[v]), np.median(tot[v])))
det.close()
