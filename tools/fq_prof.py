"""Per-phase shader-cycle shares of the quad-fit kernel per size class (needs the -DAMDAT_FQ_PROFILE build of the library:
python -c "from isaac_ros_apriltag_amd import build as b; b.build_amd_variant('prof', ['AMDAT_FQ_PROFILE'])").
Usage: python tools/fq_prof.py [B]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_prof.so")
from isaac_ros_apriltag_amd import synth
from isaac_ros_apriltag_amd.detector import AprilTagDetector
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(8)])
t = torch.from_numpy(frames).cuda().repeat(B // 8, 1, 1).contiguous()
det = AprilTagDetector(1920, 1080, max_batch=B)
det.detect_batch_ex(t)
det.set_profiling(2)
det.detect_batch_ex(t)
print({k: round(v, 3) for k, v in det.stage_ms().items()})
pa = det.debug(0, 8).view(np.uint64)[:40].astype(np.float64).reshape(5, 8)
names = ["pop+load", "bbox+dot", "keys+sort", "presort-test", "dedup+terms+prefix", "errs+smooth+maxima", "select", "pairs+combos+final"]
tot = pa.sum()
for c in range(5):
    p = pa[c]
    print("class %d: %5.1f%% of all fit cycles: " % (c, 100 * p.sum() / max(tot, 1)) + ", ".join("%s %.0f%%" % (n, 100 * v / max(p.sum(), 1)) for n, v in zip(names, p) if n != "-"))
cnt = det.debug(0, 8).view(np.uint64)[40:60].astype(np.float64).reshape(5, 4)
for c in range(5):
    print("class %d: points reaching the pre-sort test %.0f, rejected there %.0f, rejected after walk 1 %.0f (per frame)" % (c, cnt[c][0] / B, cnt[c][1] / B, cnt[c][2] / B))
pf = det.debug(0, 8).view(np.uint64)[60:64].astype(np.float64)
print("prefilter cycles: box+dot %.0f%%, sector sums %.0f%%, scan + 32-sector test %.0f%%, 64-sector test %.0f%%" % tuple(100 * pf / max(pf.sum(), 1)))
pt = det.debug(0, 8).view(np.uint64)[64:70].astype(np.float64)
print("k_points cycles: tile load %.0f%%, emission tests + scan %.0f%%, list %.0f%%, block table + staging stores %.0f%%, leftovers + barrier %.0f%%, frame table %.0f%%" % tuple(100 * pt / max(pt.sum(), 1)))
det.close()
