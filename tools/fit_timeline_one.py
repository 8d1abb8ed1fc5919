"""Per-cluster wall-clock intervals of the quad fit for ONE 1080p sigma-2 frame (tools-only build -DAMDAT_FQ_TIMELINE):
when each size class's workgroups start and end relative to the first one, and the slowest clusters.
Usage: AMDAT_LIB=tl python tools/fit_timeline_one.py"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ.get("AMDAT_LIB", "tl"))
from isaac_ros_apriltag_amd.detector import AprilTagDetector
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = synth.scene_c2(seed=1234)[1]
t = torch.from_numpy(np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(B)])).cuda()
det = AprilTagDetector(1920, 1080, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]), max_batch=B)
prep = det.prepare(t)
L = capi.lib()
L.amdAprilTagsDebugTimeline.argtypes = [C.c_void_p, C.c_uint]
buf = np.zeros((1 << 16, 2), dtype=np.uint64)
for _ in range(6):
    det.run_prepared(prep)
L.amdAprilTagsDebugTimeline(buf.ctypes.data, 1 << 16)
span = np.zeros(4, dtype=np.uint64)
L.amdAprilTagsDebugTimelineSpan.argtypes = [C.c_void_p]
L.amdAprilTagsDebugTimelineSpan(span.ctypes.data)
det.run_prepared(prep)
L.amdAprilTagsDebugTimelineSpan(span.ctypes.data)
ph = np.zeros((1 << 16, 8), dtype=np.uint32)
L.amdAprilTagsDebugTimelinePhases.argtypes = [C.c_void_p, C.c_uint]
L.amdAprilTagsDebugTimelinePhases(ph.ctypes.data, 1 << 16)
n = L.amdAprilTagsDebugTimeline(buf.ctypes.data, 1 << 16)
b = buf[:n]
ph = ph[:n]
t0 = b[:, 0].astype(np.int64); dur = (b[:, 1] >> np.uint64(32)).astype(np.int64); nt = ((b[:, 1] >> np.uint64(20)) & np.uint64(0xFFF)).astype(int)
sz = (b[:, 1] & np.uint64(0xFFFFF)).astype(int)
base = t0.min()
tick = 0.01  # us per tick (100 MHz)
print("%d clusters logged; span %.1f us" % (n, (t0 + dur).max() * tick - base * tick))
if span[1] > 0:
    print("prefilter: first block start %.1f us, last block end %.1f us; k_quad_finish first block start %.1f us (all relative to the first logged cluster start)" %
          ((int(span[0]) - int(base)) * tick, (int(span[1]) - int(base)) * tick, (int(span[3]) - int(base)) * tick))
for c in sorted(set(nt)):
    m = nt == c
    print("NT=%4d: %5d clusters, first start %.1f us, last start %.1f us, last end %.1f us, mean dur %.1f us, max dur %.1f us (sz %d)" %
          (c, m.sum(), (t0[m].min() - base) * tick, (t0[m].max() - base) * tick, ((t0 + dur)[m].max() - base) * tick, dur[m].mean() * tick,
           dur[m].max() * tick, sz[m][dur[m].argmax()]))
    ends = np.sort((t0 + dur)[m] - base) * tick
    print("      ends: 50%% %.1f  90%% %.1f  99%% %.1f  100%% %.1f us; busy wave-us %.0f" % (ends[len(ends) // 2], ends[int(len(ends) * .9)], ends[int(len(ends) * .99)], ends[-1], dur[m].sum() * tick))
    order = np.argsort(-(t0 + dur)[m])[:5]
    for k in order:
        print("      sz %5d start %.1f dur %.1f end %.1f" % (sz[m][k], (t0[m][k] - base) * tick, dur[m][k] * tick, ((t0 + dur)[m][k] - base) * tick))
    names = ["pop+load", "bbox+dot", "keys+sort", "presort", "sweep", "errors+maxima", "top-10", "pairs+corners"]
    for lo, hi in ((24, 64), (65, 128), (129, 256), (257, 512), (513, 768), (769, 2048), (2049, 4096), (4097, 1 << 20)):
        mm = m & (sz >= lo) & (sz <= hi)
        if mm.sum():
            print("      sizes %5d..%-6d n %5d mean dur %.1f us max %.1f" % (lo, hi, mm.sum(), dur[mm].mean() * tick, dur[mm].max() * tick))
            full = mm & (ph[:, 7] > 0)   # clusters that went through every phase
            if full.sum():
                print("            (%d reached the corner search) mean us per phase: " % full.sum() +
                      ", ".join("%s %.1f" % (names[j], ph[full, j].mean() * tick) for j in range(8) if j != 3))
det.close()
