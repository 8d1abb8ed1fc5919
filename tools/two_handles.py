"""Do two half-batch submissions in flight at once beat one full batch?  Two handles (B/2 frames each) driven by two
host threads, the second started half a step later, against one handle with B frames."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import capi, synth
if os.environ.get("AMDAT_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(16)])
t = torch.from_numpy(frames).cuda().repeat(B // 16, 1, 1).contiguous()
det = AprilTagDetector(1920, 1080, max_batch=B)
prep = det.prepare(t)
for _ in range(3): det.run_prepared(prep)
t0 = time.perf_counter()
for _ in range(6): det.run_prepared(prep)
one = 6 * B / (time.perf_counter() - t0)
det.close()
hb = B // nh
dets = [AprilTagDetector(1920, 1080, max_batch=hb) for _ in range(nh)]
preps = [d.prepare(t[i * hb:(i + 1) * hb]) for i, d in enumerate(dets)]
for d, p in zip(dets, preps):
    d.run_prepared(p)
steps = 12
def worker(i):
    time.sleep(i * float(os.environ.get('STAGGER_MS', '9')) * 1e-3)
    for _ in range(steps): dets[i].run_prepared(preps[i])
ths = [threading.Thread(target=worker, args=(i,)) for i in range(nh)]
t0 = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
two = nh * steps * hb / (time.perf_counter() - t0)
print("one handle B=%d: %.0f fps; %d handles B=%d each, concurrent: %.0f fps" % (B, one, nh, hb, two))
