"""GPU-vs-oracle fuzzer: adversarial frame content x ragged sizes x pitches x decimation x families.

Run on the GPU box:  python tools/fuzz_gpu.py [--cases N] [--seed S]
Every case compares every stage (gray, threshold, labels, sizes, clusters, points, quads) and the final
detections bit-for-bit through the C ABI.  Prints one line per failing case and a summary; exit code 1 on
any mismatch.  Test infrastructure (uses oracle/).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

import torch  # noqa: E402

from isaac_ros_apriltag_amd import capi, synth  # noqa: E402
if os.environ.get("AMDAT_LIB"):   # measurement / stress variant (isaac_ros_apriltag_amd.build.build_amd_variant)
    capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])
from isaac_ros_apriltag_amd.detector import AprilTagDetector  # noqa: E402
import parity_util as pu  # noqa: E402


def gen_content(rng, h, w):
    kind = rng.integers(0, 12)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:    # checkerboard with random cell size 1..9
        c = int(rng.integers(1, 10))
        img = (((yy // c) + (xx // c)) & 1) * int(rng.integers(30, 255))
    elif kind == 1:  # stripes (h or v) with random period and phase
        p = int(rng.integers(2, 24))
        a = xx if rng.integers(0, 2) else yy
        img = ((a + int(rng.integers(0, p))) % p < p // 2) * 220 + 10
    elif kind == 2:  # smooth gradient + faint noise (low contrast tiles -> 127)
        img = (xx * 255.0 / max(w - 1, 1)) + rng.normal(0, 1.0, (h, w))
    elif kind == 3:  # saturated blobs: random rectangles black/white on grey
        img = np.full((h, w), 128.0)
        for _ in range(int(rng.integers(1, 40))):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = min(w, x0 + int(rng.integers(1, 80))), min(h, y0 + int(rng.integers(1, 80)))
            img[y0:y1, x0:x1] = rng.choice([0, 255, 60, 200])
    elif kind == 4:  # constant
        img = np.full((h, w), float(rng.integers(0, 256)))
    elif kind == 5:  # uniform noise
        img = rng.integers(0, 256, (h, w)).astype(np.float64)
    elif kind == 6:  # sparse impulses on flat background
        img = np.full((h, w), 100.0)
        m = rng.random((h, w)) < 0.02
        img[m] = 255
    elif kind == 7:  # diagonal lines
        p = int(rng.integers(3, 17))
        img = ((xx + yy * int(rng.integers(1, 4))) % p == 0) * 255.0
    elif kind == 8:  # concentric rings (many nested components, long boundaries)
        cx, cy = w / 2.0, h / 2.0
        r = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
        img = ((r / float(rng.integers(2, 12))).astype(np.int64) & 1) * 240.0 + 8
    elif kind == 9:  # tags on a busy background
        img = None
    elif kind == 10:  # two-level noise: every pixel either 0 or 255
        img = (rng.random((h, w)) < rng.uniform(0.2, 0.8)) * 255.0
    else:            # big solid shapes with noisy edges (large clusters)
        img = np.full((h, w), 40.0)
        img[h // 6: 5 * h // 6, w // 6: 5 * w // 6] = 220
        img += rng.normal(0, float(rng.uniform(0, 30)), (h, w))
    return kind, img


def tag_scene(rng, h, w, fams):
    tags = []
    n = int(rng.integers(1, 6))
    K = synth.default_K(w, h)
    for _ in range(n):
        fam = fams[int(rng.integers(0, len(fams)))]
        ncodes = len(synth.family_codes(fam)[0])
        tid = int(rng.integers(0, ncodes))
        side = float(rng.uniform(0.12, 0.5)) * min(w, h)
        z = K[0, 0] * 0.22 / side
        cxp, cyp = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h
        t = np.array([(cxp - K[0, 2]) * z / K[0, 0], (cyp - K[1, 2]) * z / K[1, 1], z])
        R = synth.rot_xyz(*(np.deg2rad(rng.uniform(-45, 45, 2)).tolist() + [float(rng.uniform(-np.pi, np.pi))]))
        H = synth.homography_from_pose(R, t, K, 0.22)
        tags.append({"family": fam, "id": tid, "H": H})
    img = synth.render(w, h, tags, background=int(rng.integers(60, 220)), sigma=float(rng.uniform(0, 6)),
                       seed=int(rng.integers(0, 1 << 30)), ss=2)
    return img


def colour_frame(rng, gray_like, encoding, pitch_pad):
    """A colour frame with random chroma around the content (so that a swapped channel order cannot pass) and garbage behind every
    row; returns (host buffer [h, pitch bytes], pitch, the numpy-converted gray frame -- the fixed-point BT.601 statement)."""
    h, w = gray_like.shape
    nch = capi.ENC_CHANNELS[encoding]
    order = (0, 1, 2) if encoding in ("rgb8", "rgba8") else (2, 1, 0)
    base = gray_like.astype(np.int32)
    amp = int(rng.integers(0, 60))
    rgb = np.stack([np.clip(base + rng.integers(-amp, amp + 1, size=(h, w)), 0, 255) for _ in range(3)], axis=2).astype(np.uint8)
    pitch = w * nch + pitch_pad
    buf = rng.integers(0, 256, size=(h, pitch), dtype=np.uint8)
    px = buf[:, :w * nch].reshape(h, w, nch)
    for c in range(3):
        px[..., order[c]] = rgb[..., c]
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return buf, pitch, ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8)


MAX_DETS = 256   # records a call asks for: a frame with more (noise fields under tag16h5) comes back as the first MAX_DETS of the canonical order


def run_cases(cases, seed, maxdim=420, budget=1e9, out=print, path=None, batch=1, only=None, dump=None, colour=False, tile=4, params=False, layout=False):
    """Returns (cases run, list of failure strings).  path: None (the library picks the launch set by size: the latency set at these
    sizes), "latency", "throughput", or "alternate" (even cases latency, odd cases throughput).  batch > 1: every case submits `batch` frames
    of the case's size, each with content of its own, in ONE call, and every frame is compared (frame indexing of every stage)."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    fails = []
    done = 0
    for case in range(cases):
        if time.time() - t0 > budget:
            break
        h = int(rng.integers(4, maxdim))
        w = int(rng.integers(4, maxdim))
        if rng.random() < 0.15:
            w = int(rng.integers(4, 40))
        if rng.random() < 0.15:
            h = int(rng.integers(4, 40))
        dec = int(rng.choice([1, 1, 1, 2, 2, 3, 4]))
        if (w // dec) < 4 or (h // dec) < 4:
            dec = 1
        fams = [("tag36h11",), ("tag25h9",), ("tag16h5",), ("tag36h11", "tag25h9"), ("tag36h11", "tag25h9", "tag16h5")][
            int(rng.integers(0, 5))]
        pitch = w + int(rng.choice([0, 0, 1, 3, 13, 64]))
        imgs, bufs, kinds = [], [], []
        for _ in range(batch):
            kind, img = gen_content(rng, h, w)
            if img is None:
                img = tag_scene(rng, h, w, fams)
            img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
            buf = np.zeros((h, pitch), dtype=np.uint8)
            buf[:, :w] = img
            buf[:, w:] = rng.integers(0, 256, (h, pitch - w), dtype=np.uint8)
            imgs.append(img); bufs.append(buf); kinds.append(int(kind))
        kind = kinds[0] if batch == 1 else tuple(kinds)
        if only is not None and case != only:   # (the generator has consumed exactly what the full run consumes)
            continue
        if dump:
            np.savez(dump, imgs=np.stack(imgs), pitch=pitch, dec=dec, fams=np.array(fams), w=w, h=h)
        K = synth.default_K(w, h)
        more, tag_size = {}, 0.22
        if params:   # the decode parameters beside their defaults (refine_edges off, every max_hamming, sharpening, skew, tag size)
            prng = np.random.default_rng(seed * 1000003 + case)   # (a generator of its own: the content stream stays the plain run's)
            more = {"refine_edges": int(prng.random() < 0.5), "max_hamming": int(prng.integers(0, 4)),
                    "decode_sharpening": float(np.float32(prng.choice([0.0, 0.1, 0.25, 0.5, 1.0]))),
                    "skew": float(np.float32(prng.choice([0.0, 0.0, 1.5, -3.25])))}
            tag_size = float(np.float32(prng.choice([0.22, 0.05, 1.0])))
        try:
            det = AprilTagDetector(w, h, intrinsics=(K[0, 0], K[1, 1], K[0, 2], K[1, 2]), families=fams, decimate=dec,
                                   max_batch=batch, tile_size=tile, tag_size=tag_size, **more)
        except Exception as e:  # noqa: BLE001
            if tile > 4 and min(1 + (w - 1) // dec, 1 + (h - 1) // dec) < tile:
                continue   # (a working image below one tile a side is refused at creation: AMDAT_UNSUPPORTED, by design)
            fails.append("case %d: create failed for %dx%d dec %d: %s" % (case, w, h, dec, e))
            out(fails[-1])
            continue
        if path is not None:
            det.set_submission_path(("latency", "throughput")[case & 1] if path == "alternate" else path)
        enc = "mono8"
        if colour:   # the same content as interleaved colour frames (one encoding per submission), read by the threshold pass's loader
            enc = ("rgb8", "bgr8", "rgba8", "bgra8")[int(rng.integers(0, 4))]
            pad = int(rng.choice([0, 0, 1, 3, 16, 37]))
            cf = [colour_frame(rng, im, enc, pad) for im in imgs]
            bufs = [c[0] for c in cf]; pitch = cf[0][1]; imgs = [c[2] for c in cf]
        ptrs = None
        if layout:   # every frame of the submission at a base address and a pitch of its own (a region of a larger device buffer)
            lrng = np.random.default_rng(seed * 7919 + case)
            nch = capi.ENC_CHANNELS[enc]
            flat, ptrs = [], []
            for b in bufs:
                off = int(lrng.integers(1, 16)) if lrng.random() < 0.6 else 0
                pf = w * nch + int(lrng.choice([0, 1, 2, 3, 5, 16, 29, 64]))
                fb = lrng.integers(0, 256, size=off + h * pf + 16, dtype=np.uint8)
                fb[off:off + h * pf].reshape(h, pf)[:, :w * nch] = b[:, :w * nch]
                flat.append(fb); ptrs.append((off, pf))
            ts = [torch.from_numpy(fb).cuda() for fb in flat]
            ptrs = [(t.data_ptr() + off, pf) for t, (off, pf) in zip(ts, ptrs)]
        else:
            ts = [torch.from_numpy(b).cuda() for b in bufs]
        gs = det.detect_batch_ex(ptrs if ptrs else [(t.data_ptr(), pitch) for t in ts], max_dets=MAX_DETS, encoding=enc)
        errs = []
        for f in range(batch):
            e, odets = pu.compare_stages(det, f, np.ascontiguousarray(imgs[f]), fams, K, dec, tag_size=tag_size, tile_size=tile, **more)
            e += pu.compare_detections(gs[f], odets[:MAX_DETS])
            errs += ["frame %d: %s" % (f, x) for x in e] if batch > 1 else e
        det.close()
        done += 1
        if errs:
            fails.append("case %d FAIL kind %s %dx%d pitch %d dec %d fams %s enc %s %s: %s" % (case, kind, w, h, pitch, dec, fams, enc,
                                                                                            more, errs[:3]))
            out(fails[-1])
    return done, fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--maxdim", type=int, default=420)
    ap.add_argument("--budget", type=float, default=1e9, help="seconds")
    ap.add_argument("--batch", type=int, default=1, help="frames per case (one submission)")
    ap.add_argument("--only", type=int, default=None, help="run this case only (the cases before it are generated and skipped)")
    ap.add_argument("--dump", default=None, help="with --only: write the case's frames to this .npz")
    ap.add_argument("--tile", type=int, default=4, help="tile_size of the handle (4 or 8)")
    ap.add_argument("--colour", action="store_true", help="submit the content as rgb8 / bgr8 / rgba8 / bgra8 frames with random chroma")
    ap.add_argument("--params", action="store_true", help="random decode parameters (refine_edges, max_hamming 0..3, decode_sharpening, skew, tag_size)")
    ap.add_argument("--layout", action="store_true", help="every frame at a base address (any byte) and a pitch of its own")
    ap.add_argument("--path", default="alternate", help="launch set: latency | throughput | alternate | auto")
    a = ap.parse_args()
    t0 = time.time()
    done, fails = run_cases(a.cases, a.seed, a.maxdim, a.budget, out=lambda m: print(m, flush=True), path=None if a.path == "auto" else a.path, batch=a.batch,
                            only=a.only, dump=a.dump, colour=a.colour, tile=a.tile, params=a.params, layout=a.layout)
    print("fuzz: %d cases, %d failed, %.1f s" % (done, len(fails), time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
