#!/usr/bin/env python3
"""Rejection power of the partition bound (oracle/apriltag_oracle.c, ato_diag_log) on config-2 frames: of the
clusters that reach the moment prefixes, which exit of fit_quad do they take, and how many would a test
'ratio[k] > 1' have proven hopeless right after the first walk of the moment sweep?  Also checks soundness on the
sample: no cluster that found an admissible corner choice (exits 5..10) may have a ratio above 1."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402
from isaac_ros_apriltag_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


class Rec(C.Structure):
    _fields_ = [("reason", C.c_int), ("points", C.c_int), ("ratio", C.c_double * 8)]


def main():
    sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    dec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lib = po.lib()
    cap = 200000
    buf = (Rec * cap)()
    C.c_void_p.in_dll(lib, "ato_diag_log").value = C.addressof(buf)
    C.c_int.in_dll(lib, "ato_diag_n").value = 0
    C.c_int.in_dll(lib, "ato_diag_cap").value = cap
    for f in range(nframes):
        img, K, _ = synth.scene_c2(seed=1234 + f, sigma=sigma)
        po.detect(img, params=pu.oracle_params(K, dec, 0.22))
    n = C.c_int.in_dll(lib, "ato_diag_n").value
    C.c_int.in_dll(lib, "ato_diag_cap").value = 0
    reason = np.array([buf[i].reason for i in range(n)])
    pts = np.array([buf[i].points for i in range(n)])
    ratio = np.array([list(buf[i].ratio) for i in range(n)])
    print("%d clusters, %d points reach the prefixes (per frame %.0f / %.0f)" % (n, pts.sum(), n / nframes, pts.sum() / nframes))
    doomed = (reason == 3) | (reason == 4)
    good = reason >= 5
    print("doomed (exits 3, 4): %d clusters %d points; admissible (exits 5..10): %d clusters %d points" %
          (doomed.sum(), pts[doomed].sum(), good.sum(), pts[good].sum()))
    print("max ratio over admissible clusters, per level:", np.round(ratio[good].max(axis=0), 4))
    for k in range(8):
        rej = ratio[:, k] > 1.0
        print("level %d: rejects %5d clusters %8d points (%.1f %% of doomed points); unsound rejections %d" %
              (k, (rej & doomed).sum(), pts[rej & doomed].sum(), 100.0 * pts[rej & doomed].sum() / max(1, pts[doomed].sum()), (rej & good).sum()))
    best = ratio.max(axis=1) > 1.0
    print("any level: rejects %d clusters %d points (%.1f %% of doomed points), unsound %d" %
          ((best & doomed).sum(), pts[best & doomed].sum(), 100.0 * pts[best & doomed].sum() / max(1, pts[doomed].sum()), (best & good).sum()))
    names = ["sect32", "sect64", "sect128", "grp48", "grp64", "grp96", "grp128", "grp256"]
    for lo, hi in ((24, 768), (769, 2048), (2049, 4096), (4097, 8192), (8193, 1 << 30)):
        m = (pts >= lo) & (pts <= hi)
        print("  class %5d..%-6d: %6d points doomed, %6d rejected; %6d points admissible" %
              (lo, min(hi, 99999), pts[m & doomed].sum(), pts[m & doomed & best].sum(), pts[m & good].sum()))
        print("       per test: " + ", ".join("%s %d" % (names[k], pts[m & doomed & (ratio[:, k] > 1.0)].sum()) for k in range(8)))


if __name__ == "__main__":
    main()
