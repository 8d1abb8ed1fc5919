#!/usr/bin/env python3
"""Where do the clusters of a frame die?  Runs the CPU restatement on config-2 frames and prints, per exit of
fit_quad, the number of clusters and points (oracle/apriltag_oracle.c ato_stats), plus the size histogram of
the kept clusters.  Diagnostic for the quad-fit kernel's work distribution."""
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

REASONS = ["bbox", "border direction", "<24 after dedup", "<4 maxima", "no admissible corners", "total error",
           "final line mse", "degenerate intersection", "area", "angles/winding", "accepted"]


def main():
    sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    dec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lib = po.lib()
    stats = (C.c_longlong * 32).in_dll(lib, "ato_stats")
    for i in range(32):
        stats[i] = 0
    hist = np.zeros(16, dtype=np.int64)
    hpts = np.zeros(16, dtype=np.int64)
    ncl = 0
    for f in range(nframes):
        img, K, _ = synth.scene_c2(seed=1234 + f, sigma=sigma)
        dets, dump = po.detect(img, params=pu.oracle_params(K, dec, 0.22), want_dump=True)
        cnt = np.array([c[2] for c in dump["clusters"]])
        ncl += len(cnt)
        b = np.minimum(15, np.floor(np.log2(np.maximum(cnt, 1))).astype(int))
        hist += np.bincount(b, minlength=16)
        hpts += np.bincount(b, weights=cnt, minlength=16).astype(np.int64)
    print("frames %d sigma %g decimate %d: %.0f clusters, %.0f points per frame" % (nframes, sigma, dec, ncl / nframes, hpts.sum() / nframes))
    print("%-26s %10s %12s" % ("exit", "clusters/f", "points/f"))
    for r, name in enumerate(REASONS):
        print("%-26s %10.1f %12.1f" % (name, stats[r] / nframes, stats[16 + r] / nframes))
    print("size histogram (per frame): [2^k, 2^(k+1))")
    for k in range(4, 16):
        print("  %6d.. %8.1f clusters %10.1f points" % (1 << k, hist[k] / nframes, hpts[k] / nframes))


if __name__ == "__main__":
    main()
