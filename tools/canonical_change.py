#!/usr/bin/env python3
"""What did making upstream's float-typed statements canonical change?  Builds the oracle of an earlier commit (default: the
round-3 head 216217b) from git history into /tmp and compares its detections with the current oracle's on configs 1, 2, 3
and 5: detection sets, quad counts, max |d corner|, |d R|, |d t|, |d margin|.  The numbers go into DESIGN.md section 2.
Usage: python tools/canonical_change.py [commit]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402
from isaac_ros_apriltag_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def old_lib(commit):
    d = tempfile.mkdtemp(prefix="old_oracle_")
    os.makedirs(os.path.join(d, "oracle")); os.makedirs(os.path.join(d, "include"))
    for f in ("oracle/apriltag_oracle.c", "oracle/apriltag_oracle.h", "include/apriltag_amd_families.h"):
        open(os.path.join(d, f), "w").write(subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (commit, f)], text=True))
    so = os.path.join(d, "old.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu99", "-ffp-contract=off", "-o", so,
                           os.path.join(d, "oracle", "apriltag_oracle.c"), "-lm"])
    return so


def scenes():
    yield "c1 dec2", synth.scene_c1()[:2], ("tag36h11",), 2, 0.22
    for seed in range(1234, 1242):
        yield "c2 sigma2 %d" % seed, synth.scene_c2(seed=seed, sigma=2.0)[:2], ("tag36h11",), 1, 0.22
    yield "c2 sigma0", synth.scene_c2(seed=1234, sigma=0.0)[:2], ("tag36h11",), 1, 0.22
    yield "c2 sigma2 dec2", synth.scene_c2(seed=1236, sigma=2.0)[:2], ("tag36h11",), 2, 0.22
    for seed in (4321, 4322, 4323, 4324):
        yield "c5 %d" % seed, synth.scene_c5(seed=seed, sigma=2.0)[:2], ("tag36h11", "tag25h9"), 1, 0.22
    r = synth.scene_c3(seed=77, sigma=2.0)
    yield "c3", (r[0], r[1]), ("tag36h11",), 2, r[3]


def main():
    commit = sys.argv[1] if len(sys.argv) > 1 else "216217b"
    so = old_lib(commit)
    new = po.lib()
    rows = []
    for name, (img, K), fams, dec, size in scenes():
        res = []
        for which in ("new", "old"):
            if which == "old":
                po._lib = None
                po._LIB_PATH, keep = so, po._LIB_PATH
            dets, dump = po.detect(img, families=fams, params=pu.oracle_params(K, dec, size), want_dump=True)
            nq = len(dump["quads"])
            if which == "old":
                po._lib = None
                po._LIB_PATH = keep
            res.append((dets, nq))
        (a, nqa), (b, nqb) = res
        ka = [(d["family"], d["id"], d["hamming"]) for d in a]
        kb = [(d["family"], d["id"], d["hamming"]) for d in b]
        dc = dr = dt = dm = 0.0
        if ka == kb:
            for x, y in zip(a, b):
                dc = max(dc, float(np.abs(x["p"] - y["p"]).max()), float(np.abs(x["center"] - y["center"]).max()))
                dr = max(dr, float(np.abs(x["R"] - y["R"]).max()))
                dt = max(dt, float(np.abs(x["t"] - y["t"]).max()))
                dm = max(dm, abs(x["decision_margin"] - y["decision_margin"]) / max(1.0, abs(x["decision_margin"])))
        rows.append((name, len(a), ka == kb, nqa, nqb, dc, dr, dt, dm))
        print("%-18s dets %3d same-ids %s quads %4d / %4d  d corner %.3g px  d R %.3g  d t %.3g m  d margin %.3g" % rows[-1])
    print("worst: d corner %.3g px, d R %.3g, d t %.3g m, d margin %.3g; quad-count changes: %d scenes" %
          (max(r[5] for r in rows), max(r[6] for r in rows), max(r[7] for r in rows), max(r[8] for r in rows),
           sum(1 for r in rows if r[3] != r[4])))


if __name__ == "__main__":
    main()
