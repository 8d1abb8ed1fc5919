"""Generates tests/golden/*.json: inputs (scene recipe) and expected outputs (CPU-oracle detections,
stage counts, and a CRC of every integer stage) for a few scenes.  The GPU parity tests check the HIP
path against these committed vectors as well as against the live oracle.
Run from the repo root:  python tools/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from isaac_ros_apriltag_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
import parity_util as pu  # noqa: E402

SCENES = {
    "c1_dec2": dict(scene="scene_c1", kwargs={}, families=["tag36h11"], decimate=2),
    "pol_golden": dict(scene="scene_pol_golden", kwargs={}, families=["tag36h11"], decimate=1),
    "c2_seed1234_sigma2": dict(scene="scene_c2", kwargs={"seed": 1234, "sigma": 2.0}, families=["tag36h11"], decimate=1),
    "c2_seed1240_sigma2_dec2": dict(scene="scene_c2", kwargs={"seed": 1240, "sigma": 2.0}, families=["tag36h11"], decimate=2),
    "c2_high_ids_sigma2": dict(scene="scene_c2_ids", kwargs={"ids": [100, 137, 211, 298, 333, 402, 467, 511, 560, 586], "seed": 1301, "sigma": 2.0},
                               families=["tag36h11"], decimate=1),
    "c5_two_families": dict(scene="scene_c5", kwargs={"seed": 1234, "sigma": 2.0}, families=["tag36h11", "tag25h9"], decimate=1),
}


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, spec in SCENES.items():
        r = getattr(synth, spec["scene"])(**spec["kwargs"])
        img, K = r[0], r[1]
        dets, dump = po.detect(img, families=tuple(spec["families"]), params=pu.oracle_params(K, spec["decimate"]), want_dump=True)
        rec = dict(spec)
        rec["image_crc32"] = zlib.crc32(img.tobytes())
        rec["K"] = [float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])]
        rec["thr_crc32"] = zlib.crc32(dump["thr"].tobytes())
        rec["label_crc32"] = zlib.crc32(dump["label"].tobytes())
        rec["nclusters"] = len(dump["clusters"])
        rec["npoints"] = int(len(dump["points"]))
        rec["points_crc32"] = zlib.crc32(dump["points"].tobytes())
        rec["nquads"] = len(dump["quads"])
        rec["quads_hex"] = [np.asarray(q["p"], dtype="<f4").tobytes().hex() for q in dump["quads"]]
        rec["detections"] = [{"family": d["family"], "id": d["id"], "hamming": d["hamming"],
                              "decision_margin_hex": np.float32(d["decision_margin"]).tobytes().hex(),
                              "p_hex": np.asarray(d["p"], dtype="<f8").tobytes().hex(),
                              "center_hex": np.asarray(d["center"], dtype="<f8").tobytes().hex(),
                              "R_hex": np.asarray(d["R"], dtype="<f8").tobytes().hex(),
                              "t_hex": np.asarray(d["t"], dtype="<f8").tobytes().hex(),
                              "p": np.round(d["p"], 4).tolist(), "t": np.round(d["t"], 6).tolist()} for d in dets]
        json.dump(rec, open(os.path.join(outdir, name + ".json"), "w"), indent=1)
        print(name, "dets", len(dets), "quads", rec["nquads"], "points", rec["npoints"])


if __name__ == "__main__":
    main()
