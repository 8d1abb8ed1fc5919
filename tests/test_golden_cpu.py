"""The oracle reproduces the committed golden vectors (tests/golden/*.json, made by tools/make_golden.py)."""
import glob
import json
import os
import zlib

import numpy as np
import pytest

from isaac_ros_apriltag_amd import synth
from oracle import pyoracle as po
import parity_util as pu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


def load_scene(rec):
    r = getattr(synth, rec["scene"])(**rec["kwargs"])
    return r[0], r[1]


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-5] for p in GOLD])
def test_oracle_matches_golden(built, path):
    rec = json.load(open(path))
    img, K = load_scene(rec)
    assert zlib.crc32(img.tobytes()) == rec["image_crc32"], "renderer bytes changed"
    dets, dump = po.detect(img, families=tuple(rec["families"]), params=pu.oracle_params(K, rec["decimate"]), want_dump=True)
    assert zlib.crc32(dump["thr"].tobytes()) == rec["thr_crc32"]
    assert zlib.crc32(dump["label"].tobytes()) == rec["label_crc32"]
    assert zlib.crc32(dump["points"].tobytes()) == rec["points_crc32"]
    assert [np.asarray(q["p"], dtype="<f4").tobytes().hex() for q in dump["quads"]] == rec["quads_hex"]
    assert len(dets) == len(rec["detections"])
    for d, g in zip(dets, rec["detections"]):
        assert (d["family"], d["id"], d["hamming"]) == (g["family"], g["id"], g["hamming"])
        assert np.asarray(d["p"], dtype="<f8").tobytes().hex() == g["p_hex"]
        assert np.asarray(d["t"], dtype="<f8").tobytes().hex() == g["t_hex"]
