"""How far are the oracle's canonical definitions from AprilRobotics' own formulations?

The HIP path is checked bit for bit against the canonical oracle (oracle/apriltag_oracle.c, variant 0).  Wherever
upstream's statement is deterministic the oracle simply IS that statement (float corner differences, the float border
division, the float cos_critical_rad field, H * Rz with libm's cos / sin values).  What remains defined differently, on
purpose, is what upstream leaves to iteration order or to libm: exact cumulative moment sums
instead of sequential double additions in hash order, the exact integer border-direction dot instead of a float
accumulation in hash order, the eigenvector line normal instead of atan2f/cosf/sinf in the edge refinement, Newton steps
instead of the SVD for the polar factor of the pose, and the row-major order of the decision margin's float sums where
AprilTag 3 walks its quadrant bit order.  ATO_VAR_* switches each step to the upstream formulation; this file BOUNDS
what that changes, on configs 1, 2, 3 and 5:

  ids, hamming, detection count        identical
  decision margin                      within 1e-4 relative (float sums in another order)
  corners / centre                     within 2.5e-4 px (measured: 6.1e-5 px at 1080p, 1.2e-4 px at 4K, all of it
                                       from the float atan2f/cosf/sinf of the edge normal), i.e. identical after
                                       rounding to 1e-3 px except for a value that sits within 2.5e-4 px of a
                                       rounding boundary  (north_star: "IDs/corners bit-identical after rounding")
  rotation entries / translation       within 1e-4  (north_star: "pose within 1e-4 of the reference
                                       homography solve")
"""
import numpy as np
import pytest

import parity_util as pu
from isaac_ros_apriltag_amd import synth
from oracle import pyoracle as po

ALL = po.VAR_SEQ_MOMENTS | po.VAR_ATAN_NORMAL | po.VAR_SVD_POLAR | po.VAR_FLOAT_DOT | po.VAR_AT3_BIT_ORDER
MARGIN_TOL = 1e-4    # decision margins are floats around 30..120: a few ulp of a float sum taken in another order
ROUND_PX = 1e-3
CORNER_TOL = 0.25 * ROUND_PX
POSE_TOL = 1e-4


def _scenes():
    yield "c1_dec2", synth.scene_c1(), ("tag36h11",), 2
    for seed in (1234, 1240):
        yield "c2_sigma2_seed%d" % seed, synth.scene_c2(seed=seed, sigma=2.0), ("tag36h11",), 1
    yield "c2_sigma0", synth.scene_c2(seed=1234, sigma=0.0), ("tag36h11",), 1
    yield "c2_sigma2_dec2", synth.scene_c2(seed=1236, sigma=2.0), ("tag36h11",), 2
    yield "c5_two_families", synth.scene_c5(seed=4321, sigma=2.0), ("tag36h11", "tag25h9"), 1


def _run(img, K, fams, dec, variant):
    prm = pu.oracle_params(K, dec, 0.22)
    prm.variant = variant
    return po.detect(img, families=fams, params=prm)[0]


def _compare(a, b):
    assert [(d["family"], d["id"], d["hamming"]) for d in a] == [(d["family"], d["id"], d["hamming"]) for d in b]
    dc = dr = dt = 0.0
    for x, y in zip(a, b):
        dc = max(dc, float(np.abs(x["p"] - y["p"]).max()), float(np.abs(x["center"] - y["center"]).max()))
        dr = max(dr, float(np.abs(x["R"] - y["R"]).max()))
        dt = max(dt, float(np.abs(x["t"] - y["t"]).max()))
        assert abs(x["decision_margin"] - y["decision_margin"]) <= MARGIN_TOL * max(1.0, abs(x["decision_margin"]))
    return dc, dr, dt


@pytest.mark.parametrize("variant,name", [(po.VAR_SEQ_MOMENTS, "sequential moment sums"),
                                          (po.VAR_ATAN_NORMAL, "atan2f normal"),
                                          (po.VAR_SVD_POLAR, "SVD polar factor"),
                                          (po.VAR_FLOAT_DOT, "float border dot"),
                                          (po.VAR_AT3_BIT_ORDER, "AprilTag 3 bit order of the score sums"),
                                          (ALL, "all upstream formulations")])
def test_upstream_formulations_within_rounding(built, variant, name):
    worst = [0.0, 0.0, 0.0]
    ntags = 0
    for sname, (img, K, truth), fams, dec in _scenes():
        a = _run(img, K, fams, dec, 0)
        b = _run(img, K, fams, dec, variant)
        assert len(a) == len(truth), sname
        dc, dr, dt = _compare(a, b)
        ntags += len(a)
        worst = [max(w, v) for w, v in zip(worst, (dc, dr, dt))]
        # bound the raw difference well below the rounding unit (two values can still straddle a boundary)
        assert dc < CORNER_TOL, (sname, name, dc)
        assert dr < POSE_TOL and dt < POSE_TOL, (sname, name, dr, dt)
    print("\\n%-28s %3d tags: max |d corner| %.3g px, max |d R| %.3g, max |d t| %.3g m" % (name, ntags, *worst))


def test_c3_board_all_upstream_formulations(built):
    """Config 3 (3840x2160, 100-tag board, decimate 2) with every upstream formulation at once."""
    img, K, truth, size = synth.scene_c3(seed=77, sigma=2.0)

    def run(variant):
        prm = pu.oracle_params(K, 2, size)
        prm.variant = variant
        return po.detect(img, families=("tag36h11",), params=prm)[0]
    a, b = run(0), run(ALL)
    assert len(a) == len(truth) == 100
    dc, dr, dt = _compare(a, b)
    print("\\nc3: max |d corner| %.3g px, |d R| %.3g, |d t| %.3g" % (dc, dr, dt))
    assert dc < CORNER_TOL and dr < POSE_TOL and dt < POSE_TOL


def test_skew_pose_recovers_truth(built):
    """The VPI path of the reference passes K[1] (apriltag_node.cpp:215-225): with a skewed camera matrix the
    pose solve must reproduce the rotation and translation that generated the homography."""
    rng = np.random.default_rng(5)
    fx, fy, cx, cy, sk, size = 1100.0, 1050.0, 900.0, 500.0, 7.5, 0.22
    K = np.array([[fx, sk, cx], [0, fy, cy], [0, 0, 1.0]])
    for _ in range(20):
        R = synth.rot_xyz(*rng.uniform(-0.5, 0.5, size=3))
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(0.8, 2.0)])
        H = synth.homography_from_pose(R, t, K, size)
        R2, t2 = po.pose_from_homography(H, fx, fy, cx, cy, size, skew=sk)
        R0, t0 = po.pose_from_homography(H, fx, fy, cx, cy, size, skew=0.0)
        assert np.abs(R2 - R).max() < 1e-4 and np.abs(t2 - t).max() < 1e-4
        assert max(np.abs(R0 - R).max(), np.abs(t0 - t).max()) > 1e-4   # ignoring the skew is measurably wrong


def test_fast_paths_change_nothing():
    """ATO_VAR_FAST_PATHS (bench.py's second CPU row) is cheaper CODE, not another definition: radix-sorted slope keys and the
    quick_decode table give the same quads and detections bit for bit, on clean and noisy frames, two families, and tags whose
    code words carry one or two flipped bits (the table's Hamming entries); with ATO_VAR_SEQ_MOMENTS beside it the result is the
    SEQ_MOMENTS variant's, bit for bit."""
    cases = list(_scenes())
    codes, _ = synth.family_codes("tag36h11")
    for nflip in (1, 2, 3):   # (three flipped bits: beyond what is corrected -- no detection either way)
        tags = []
        for k in range(6):
            code = codes[40 + k]
            for b in range(nflip):
                code ^= 1 << ((7 * k + 13 * b + 3) % 36)
            H = np.array([[60.0, 0, 110.0 + 190.0 * (k % 3)], [0, 60.0, 130.0 + 210.0 * (k // 3)], [0, 0, 1.0]])
            tags.append({"family": "tag36h11", "id": 40 + k, "H": H, "code": code})
        img = synth.render(640, 480, tags, background=160, sigma=1.0, seed=77 + nflip)
        cases.append(("vga_%d_flipped_bits" % nflip, (img, synth.default_K(640, 480)), ("tag36h11",), 1))
    for name, r, fams, dec in cases:
        img, K = r[0], r[1]
        for base in (0, po.VAR_SEQ_MOMENTS | po.VAR_FLOAT_DOT):
            prm = pu.oracle_params(K, dec, 0.22)
            prm.variant = base
            a, da = po.detect(img, families=fams, params=prm, want_dump=True)
            prm.variant = base | po.VAR_FAST_PATHS
            b, db = po.detect(img, families=fams, params=prm, want_dump=True)
            assert not pu.compare_detections(b, a), (name, base)
            assert pu.stage_digest_oracle(da) == pu.stage_digest_oracle(db), (name, base)
