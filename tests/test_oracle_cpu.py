"""CPU tests of the oracle (oracle/apriltag_oracle.c): it is pinned against the reference's own golden
vector (isaac_ros_apriltag/test/isaac_ros_apriltag_pol_test.py:113-175, re-rendered frame), against the
renderer's analytic ground truth, and stage by stage against independent numpy/pure-Python
restatements on small inputs."""
import math

import numpy as np
import pytest

from isaac_ros_apriltag_amd import synth
from oracle import pyoracle as po


def _params(K, **kw):
    return po.default_params(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], **kw)


def _quat_wxyz(R):
    # same construction the node uses (Eigen quaternion from a rotation matrix), normalised
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[1 + i] = 0.25 * s
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def test_pol_golden_vector(built):
    """Reference golden numbers, reference tolerances (2 px, 0.01 m, 0.01)."""
    img, K, truth = synth.scene_pol_golden()
    dets, _ = po.detect(img, params=_params(K))
    assert len(dets) >= 1
    for d in dets:
        assert d["id"] == 0 and d["family"] == "tag36h11"
        corners = d["p"][::-1]  # message order: corners[i] = p[3-i]  (apriltag_node.cpp:337-344,512-517)
        gold = [(1044.0, 665.0), (808.0, 665.0), (808.0, 429.0), (1044.0, 429.0)]
        for c, g in zip(corners, gold):
            assert abs(c[0] - g[0]) <= 2 and abs(c[1] - g[1]) <= 2
        assert abs(d["center"][0] - 926.0) <= 2 and abs(d["center"][1] - 547.0) <= 2
        for v, g in zip(d["t"], (0.255342, 0.098358, 0.403961)):
            assert abs(v - g) <= 0.01
        q = _quat_wxyz(d["R"])
        if q[3] < 0:
            q = -q
        for v, g in zip(q, (0.0, 0.0, 0.0, 1.0)):
            assert abs(v - g) <= 0.01


def test_mono8_at_least_one_detection(built):
    img, K, _ = synth.scene_pol_golden()
    dets, _ = po.detect(img, params=_params(K))
    assert len(dets) >= 1


@pytest.mark.parametrize("decimate", [1, 2])
def test_c1_truth(built, decimate):
    img, K, truth = synth.scene_c1()
    dets, _ = po.detect(img, params=_params(K, decimate=decimate))
    assert [d["id"] for d in dets] == [0]
    assert np.abs(dets[0]["p"] - truth[0]["p"]).max() < 0.5
    # upright fronto-parallel tag: identity orientation
    assert np.abs(dets[0]["R"] - np.eye(3)).max() < 0.02


@pytest.mark.parametrize("sigma", [0.0, 2.0])
def test_c2_truth(built, sigma):
    img, K, truth = synth.scene_c2(seed=1234, sigma=sigma)
    dets, _ = po.detect(img, params=_params(K))
    assert [d["id"] for d in dets] == list(range(10))
    tm = {t["id"]: t for t in truth}
    for d in dets:
        assert d["hamming"] == 0
        assert np.abs(d["p"] - tm[d["id"]]["p"]).max() < 0.5
        assert np.abs(d["t"] - tm[d["id"]]["t"]).max() < 0.05


def test_c5_two_families(built):
    img, K, truth = synth.scene_c5()
    dets, _ = po.detect(img, families=("tag36h11", "tag25h9"), params=_params(K))
    assert sorted((d["family"], d["id"]) for d in dets) == sorted((t["family"], t["id"]) for t in truth)


@pytest.mark.parametrize("quarter", [0, 1, 2, 3])
def test_rotation_corner_order(built, quarter):
    ang = quarter * math.pi / 2
    c, s = math.cos(ang), math.sin(ang)
    H = np.array([[60 * c, -60 * s, 320.0], [60 * s, 60 * c, 240.0], [0, 0, 1.0]])
    img = synth.render(640, 480, [{"family": "tag36h11", "id": 3, "H": H}], background=160)
    dets, _ = po.detect(img, params=_params(synth.default_K(640, 480)))
    assert [d["id"] for d in dets] == [3]
    truth = synth.truth_from_H("tag36h11", 3, H)
    assert np.abs(dets[0]["p"] - truth["p"]).max() < 0.5


@pytest.mark.parametrize("nflip", [1, 2, 3])
def test_hamming_correction(built, nflip):
    codes, d = po.family_codes("tag36h11")
    code = codes[5]
    for b in (3, 17, 30)[:nflip]:
        code ^= 1 << b
    H = np.array([[70.0, 0, 300.0], [0, 70.0, 220.0], [0, 0, 1.0]])
    img = synth.render(640, 480, [{"family": "tag36h11", "id": 5, "H": H, "code": code}], background=160)
    dets, _ = po.detect(img, params=_params(synth.default_K(640, 480)))
    if nflip <= 2:
        assert [(x["id"], x["hamming"]) for x in dets] == [(5, nflip)]
    else:
        assert dets == []


def test_empty_and_degenerate_inputs(built):
    flat = np.full((64, 64), 128, dtype=np.uint8)
    dets, dump = po.detect(flat, want_dump=True)
    assert dets == [] and (dump["thr"] == 127).all()
    tiny = np.zeros((4, 4), dtype=np.uint8)
    assert po.detect(tiny)[0] == []


def _numpy_threshold(im, tile=4, min_diff=5):
    h, w = im.shape
    tw, th = w // tile, h // tile
    t = im[:th * tile, :tw * tile].reshape(th, tile, tw, tile)
    tmin, tmax = t.min(axis=(1, 3)).astype(int), t.max(axis=(1, 3)).astype(int)
    pmin = np.pad(tmin, 1, constant_values=255)
    pmax = np.pad(tmax, 1, constant_values=0)
    dmin = np.min([pmin[1 + dy:1 + dy + th, 1 + dx:1 + dx + tw] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0)
    dmax = np.max([pmax[1 + dy:1 + dy + th, 1 + dx:1 + dx + tw] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0)
    ty = np.minimum(np.arange(h) // tile, th - 1)
    tx = np.minimum(np.arange(w) // tile, tw - 1)
    mn, mx = dmin[ty][:, tx], dmax[ty][:, tx]
    out = np.where(im.astype(int) > mn + (mx - mn) // 2, 255, 0).astype(np.uint8)
    low = (mx - mn) < min_diff
    low[th * tile:, :] = False
    low[:, tw * tile:] = False
    out[low] = 127
    return out


@pytest.mark.parametrize("shape", [(32, 48), (37, 50), (16, 16), (5, 9)])
def test_threshold_vs_numpy(built, shape):
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    im = rng.integers(0, 256, size=shape, dtype=np.uint8)
    im[: shape[0] // 2] = (im[: shape[0] // 2] // 64) + 100   # low-contrast half
    assert np.array_equal(po.threshold(im), _numpy_threshold(im))


def _bruteforce_cc(thr):
    h, w = thr.shape
    parent = list(range(w * h))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    def union(a, b):
        a, b = find(a), find(b)
        if a != b:
            parent[max(a, b)] = min(a, b)
    for y in range(h):
        for x in range(1, w - 1):
            v = thr[y, x]
            if v == 127:
                continue
            if thr[y, x - 1] == v:
                union(y * w + x, y * w + x - 1)
            if y > 0:
                if thr[y - 1, x] == v:
                    union(y * w + x, (y - 1) * w + x)
                if v == 255:
                    if thr[y - 1, x - 1] == v:
                        union(y * w + x, (y - 1) * w + x - 1)
                    if thr[y - 1, x + 1] == v:
                        union(y * w + x, (y - 1) * w + x + 1)
    lab = np.array([find(i) if thr.flat[i] != 127 else 0xFFFFFFFF for i in range(w * h)], dtype=np.uint32)
    return lab.reshape(h, w)


def test_cc_vs_bruteforce(built):
    rng = np.random.default_rng(7)
    thr = rng.choice(np.array([0, 127, 255], dtype=np.uint8), size=(40, 70), p=[0.45, 0.1, 0.45])
    label, csize = po.connected_components(thr)
    ref = _bruteforce_cc(thr)
    assert np.array_equal(label, ref)
    for r in np.unique(ref[ref != 0xFFFFFFFF]):
        assert csize.flat[r] == (ref == r).sum()


def test_decimate_point_sampling(built):
    rng = np.random.default_rng(3)
    im = rng.integers(0, 256, size=(31, 45), dtype=np.uint8)
    for f in (1, 2, 3):
        assert np.array_equal(po.decimate(im, f), im[::f, ::f])


def test_pose_matches_truth_homography(built):
    """Pose from the exact homography of a known pose reproduces it (north_star tolerance 1e-4)."""
    K = synth.default_K(1920, 1080)
    R = synth.rot_xyz(0.2, -0.3, 0.4)
    t = np.array([0.1, -0.05, 1.2])
    H = synth.homography_from_pose(R, t, K, 0.22)
    R2, t2 = po.pose_from_homography(H, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.22)
    assert np.abs(R2 - R).max() < 1e-4 and np.abs(t2 - t).max() < 1e-4


def test_front_steps_resize_and_rectify(built):
    """Oracle definitions of the front steps: identity cases are exact, a 4K board resized to 1080p is
    still fully detected, and rectifying with zero distortion onto the same camera is the identity."""
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    assert np.array_equal(po.resize_mono8(a, 53, 37), a)
    assert (po.resize_mono8(np.full((40, 60), 99, dtype=np.uint8), 17, 23) == 99).all()
    img, K, truth, size = synth.scene_c3()
    small = po.resize_mono8(img, 1920, 1080)
    dets, _ = po.detect(small, families=("tag36h11",), params=po.default_params(fx=2000, fy=2000, cx=960, cy=540, tag_size=size))
    assert sorted(d["id"] for d in dets) == list(range(100))
    img1, K1, _ = synth.scene_c1()
    assert np.array_equal(po.rectify_mono8(img1, K1, [0, 0, 0, 0, 0], K1), img1)
    warped = po.rectify_mono8(img1, K1, [-0.25, 0.07, 0.001, -0.002, 0.0], K1)
    assert (warped != img1).mean() > 0.001 and [d["id"] for d in po.detect(warped, params=_params(K1))[0]] == [0]


def test_random_scene_sweep_vs_ground_truth(built):
    """The oracle against the renderer's ground truth over 40 random VGA scenes: every error-free
    (hamming 0) tag36h11/tag25h9 detection is a rendered tag with corners within 1 px (2-bit-corrected
    false positives from noise blobs are a known property of the small families and are tolerated), and
    at least 85 % of the rendered tags are found."""
    rng = np.random.default_rng(7)
    K = synth.default_K(640, 480)
    found = expected = 0
    for case in range(40):
        fam = ("tag36h11", "tag25h9")[case % 2]
        ncodes = {"tag36h11": 27, "tag25h9": 35}[fam]
        tags, truth = [], []
        centers = [(170.0, 130.0), (470.0, 130.0), (170.0, 350.0), (470.0, 350.0)]
        for t in range(int(rng.integers(1, 5))):
            side = float(rng.uniform(50, 100))
            cx, cy = centers[t][0] + float(rng.uniform(-25, 25)), centers[t][1] + float(rng.uniform(-20, 20))
            R = synth.rot_xyz(float(rng.uniform(-0.45, 0.45)), float(rng.uniform(-0.45, 0.45)), float(rng.uniform(-3.1, 3.1)))
            z = K[0, 0] * 0.1 / side
            tvec = np.array([(cx - 320) / K[0, 0] * z, (cy - 240) / K[1, 1] * z, z])
            H = synth.homography_from_pose(R, tvec, K, 0.1)
            tid = int(rng.integers(0, ncodes))
            tags.append({"family": fam, "id": tid, "H": H})
            truth.append(synth.truth_from_H(fam, tid, H, R, tvec))
        sigma = float(rng.choice([0.0, 1.0, 2.0, 4.0]))
        img = synth.render(640, 480, tags, background=int(rng.integers(100, 190)), sigma=sigma, seed=500 + case)
        dets, _ = po.detect(img, families=(fam,), params=_params(K, tag_size=0.1))
        expected += len(truth)
        for d in dets:
            match = [t for t in truth if t["id"] == d["id"] and np.abs(t["p"] - d["p"]).max() < 1.0]
            if not match and d["hamming"] > 0:
                continue
            assert match, (case, d["id"], d["hamming"])
            assert np.abs(match[0]["t"] - d["t"]).max() < 0.05 * match[0]["t"][2] + 0.01
            found += 1
    assert found >= 0.85 * expected, (found, expected)
