"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded frames -- bit-exact on every integer stage (threshold, labels,
component sizes, clusters, points), bit-exact floats for quads, and bit-exact ids / corners /
homographies / poses for the detections -- plus the committed golden vectors and size-independent
properties at the full BASELINE.json sizes."""
import ctypes as C
import glob
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from isaac_ros_apriltag_amd import capi, synth  # noqa: E402
from isaac_ros_apriltag_amd.detector import AprilTagDetector  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
import parity_util as pu
from isaac_ros_apriltag_amd import capi as capi_mod  # noqa: E402

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


@pytest.fixture(autouse=True)
def _registry_left_clean():
    """The family registry is process-wide: whatever a test registers (slots 3..8) is gone when the next one starts."""
    yield
    for slot in range(capi.SLOT_TAG36H10, 9):
        capi.unregister_family(slot)


def _k4(K):
    return (K[0, 0], K[1, 1], K[0, 2], K[1, 2])


# The library has two launch sets (include/apriltag_amd_debug.h): submissions of up to eight 1080p frames' worth of pixels
# take the LATENCY set, larger ones the THROUGHPUT set (k_cc_local<4>, k_fit_small<2>, the per-wave prefilter, chunked select).
# The stage-level tests below pin each set in turn, so that both meet the oracle on every case whatever its size.
PATHS = ("latency", "throughput")


def _run(img, K, families=("tag36h11",), decimate=1, path=None, **kw):
    h, w = img.shape
    det = AprilTagDetector(w, h, families=families, decimate=decimate, intrinsics=_k4(K), max_batch=1, **kw)
    if path is not None:
        det.set_submission_path(path)
    g = det.detect_batch_ex(torch.from_numpy(np.ascontiguousarray(img)).cuda(), max_dets=256)[0]
    return det, g


def test_device_arithmetic_is_ieee(built):
    rng = np.random.default_rng(0)
    a = np.abs(rng.standard_normal(200000)) * 10 ** rng.uniform(-8, 10, 200000)
    b = rng.standard_normal(200000) * 10 ** rng.uniform(-4, 4, 200000)
    assert np.array_equal(capi.debug_math(0, a, b), np.sqrt(a))
    assert np.array_equal(capi.debug_math(1, a, b), a / b)
    af, bf = a.astype(np.float32), b.astype(np.float32)
    assert np.array_equal(capi.debug_math(2, a, b), np.sqrt(af).astype(np.float64))
    assert np.array_equal(capi.debug_math(3, a, b), (af / bf).astype(np.float64))
    # the line fit's shared-reciprocal division in its operand range: weights 1 <= W < 2^40 (plus mantissas
    # of all ones / a single one, where reciprocal rounding is hardest), numerators up to 2^70 of either sign
    W = np.concatenate([1 + rng.random(300000) * 10 ** rng.uniform(0, 12, 300000),
                        np.ldexp(2 - 2.0 ** -52, rng.integers(0, 40, 50000)), np.ldexp(1 + 2.0 ** -52, rng.integers(0, 40, 50000)),
                        np.ldexp(1.0, rng.integers(0, 40, 20000))])
    W = W[W < 2.0 ** 40]
    nmr = rng.standard_normal(W.size) * 10 ** rng.uniform(-3, 21, W.size)
    nmr[::7] = W[::7] * np.rint(rng.standard_normal(W[::7].size) * 1000)          # exact quotients
    nmr[3::11] = W[3::11] * (np.rint(rng.standard_normal(W[3::11].size) * 1e6) + 0.5) * 2.0 ** -30   # near ties
    assert np.array_equal(capi.debug_math(4, nmr, W), nmr / W)
    # the line-fit weights' integer square root (sqrt_u18), every possible argument
    g = np.arange(1 << 18, dtype=np.float64)
    assert np.array_equal(capi.debug_math(5, g, g), np.sqrt(g))


@pytest.mark.parametrize("name,scene,families,decimate", [
    ("c1", lambda: synth.scene_c1(), ("tag36h11",), 1),
    ("c1_dec2", lambda: synth.scene_c1(), ("tag36h11",), 2),
    ("c1_dec3", lambda: synth.scene_c1(), ("tag36h11",), 3),
    ("pol", lambda: synth.scene_pol_golden(), ("tag36h11",), 1),
    ("c2_clean", lambda: synth.scene_c2(sigma=0), ("tag36h11",), 1),
    ("c2", lambda: synth.scene_c2(), ("tag36h11",), 1),
    ("c2_dec2", lambda: synth.scene_c2(seed=1301), ("tag36h11",), 2),
    ("c5", lambda: synth.scene_c5(), ("tag36h11", "tag25h9"), 1),
])
@pytest.mark.parametrize("path", PATHS)
def test_stage_and_detection_parity(built, name, scene, families, decimate, path):
    r = scene()
    img, K = r[0], r[1]
    det, g = _run(img, K, families, decimate, path=path)
    assert det.last_submission_path() == path
    errs, odets = pu.compare_stages(det, 0, img, families, K, decimate)
    errs += pu.compare_detections(g, odets, exact=True)
    det.close()
    assert not errs, errs[:5]
    assert len(g) == len(odets)


def _content_frames(w=1920, h=1080):
    """Non-tag content at full size: uniform noise, two-level noise, stripes, a flat frame, an 11-px checkerboard (8 500 clusters of
    about 90 points -- every one a quad, all of them through the small-cluster fit of the throughput set), a big noisy-edged
    rectangle (clusters of thousands of points: the large size classes and their prefilter) and rings."""
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:h, 0:w]
    out = [rng.integers(0, 256, size=(h, w), dtype=np.uint8),
           ((rng.random((h, w)) < 0.45) * 255).astype(np.uint8),
           (((xx + 3) % 7 < 3) * 220 + 10).astype(np.uint8),
           np.full((h, w), 99, dtype=np.uint8),
           ((((yy // 11) + (xx // 11)) & 1) * 200 + 20).astype(np.uint8)]
    rect = np.full((h, w), 40.0)
    rect[h // 6: 5 * h // 6, w // 6: 5 * w // 6] = 220
    out.append(np.clip(np.rint(rect + rng.normal(0, 18.0, (h, w))), 0, 255).astype(np.uint8))
    r = np.sqrt((xx - w / 2.0) ** 2 + (yy - h / 2.0) ** 2)
    out.append((((r / 9.0).astype(np.int64) & 1) * 240 + 8).astype(np.uint8))
    return out


def _check_every_frame(det, res, frames, families, K, decimate=1):
    for i, img in enumerate(frames):
        errs, odets = pu.compare_stages(det, i, img, families, K, decimate)
        errs += pu.compare_detections(res[i], odets)
        assert not errs, (i, errs[:4])


def test_throughput_set_every_stage_of_every_frame(built):
    """FOURTEEN distinct 1080p frames in ONE call -- 14 x 2.07 Mpx is beyond the sixteen Mi pixels up
    to which a submission takes the latency set, so the library itself (path AUTO) picks what the bench runs: k_cc_local<4>,
    the normal work-list bucketing, k_fit_small<2> for clusters up to 128 points, k_fit_prefilter<64>, chunked select, copy
    commands.  Labels, component sizes, clusters, points, quads and detections of EVERY frame equal the oracle's."""
    frames = [synth.scene_c2(seed=1234 + 7 * i)[0] for i in range(6)] + [synth.scene_c2(seed=1300, sigma=0.0)[0]] + _content_frames()
    assert len(frames) == 14
    K = synth.default_K(1920, 1080)
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=len(frames))
    res = det.detect_batch_ex(torch.from_numpy(np.stack(frames)).cuda(), max_dets=64)
    assert det.last_submission_path() == "throughput"
    assert det.frame_flags(len(frames)) == [0] * len(frames)
    _check_every_frame(det, res, frames, ("tag36h11",), K)
    # the checkerboard's quads all come from clusters of at most 128 points: k_fit_small's output is under the comparison
    cl = det.debug(11, capi.DBG_CLUSTERS)
    nq = len(det.debug(11, capi.DBG_QUADS))
    assert nq > 8000 and np.percentile(cl["count"], 90) <= 128
    # ... and the same frames through the latency set give the same bytes (both equal the oracle's)
    det.set_submission_path("latency")
    res2 = det.detect_batch_ex(torch.from_numpy(np.stack(frames)).cuda(), max_dets=64)
    assert det.last_submission_path() == "latency"
    _check_every_frame(det, res2, frames, ("tag36h11",), K)
    det.close()


@pytest.mark.parametrize("families,decimate,scene", [
    (("tag36h11",), 2, lambda i: synth.scene_c2(seed=1301 + 5 * i)[0]),
    (("tag36h11", "tag25h9"), 1, lambda i: synth.scene_c5(seed=4321 + 3 * i)[0]),
])
def test_throughput_set_decimate2_and_two_families(built, families, decimate, scene):
    """The same for twelve frames at decimate 2 (pinned: twelve half-resolution working images are a small submission by size)
    and for config 5's two families at decimate 1 (picked by size)."""
    frames = [scene(i) for i in range(10)] + _content_frames()[:2]
    K = synth.default_K(1920, 1080)
    det = AprilTagDetector(1920, 1080, families=families, decimate=decimate, intrinsics=_k4(K), max_batch=len(frames))
    if decimate > 1:
        det.set_submission_path("throughput")
    res = det.detect_batch_ex(torch.from_numpy(np.stack(frames)).cuda(), max_dets=64)
    assert det.last_submission_path() == "throughput"
    _check_every_frame(det, res, frames, families, K, decimate)
    assert sum(len(r) for r in res[:10]) >= 90   # (ten tags per scene; a steeply tilted one may be missed by oracle and library alike)
    det.close()


@pytest.mark.parametrize("name,scene,decimate", [
    ("c1", lambda: synth.scene_c1(), 1),
    ("c1_dec2", lambda: synth.scene_c1(), 2),
    ("c2", lambda: synth.scene_c2(), 1),
    ("noise_ragged", lambda: (np.random.default_rng(8).integers(0, 256, size=(477, 635), dtype=np.uint8), synth.default_K(635, 477)), 1),
    ("noise_ragged_dec3", lambda: (np.random.default_rng(9).integers(0, 256, size=(203, 301), dtype=np.uint8), synth.default_K(301, 203)), 3),
])
@pytest.mark.parametrize("path", PATHS)
def test_tile_size_8(built, name, scene, decimate, path):
    """`tile_size` as the reference declares it (apriltag_node.cpp:566, handed to the library at :451): 8 constructs and detects,
    threshold image and every later stage bit-exact against the oracle run with tile 8 -- config 1, config 2, ragged noise (the
    pixels right of / below the last full 8 x 8 tile) -- on BOTH launch sets, and anything but 4 or 8 is still refused."""
    r = scene()
    img, K = np.ascontiguousarray(r[0]), r[1]
    h, w = img.shape
    det = AprilTagDetector(w, h, decimate=decimate, intrinsics=_k4(K), max_batch=1, tile_size=8)
    det.set_submission_path(path)
    g = det.detect_batch_ex(torch.from_numpy(img).cuda(), max_dets=64)[0]
    assert det.last_submission_path() == path
    errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, decimate, tile_size=8)
    errs += pu.compare_detections(g, odets)
    # the threshold does depend on the tile: the same frame with tile 4 gives another image
    thr8 = det.debug(0, capi.DBG_THRESH).copy()
    det.close()
    det4, _ = _run(img, K, decimate=decimate)
    thr4 = det4.debug(0, capi.DBG_THRESH)
    det4.close()
    assert not errs, errs[:4]
    assert not np.array_equal(thr4, thr8)
    if name.startswith("c"):
        assert len(g) >= 1
    for bad in (0, 2, 5, 16):
        with pytest.raises(capi.AprilTagsError) as e:
            AprilTagDetector(w, h, intrinsics=_k4(K), tile_size=bad)
        assert e.value.code == 2


def test_tile_size_8_through_the_node_shell(built):
    """tile_size:=8 on the node (the reference's parameter): constructs at the first frame and publishes the golden tag."""
    from isaac_ros_apriltag_amd import node as nd
    img, K, _ = synth.scene_pol_golden()
    K9 = [K[0, 0], 0, K[0, 2], 0, K[1, 1], K[1, 2], 0, 0, 1]
    n = nd.AprilTagNode(max_tags=64, size=0.22, tile_size=8)
    dets, _ = n.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9)
    n.close()
    assert [d["id"] for d in dets] == [0]
    o, _ = po.detect(img, params=pu.oracle_params(K, 1, 0.22, tile_size=8))
    assert len(o) == 1 and np.abs(np.array(dets[0]["corners"]) - o[0]["p"][::-1]).max() < 1e-3
    n5 = nd.AprilTagNode(tile_size=5)
    with pytest.raises(RuntimeError):
        n5.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9)
    n5.close()


@pytest.mark.parametrize("path", PATHS)
def test_c3_4k_board_decimate2(built, path):
    img, K, truth, size = synth.scene_c3()
    h, w = img.shape
    det = AprilTagDetector(w, h, families=("tag36h11",), decimate=2, intrinsics=_k4(K), tag_size=size, max_batch=1)
    det.set_submission_path(path)
    g = det.detect_batch_ex(torch.from_numpy(img).cuda(), max_dets=256)[0]
    assert det.last_submission_path() == path
    errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 2, tag_size=size)
    errs += pu.compare_detections(g, odets)
    det.close()
    assert not errs, errs[:5]
    assert sorted(d["id"] for d in g) == list(range(100))


def test_c3_ten_4k_boards_in_one_submission(built):
    """Config 3 as the throughput set sees it: TEN distinct 3840 x 2160 boards at decimate 2 in one call --
    ten 1920 x 1080 working images are beyond the size up to which a submission takes the latency set, so the library itself
    (path AUTO) runs k_cc_local<4>, k_fit_small, the per-wave prefilter on them -- and every stage of EVERY frame equals the oracle's."""
    scenes = [synth.scene_c3(seed=500 + 11 * i, sigma=(2.0 if i % 2 else 0.0)) for i in range(10)]
    K, size = scenes[0][1], scenes[0][3]
    frames = [np.ascontiguousarray(sc[0]) for sc in scenes]
    det = AprilTagDetector(3840, 2160, decimate=2, intrinsics=_k4(K), tag_size=size, max_batch=len(frames))
    res = det.detect_batch_ex(torch.from_numpy(np.stack(frames)).cuda(), max_dets=128)
    assert det.last_submission_path() == "throughput"
    assert det.frame_flags(len(frames)) == [0] * len(frames)
    for i, img in enumerate(frames):
        errs, odets = pu.compare_stages(det, i, img, ("tag36h11",), K, 2, tag_size=size)
        errs += pu.compare_detections(res[i], odets)
        assert not errs, (i, errs[:4])
        assert sorted(d["id"] for d in res[i]) == list(range(100))
    det.close()


@pytest.mark.parametrize("shape,pitch", [((480, 644), 644), ((477, 635), 640), ((203, 301), 301), ((64, 64), 64), ((33, 70), 83)])
@pytest.mark.parametrize("path", PATHS)
def test_integer_stages_on_noise_ragged_sizes(built, shape, pitch, path):
    """Uniform random bytes (every tile high-contrast, salt-and-pepper components), odd sizes, odd pitch
    (exercises the unaligned loader and the leftover strips)."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    buf = rng.integers(0, 256, size=(shape[0], pitch), dtype=np.uint8)
    img = buf[:, :shape[1]]
    K = synth.default_K(shape[1], shape[0])
    det = AprilTagDetector(shape[1], shape[0], intrinsics=_k4(K), max_batch=1)
    det.set_submission_path(path)
    t = torch.from_numpy(buf).cuda()
    g = det.detect_batch_ex([(t.data_ptr(), pitch)], max_dets=64)[0]
    errs, odets = pu.compare_stages(det, 0, np.ascontiguousarray(img), ("tag36h11",), K, 1)
    errs += pu.compare_detections(g, odets)
    det.close()
    assert not errs, errs[:5]


def test_three_valued_blocks_cc(built):
    """Piecewise-constant image with low-contrast regions: exercises value-127 exclusion and large runs."""
    rng = np.random.default_rng(11)
    base = rng.choice(np.array([20, 128, 130, 240], dtype=np.uint8), size=(40, 60))
    img = np.kron(base, np.ones((8, 8), dtype=np.uint8))
    K = synth.default_K(img.shape[1], img.shape[0])
    det, g = _run(img, K)
    errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 1)
    det.close()
    assert not errs, errs[:5]


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-5] for p in GOLD])
@pytest.mark.parametrize("subpath", PATHS)
def test_against_committed_golden_vectors(built, path, subpath):
    rec = json.load(open(path))
    r = getattr(synth, rec["scene"])(**rec["kwargs"])
    img, K = r[0], r[1]
    assert zlib.crc32(img.tobytes()) == rec["image_crc32"]
    det, g = _run(img, K, tuple(rec["families"]), rec["decimate"], path=subpath)
    h, w = (1 + (img.shape[0] - 1) // rec["decimate"]), (1 + (img.shape[1] - 1) // rec["decimate"])
    assert zlib.crc32(det.debug(0, capi.DBG_THRESH).tobytes()) == rec["thr_crc32"]
    assert zlib.crc32(det.debug(0, capi.DBG_LABEL).tobytes()) == rec["label_crc32"]
    q = det.debug(0, capi.DBG_QUADS)
    qo = np.argsort(q["key"], kind="stable")
    assert [q["p"][i].astype("<f4").tobytes().hex() for i in qo] == rec["quads_hex"]
    det.close()
    assert len(g) == len(rec["detections"])
    for d, gd in zip(g, rec["detections"]):
        assert (d["family"], d["id"], d["hamming"]) == (gd["family"], gd["id"], gd["hamming"])
        assert np.float32(d["decision_margin"]).tobytes().hex() == gd["decision_margin_hex"]
        assert np.asarray(d["p"], dtype="<f8").tobytes().hex() == gd["p_hex"]
        assert np.asarray(d["center"], dtype="<f8").tobytes().hex() == gd["center_hex"]
        assert np.asarray(d["R"], dtype="<f8").tobytes().hex() == gd["R_hex"]
        assert np.asarray(d["t"], dtype="<f8").tobytes().hex() == gd["t_hex"]


def test_reference_pol_golden_through_c_abi(built):
    """The reference's pol_test assertions (tolerances 2 px / 0.01 / 0.01) on the cuAprilTagsID_t-shaped
    records of amdAprilTagsDetect: id, family, corner order, column-major orientation, translation."""
    img, K, _ = synth.scene_pol_golden()
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), tag_size=0.22, max_batch=1)
    tags, cnt = det.detect_batch_raw(torch.from_numpy(img).cuda(), max_tags=64)
    det.close()
    assert cnt[0] >= 1
    for i in range(cnt[0]):
        t = tags[i]
        assert t.id == 0 and t.family == 0
        gold = [(1044.0, 665.0), (808.0, 665.0), (808.0, 429.0), (1044.0, 429.0)]
        for c, g in zip(t.corners, gold):
            assert abs(c.x - g[0]) <= 2 and abs(c.y - g[1]) <= 2
        assert abs(t.center.x - 926.0) <= 2 and abs(t.center.y - 547.0) <= 2
        for v, g in zip(t.translation, (0.255342, 0.098358, 0.403961)):
            assert abs(v - g) <= 0.01
        R = np.array(list(t.orientation)).reshape(3, 3).T   # column-major -> row-major
        assert np.abs(R - np.diag([-1.0, -1.0, 1.0])).max() <= 0.02   # quaternion (0,0,0,1)


def test_corner_convention_switch(built):
    """amdAprilTagsConfig_t.corner_convention (the 180-degree reading of the reference's golden frame
    is an inference, so the other reading stays selectable): AMDAT_CORNERS_ROTATED_180 returns the same detections with the
    corner index turned by two and the orientation multiplied by Rz(pi) from the right; everything else is unchanged, and
    an unknown value is refused at creation."""
    img, K, _ = synth.scene_c2(seed=1234, sigma=2.0)
    t = torch.from_numpy(img).cuda()
    out = []
    for conv in (0, 1):
        det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), tag_size=0.22, max_batch=1, corner_convention=conv)
        tags, cnt = det.detect_batch_raw(t, max_tags=64)
        det.close()
        out.append([tags[i] for i in range(cnt[0])])
    a, b = out
    assert len(a) == len(b) == 10
    for x, y in zip(a, b):
        assert (x.id, x.family, x.hamming_error) == (y.id, y.family, y.hamming_error)
        for i in range(4):
            assert (x.corners[i].x, x.corners[i].y) == (y.corners[(i + 2) & 3].x, y.corners[(i + 2) & 3].y)
        Rx = np.array(list(x.orientation)).reshape(3, 3).T
        Ry = np.array(list(y.orientation)).reshape(3, 3).T
        assert np.array_equal(Ry, Rx @ np.diag([-1.0, -1.0, 1.0]))
        assert list(x.translation) == list(y.translation) and (x.center.x, x.center.y) == (y.center.x, y.center.y)
    with pytest.raises(capi.AprilTagsError):
        AprilTagDetector(1920, 1080, intrinsics=_k4(K), corner_convention=2)


def _bt601(rgb):
    """numpy statement of the conversion: Y = (4899 R + 9617 G + 1868 B + 8192) >> 14 (the 14-bit fixed-point
    BT.601 weights 0.299 / 0.587 / 0.114 that cv_bridge / OpenCV apply for the reference's mono8 test input)."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8)


@pytest.mark.parametrize("encoding,nch,order", [("rgb8", 3, (0, 1, 2)), ("bgr8", 3, (2, 1, 0)), ("rgba8", 4, (0, 1, 2)),
                                                ("bgra8", 4, (2, 1, 0))])
def test_colour_conversion_random_pixels(built, encoding, nch, order):
    """All four colour encodings of apriltag_node.cpp:76-82 on RANDOM colours (so that a swapped channel order
    cannot pass), odd width, padded pitch, against the numpy BT.601 statement."""
    rng = np.random.default_rng(len(encoding) * 7 + nch)
    h, w = 37, 1001
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    pitch = w * nch + 12
    buf = rng.integers(0, 256, size=(h, pitch), dtype=np.uint8)       # garbage in the padding and in alpha
    px = buf[:, :w * nch].reshape(h, w, nch)
    for c in range(3):
        px[..., order[c]] = rgb[..., c]                                # channel `order[c]` of the pixel holds R/G/B
    src = torch.from_numpy(buf).cuda()
    dpitch = 1024
    dst = torch.zeros((h, dpitch), dtype=torch.uint8, device="cuda")
    rc = capi.lib().amdAprilTagsConvertToMono8(src.data_ptr(), pitch, encoding.encode(), w, h, dst.data_ptr(), dpitch, None)
    assert rc == 0
    got = dst.cpu().numpy()
    assert np.array_equal(got[:, :w], _bt601(rgb))
    assert (got[:, w:] == 0).all()                                      # nothing written beyond the row


def test_bgr8_input_through_conversion(built):
    """Reference fixture encoding is bgr8 (test_cases/apriltag0/image.json): replicate the gray frame
    into 3 channels, convert on the device, detect."""
    img, K, _ = synth.scene_pol_golden()
    bgr = np.repeat(img[:, :, None], 3, axis=2).copy()
    src = torch.from_numpy(bgr).cuda()
    dst = torch.empty((1080, 1920), dtype=torch.uint8, device="cuda")
    rc = capi.lib().amdAprilTagsConvertToMono8(src.data_ptr(), 1920 * 3, b"bgr8", 1920, 1080, dst.data_ptr(), 1920, None)
    assert rc == 0
    assert torch.equal(dst.cpu(), torch.from_numpy(img))   # (4899+9617+1868)/16384 == 1 exactly
    assert capi.lib().amdAprilTagsConvertToMono8(src.data_ptr(), 1920 * 3, b"yuv422", 1920, 1080, dst.data_ptr(), 1920, None) == 2
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=1)
    g = det.detect_batch_ex(dst)[0]
    det.close()
    assert [d["id"] for d in g] == [0]


def _colour_frame(gray_like, encoding, seed, pitch_pad=0):
    """A colour frame whose BT.601 gray value is NOT the input (random chroma around it), in the given encoding, with `pitch_pad`
    bytes of garbage behind every row; returns (host buffer [h, pitch], the numpy-converted gray frame)."""
    rng = np.random.default_rng(seed)
    h, w = gray_like.shape
    nch = capi.ENC_CHANNELS[encoding]
    order = (0, 1, 2) if encoding in ("rgb8", "rgba8") else (2, 1, 0)
    base = gray_like.astype(np.int32)
    rgb = np.stack([np.clip(base + rng.integers(-40, 41, size=(h, w)), 0, 255) for _ in range(3)], axis=2).astype(np.uint8)
    pitch = w * nch + pitch_pad
    buf = rng.integers(0, 256, size=(h, pitch), dtype=np.uint8)
    px = buf[:, :w * nch].reshape(h, w, nch)
    for c in range(3):
        px[..., order[c]] = rgb[..., c]
    return buf, _bt601(rgb)


@pytest.mark.parametrize("encoding", ["rgb8", "bgr8", "rgba8", "bgra8"])
@pytest.mark.parametrize("path", PATHS)
def test_colour_frames_through_the_fused_threshold_loader(built, encoding, path):
    """Colour frames as the reference feeds them (apriltag_node.cpp:469-486 rgb8 / bgr8 uchar3; the table of :76-82): one
    amdAprilTagsDetectBatchColorEx call, no conversion launch -- the threshold pass reads the interleaved frame, writes the gray
    plane and thresholds.  Gray plane, threshold image and every later stage equal the oracle's run on the numpy-converted
    frame bit for bit, on both launch sets; config 2 with random chroma (a swapped channel order cannot pass), 16-byte aligned
    and unaligned pitches."""
    img, K, _ = synth.scene_c2(seed=1402)
    for pad in (0, 7):
        buf, gray = _colour_frame(img, encoding, seed=len(encoding) + pad, pitch_pad=pad)
        nch = capi.ENC_CHANNELS[encoding]
        det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=1)
        det.set_submission_path(path)
        src = torch.from_numpy(buf).cuda()
        g = det.detect_batch_ex([(src.data_ptr(), buf.shape[1])], max_dets=64, encoding=encoding)[0]
        errs, odets = pu.compare_stages(det, 0, gray, ("tag36h11",), K)
        errs += pu.compare_detections(g, odets)
        # the mono8 call on the converted frame gives the same records, and so does the colour call again (graph replay)
        g2 = det.detect_batch_ex(torch.from_numpy(gray).cuda(), max_dets=64)[0]
        g3 = det.detect_batch_ex([(src.data_ptr(), buf.shape[1])], max_dets=64, encoding=encoding)[0]
        errs += pu.compare_detections(g2, odets) + pu.compare_detections(g3, odets)
        det.close()
        assert not errs, (pad, errs[:4])
        assert len(g) == 10


@pytest.mark.parametrize("shape,decimate,tile", [((477, 635), 1, 4), ((203, 301), 1, 4), ((480, 640), 2, 4), ((480, 640), 1, 8),
                                                 ((36, 52), 1, 4)])
def test_colour_frames_ragged_decimated_and_tile8(built, shape, decimate, tile):
    """The colour entry point off the fast case: sizes that are no multiple of 4 or 16 (the slow loader and the leftover
    kernel read the colour frame too), and the configurations that take the conversion launch inside the submission
    (decimate 2, tile_size 8).  Random colour noise; every stage against the oracle on the converted frame."""
    h, w = shape
    rng = np.random.default_rng(h * 7 + w + decimate + tile)
    K = synth.default_K(w, h)
    for encoding in ("bgr8", "rgba8"):
        nch = capi.ENC_CHANNELS[encoding]
        buf = rng.integers(0, 256, size=(h, w * nch + 5), dtype=np.uint8)
        px = buf[:, :w * nch].reshape(h, w, nch)
        rgb = px[..., :3] if encoding == "rgba8" else px[..., ::-1]
        gray = _bt601(np.ascontiguousarray(rgb))
        det = AprilTagDetector(w, h, decimate=decimate, tile_size=tile, intrinsics=_k4(K), max_batch=2)
        src = torch.from_numpy(buf).cuda()
        r = det.detect_batch_ex([(src.data_ptr(), buf.shape[1])] * 2, max_dets=64, encoding=encoding)
        for f in (0, 1):
            errs, odets = pu.compare_stages(det, f, gray, ("tag36h11",), K, decimate, tile_size=tile)
            errs += pu.compare_detections(r[f], odets)
            assert not errs, (encoding, f, errs[:4])
        det.close()


def test_colour_batch_on_the_throughput_set_and_errors(built):
    """Ten bgr8 1080p frames in one call (path AUTO: the throughput set), every stage of every frame; Submit / Wait with an
    encoding; and the argument checks of the colour calls."""
    K = synth.default_K(1920, 1080)
    pairs = [_colour_frame(synth.scene_c2(seed=1500 + 3 * i)[0], "bgr8", seed=i) for i in range(10)]
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=10)
    srcs = [torch.from_numpy(b).cuda() for b, _ in pairs]
    frames = [(t.data_ptr(), t.shape[1]) for t in srcs]
    res = det.detect_batch_ex(frames, max_dets=64, encoding="bgr8")
    assert det.last_submission_path() == "throughput"
    _check_every_frame(det, res, [g for _, g in pairs], ("tag36h11",), K)
    prep = det.prepare(frames, max_dets=64, encoding="bgr8")
    det.submit_prepared(prep)
    det.wait_prepared(prep)
    for a, b in zip(det.unpack(prep), res):
        assert not pu.compare_detections(a, b)
    L = capi.lib()
    assert L.amdAprilTagsEncodingFromName(b"bgr8") == 2 and L.amdAprilTagsEncodingFromName(b"yuv422") == -1
    imgs = (capi.ImageInput * 1)()
    imgs[0].width, imgs[0].height, imgs[0].dev_ptr, imgs[0].pitch = 1920, 1080, srcs[0].data_ptr(), 1920 * 3 - 1
    out, cnt = (capi.TagID * 16)(), (C.c_uint32 * 1)()
    assert L.amdAprilTagsDetectColor(det._h, imgs, 2, out, cnt, 16, None) == 1      # pitch below three bytes per pixel
    imgs[0].pitch = 1920 * 3
    assert L.amdAprilTagsDetectColor(det._h, imgs, 7, out, cnt, 16, None) == 2      # no such encoding
    assert L.amdAprilTagsDetectColor(det._h, imgs, 2, out, cnt, 16, None) == 0 and cnt[0] == 10
    det.close()


def test_skew_in_pose(built):
    """K[0][1] of the VPI path (apriltag_node.cpp:215-225): bit-identical to the oracle's skewed pose solve, and
    different from the skew-free pose; corners do not depend on the intrinsics."""
    img, K, _ = synth.scene_c2(seed=1250, sigma=0.0)
    det0 = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=1)
    det1 = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=1, skew=3.25)
    t = torch.from_numpy(img).cuda()
    g0, g1 = det0.detect_batch_ex(t)[0], det1.detect_batch_ex(t)[0]
    det0.close(); det1.close()
    prm = pu.oracle_params(K, 1, 0.22)
    prm.skew = 3.25
    o1 = po.detect(img, params=prm)[0]
    assert len(g1) == 10 and not pu.compare_detections(g1, o1)
    for a, b in zip(g0, g1):
        assert np.array_equal(a["p"], b["p"]) and not np.array_equal(a["t"], b["t"])


@pytest.mark.parametrize("more", [
    {"refine_edges": 0},
    {"max_hamming": 0},
    {"max_hamming": 1, "decode_sharpening": 0.0},
    {"max_hamming": 3, "decode_sharpening": 1.0, "skew": 1.5},
    {"refine_edges": 0, "max_hamming": 3, "decode_sharpening": 0.5},
], ids=lambda m: ",".join("%s=%s" % kv for kv in m.items()))
def test_decode_parameters_beside_their_defaults(built, more):
    """refine_edges, max_hamming, decode_sharpening (cuAprilTags fixes them; the VPI-shaped path and AprilRobotics expose them) at other
    values than the defaults every other test runs: a noisy 1080p frame under tag36h11 + tag16h5 -- the second family turns noise
    blobs into dozens of hamming-1 .. 3 records, so the hamming bound and the sharpened decision margins are exercised on real
    candidates -- every stage and every record bit-identical to the oracle with the same parameters, tag size 5 cm."""
    img, K, _ = synth.scene_c2(sigma=2.0)
    fams = ("tag36h11", "tag16h5")
    det = AprilTagDetector(1920, 1080, families=fams, intrinsics=_k4(K), max_batch=1, tag_size=0.05, **more)
    got = det.detect_batch_ex(torch.from_numpy(img).cuda(), max_dets=512)[0]
    errs, odets = pu.compare_stages(det, 0, img, fams, K, 1, tag_size=0.05, **more)
    det.close()
    errs += pu.compare_detections(got, odets)
    assert not errs, errs[:3]
    assert len(odets) >= 10 and sum(1 for d in odets if d["family"] == "tag36h11" and d["hamming"] == 0) == 10
    hs = {d["hamming"] for d in odets}
    assert max(hs) <= more.get("max_hamming", 2)
    if more.get("max_hamming", 2) == 3:
        assert 3 in hs   # (the bound is reached: the case is not vacuous)


@pytest.mark.parametrize("encoding", ["mono8", "bgr8", "rgba8"])
@pytest.mark.parametrize("path", PATHS)
def test_frames_at_any_address_with_pitches_of_their_own(built, encoding, path):
    """The frames of one submission as regions of larger device buffers: base addresses at any byte (1, 7, 13 past an aligned
    block, one aligned) and a pitch of its own per frame (cuAprilTagsImageInput_t carries both per image, apriltag_node.cpp:481-486).
    Every stage and every record of every frame equals the oracle on the frame's gray content."""
    nch = capi.ENC_CHANNELS[encoding]
    rng = np.random.default_rng(77)
    frames, ptrs, keep = [], [], []
    for i, (off, pad) in enumerate([(1, 0), (7, 3), (0, 64), (13, 29)]):
        gray = synth.scene_c2(seed=1300 + i)[0]
        if nch == 1:
            px = gray[..., None]
        else:   # equal channels convert back to the gray value exactly: (4899 + 9617 + 1868) v + 8192 >> 14 = v
            px = np.repeat(gray[..., None], nch, axis=2)
            if nch == 4:
                px[..., 3] = rng.integers(0, 256, size=gray.shape, dtype=np.uint8)
        h, w = gray.shape
        pitch = w * nch + pad
        fb = rng.integers(0, 256, size=off + h * pitch + 16, dtype=np.uint8)
        fb[off:off + h * pitch].reshape(h, pitch)[:, :w * nch] = px.reshape(h, w * nch)
        t = torch.from_numpy(fb).cuda()
        keep.append(t); frames.append(gray); ptrs.append((t.data_ptr() + off, pitch))
    K = synth.default_K(1920, 1080)
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=4)
    det.set_submission_path(path)
    got = det.detect_batch_ex(ptrs, max_dets=64, encoding=encoding)
    for i, gray in enumerate(frames):
        errs, odets = pu.compare_stages(det, i, gray, ("tag36h11",), K, 1)
        errs += pu.compare_detections(got[i], odets)
        assert not errs, (i, errs[:3])
        assert len(odets) == 10
    det.close()


def test_4k_decimate1_batch(built):
    """3840x2160 at decimate 1 in a batch: clusters may exceed the LDS key array (3(2W+2H) = 36000 points), so the
    last size class sorts in its global scratch slot; the handle's memory stays bounded (no per-point moment
    arrays).  Detections bit-identical to the oracle."""
    img, K, truth, size = synth.scene_c3(seed=77, sigma=2.0)
    img2 = synth.scene_c3(seed=78, sigma=2.0)[0]
    det = AprilTagDetector(3840, 2160, intrinsics=_k4(K), tag_size=size, max_batch=4)
    assert det.device_bytes() < 4 * 1000e6
    batch = torch.from_numpy(np.stack([img, img2, img, img2])).cuda()
    r = det.detect_batch_ex(batch, max_dets=128)
    assert det.frame_flags(4) == [0, 0, 0, 0]
    assert det.last_submission_path() == "throughput"
    # every stage of the two distinct frames: working images above 2048 x 2048 have no k_fit_small, so all
    # clusters go through k_fit_quads' classes and the large ones through the per-wave prefilter
    for f, im in ((0, img), (1, img2)):
        errs, o = pu.compare_stages(det, f, im, ("tag36h11",), K, 1, tag_size=size)
        assert not errs, (f, errs[:4])
        assert sorted(d["id"] for d in o) == list(range(100))
        assert not pu.compare_detections(r[f], o)
        assert not pu.compare_detections(r[f + 2], o)
    det.close()


def test_batch_properties_full_size(built):
    """Size-independent properties at BASELINE.json's full size: a batch equals the per-frame results
    (frames are independent), is permutation-equivariant, deterministic across runs, and per-frame
    intrinsics only change the pose."""
    frames = [synth.scene_c2(seed=1234 + i)[0] for i in range(4)]
    K = synth.default_K(1920, 1080)
    batch = torch.from_numpy(np.stack(frames)).cuda()
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=8)
    r1 = det.detect_batch_ex(batch)
    r2 = det.detect_batch_ex(batch)
    perm = [2, 0, 3, 1]
    r3 = det.detect_batch_ex(batch[perm].contiguous())
    single = [det.detect_batch_ex(batch[i])[0] for i in range(4)]
    K2 = [(1200.0, 1190.0, 950.0, 530.0)] * 4
    r4 = det.detect_batch_ex(batch, intrinsics=K2)
    assert det.frame_flags(4) == [0, 0, 0, 0]
    det.close()
    for i in range(4):
        assert not pu.compare_detections(r1[i], r2[i])
        assert not pu.compare_detections(r1[i], single[i])
        assert not pu.compare_detections(r3[i], r1[perm[i]])
        assert [d["id"] for d in r1[i]] == list(range(10))
        for a, b in zip(r1[i], r4[i]):
            assert np.array_equal(a["p"], b["p"]) and not np.array_equal(a["t"], b["t"])
    # pose of frame 0 under K2 equals the oracle's pose for those intrinsics
    o, _ = po.detect(frames[0], params=pu.oracle_params(np.array([[1200.0, 0, 950.0], [0, 1190.0, 530.0], [0, 0, 1]])))
    assert not pu.compare_detections(r4[0], o)


def test_error_behaviour(built):
    img, K, _ = synth.scene_c1()
    det = AprilTagDetector(640, 480, intrinsics=_k4(K), max_batch=2)
    t = torch.from_numpy(img).cuda()
    L = capi.lib()
    imgs = (capi.ImageInput * 3)()
    for i in range(3):
        imgs[i].width, imgs[i].height, imgs[i].dev_ptr, imgs[i].pitch = 640, 480, t.data_ptr(), 640
    out = (capi.TagID * 192)()
    cnt = (C.c_uint32 * 3)()
    assert L.amdAprilTagsDetectBatch(det._h, 3, imgs, None, out, cnt, 64, None) == 6      # batch too large
    imgs[0].width = 320
    assert L.amdAprilTagsDetectBatch(det._h, 1, imgs, None, out, cnt, 64, None) == 4      # size mismatch
    imgs[0].width = 640
    imgs[0].dev_ptr = None
    assert L.amdAprilTagsDetectBatch(det._h, 1, imgs, None, out, cnt, 64, None) == 1      # null image
    imgs[0].dev_ptr = t.data_ptr()
    imgs[0].pitch = 1 << 23                                                               # frame spans more than 2^31 bytes
    assert L.amdAprilTagsDetectBatch(det._h, 1, imgs, None, out, cnt, 64, None) == 1
    imgs[0].pitch = 640
    assert L.amdAprilTagsDetect(det._h, imgs, out, cnt, 64, None) == 0 and cnt[0] == 1
    # capacity overflow is reported, never UB
    small = AprilTagDetector(640, 480, intrinsics=_k4(K), max_batch=1, max_points=1000)
    small.detect_batch_ex(t)
    assert small.frame_flags(1)[0] & 0x1
    small.close()
    det.close()


def test_node_shell_pol_and_mono8(built):
    """The reference's launch tests through the node shell: pol_test (bgr8 host image -> assertions of
    isaac_ros_apriltag_pol_test.py:113-175), mono8_test (>= 1 detection), TF naming, header mapping,
    ExactTime gating and the encoding error."""
    from isaac_ros_apriltag_amd import node as nd
    img, K, _ = synth.scene_pol_golden()
    bgr = np.ascontiguousarray(np.repeat(img[:, :, None], 3, axis=2))
    K9 = [K[0, 0], 0, K[0, 2], 0, K[1, 1], K[1, 2], 0, 0, 1]
    n = nd.AprilTagNode(max_tags=64, size=0.22, tile_size=4)
    dets, frame_id = n.on_frame(bgr.ctypes.data, False, "bgr8", 1920, 1080, 1920 * 3, K9, frame_id="tf_camera")
    assert frame_id == "tf_camera" and len(dets) >= 1
    for d in dets:
        assert d["id"] == 0 and d["family"] == "tag36h11" and d["child_frame_id"] == "tag36h11:0"
        gold = [(1044.0, 665.0), (808.0, 665.0), (808.0, 429.0), (1044.0, 429.0)]
        for c, g in zip(d["corners"], gold):
            assert abs(c[0] - g[0]) <= 2 and abs(c[1] - g[1]) <= 2
        assert abs(d["center"][0] - 926.0) <= 2 and abs(d["center"][1] - 547.0) <= 2
        for v, g in zip(d["position"], (0.255342, 0.098358, 0.403961)):
            assert abs(v - g) <= 0.01
        x, y, z, w = d["orientation_xyzw"]
        if z < 0:
            x, y, z, w = -x, -y, -z, -w
        assert abs(w) <= 0.01 and abs(x) <= 0.01 and abs(y) <= 0.01 and abs(z - 1.0) <= 0.01
    # mono8 host image and device pointer
    d2, _ = n.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9)
    assert len(d2) >= 1 and d2[0]["corners"] == dets[0]["corners"]
    t = torch.from_numpy(img).cuda()
    d3, _ = n.on_frame(t.data_ptr(), True, "mono8", 1920, 1080, 1920, K9)
    assert d3[0]["corners"] == dets[0]["corners"]
    # stamps differ -> the synchroniser does not fire
    assert n.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9, stamp=(1, 0), info_stamp=(1, 5)) == (None, None)
    with pytest.raises(RuntimeError):
        n.on_frame(img.ctypes.data, False, "yuv422", 1920, 1080, 1920 * 2, K9)
    # a later frame of another size (or a step too small for its width) is dropped before any device write
    big = np.zeros((1200, 2048, 3), dtype=np.uint8)
    assert n.on_frame(big.ctypes.data, False, "bgr8", 2048, 1200, 2048 * 3, K9)[0] == []
    assert n.on_frame(bgr.ctypes.data, False, "bgr8", 1920, 1080, 1920 * 2, K9)[0] == []
    d4, _ = n.on_frame(bgr.ctypes.data, False, "bgr8", 1920, 1080, 1920 * 3, K9)
    assert d4 == dets                                      # the node keeps working afterwards
    n.close()
    # VPI-mode node (any backend list other than exactly CUDA) passes the skew K[1] on
    K9s = list(K9)
    K9s[1] = 4.0
    nv = nd.AprilTagNode(backends="CUDA,CPU")
    dv, _ = nv.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9s)
    nc = nd.AprilTagNode(backends="CUDA")
    dc, _ = nc.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9s)
    assert dv[0]["corners"] == dc[0]["corners"] and dv[0]["position"] != dc[0]["position"]
    assert dc[0]["position"] == dets[0]["position"]        # cuAprilTags mode has no skew term
    nv.close(); nc.close()


def test_node_shell_colour_frames_and_strict_cuapriltags_encodings(built):
    """Colour frames as the reference's cuAprilTags branch takes them (apriltag_node.cpp:469-486): the shell hands the frame to
    amdAprilTagsDetectColor in the encoding it arrived in -- random chroma, so a swapped channel order cannot pass; host and device
    payloads, padded step -- and publishes what a node fed the numpy-converted mono8 frame publishes.  With
    strict_cuapriltags_encodings the cuAprilTags mode refuses everything but rgb8 / bgr8 with the reference's own text (:469-476);
    a VPI-mode node keeps the five encodings of :76-82."""
    from isaac_ros_apriltag_amd import node as nd
    img, K, _ = synth.scene_c2(seed=1777)
    K9 = [K[0, 0], 0, K[0, 2], 0, K[1, 1], K[1, 2], 0, 0, 1]
    ref = nd.AprilTagNode()
    want = {}
    n = nd.AprilTagNode()
    for enc in ("rgb8", "bgr8", "rgba8", "bgra8"):
        buf, gray = _colour_frame(img, enc, seed=len(enc) * 3, pitch_pad=8)
        want[enc], _ = ref.on_frame(gray.ctypes.data, False, "mono8", 1920, 1080, 1920, K9)
        assert len(want[enc]) == 10
        got, _ = n.on_frame(buf.ctypes.data, False, enc, 1920, 1080, buf.shape[1], K9)
        assert got == want[enc], enc
        t = torch.from_numpy(buf).cuda()
        got_dev, _ = n.on_frame(t.data_ptr(), True, enc, 1920, 1080, buf.shape[1], K9)
        assert got_dev == want[enc], enc
    n.close()
    strict = nd.AprilTagNode(backends="CUDA", strict_cuapriltags_encodings=True)
    buf, gray = _colour_frame(img, "bgr8", seed=6)
    want_strict, _ = ref.on_frame(gray.ctypes.data, False, "mono8", 1920, 1080, 1920, K9)
    ref.close()
    d, _ = strict.on_frame(buf.ctypes.data, False, "bgr8", 1920, 1080, buf.shape[1], K9)
    assert len(d) == 10 and d == want_strict
    for enc, step in (("mono8", 1920), ("rgba8", 1920 * 4)):
        with pytest.raises(RuntimeError) as e:
            strict.on_frame(img.ctypes.data, False, enc, 1920, 1080, step, K9)
        assert "cuAprilTags detector only supports 'rgb8' or 'bgr8' image input" in str(e.value)
    strict.close()
    vpi = nd.AprilTagNode(backends="CUDA,CPU", strict_cuapriltags_encodings=True)     # not the cuAprilTags mode: nothing to be strict about
    assert len(vpi.on_frame(img.ctypes.data, False, "mono8", 1920, 1080, 1920, K9)[0]) == 10
    vpi.close()


@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 20), (5, 7), (64, 4)])
def test_tiny_and_degenerate_frames(built, shape):
    """Smallest legal sizes, flat frames (everything 127) and frames without any tag: no detections, no
    flags, integer stages still bit-exact."""
    h, w = shape
    K = synth.default_K(w, h)
    rng = np.random.default_rng(h * 31 + w)
    for img in (np.full((h, w), 77, dtype=np.uint8), rng.integers(0, 256, size=(h, w), dtype=np.uint8)):
        det, g = _run(img, K)
        errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 1)
        det.close()
        assert not errs, errs[:3]
        assert g == [] and odets == []


@pytest.mark.parametrize("path", PATHS)
def test_tiny_working_image_whose_clusters_all_fit_the_small_class(built, path):
    """Found by the fuzzer in round 5 (seed 61002, case 4254): a 25 x 24 frame at decimate 3 is a 9 x 8 working image, where no
    cluster can exceed 3 (2 W + 2 H) = 102 points -- below the 128-point bound of the small-cluster class.  The latency set buckets
    every cluster from 24 points on into the one-wave class of k_fit_quads, which was neither allocated nor launched on such a
    handle: the frame's only cluster (34 points, one quad) was dropped, while the throughput set (k_fit_small) found it."""
    img = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzz_r05_tiny_working_image.npy"))
    assert img.shape == (24, 25)
    K = synth.default_K(25, 24)
    det, g = _run(img, K, ("tag16h5",), 3, path=path)
    errs, odets = pu.compare_stages(det, 0, img, ("tag16h5",), K, 3)
    errs += pu.compare_detections(g, odets)
    nq = len(det.debug(0, capi.DBG_QUADS))
    det.close()
    assert not errs, errs[:3]
    assert nq == 1


def test_create_rejects_unsupported_sizes(built):
    L = capi.lib()
    h = C.c_void_p()
    cam = capi.Intrinsics(100, 100, 1, 1)
    assert L.amdCreateAprilTagsDetector(C.byref(h), 3, 3, 4, 0, C.byref(cam), 0.1) == 2       # smaller than one tile
    assert L.amdCreateAprilTagsDetector(C.byref(h), 10000, 100, 4, 0, C.byref(cam), 0.1) == 2  # 2x+1 must fit 14 bits
    with pytest.raises(capi.AprilTagsError) as e:
        AprilTagDetector(640, 480, decimate=5)             # the threshold loader exists for decimate 1..4
    assert e.value.code == 2
    with pytest.raises(capi.AprilTagsError) as e:
        AprilTagDetector(640, 480, max_hamming=4)
    assert e.value.code == 1


def test_max_tags_truncation_and_order(built):
    """More tags than max_tags: the first max_tags records of the canonical order are returned."""
    img, K, truth = synth.scene_c2(sigma=0)
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=1)
    t = torch.from_numpy(img).cuda()
    full = det.detect_batch_ex(t, max_dets=64)[0]
    part = det.detect_batch_ex(t, max_dets=4)[0]
    tags, cnt = det.detect_batch_raw(t, max_tags=3)
    det.close()
    assert [d["id"] for d in full] == list(range(10))
    assert not pu.compare_detections(part, full[:4])
    assert cnt == [3] and [tags[i].id for i in range(3)] == [0, 1, 2]


def test_mixed_batch_with_empty_frames(built):
    """A batch mixing tag frames, a flat frame and a pure-noise frame equals the per-frame oracle."""
    a = synth.scene_c2(seed=1250)[0]
    b = np.full((1080, 1920), 200, dtype=np.uint8)
    c = np.random.default_rng(3).integers(0, 256, size=(1080, 1920), dtype=np.uint8)
    d = synth.scene_c2(seed=1251, sigma=0)[0]
    K = synth.default_K(1920, 1080)
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=4)
    res = det.detect_batch_ex(torch.from_numpy(np.stack([a, b, c, d])).cuda(), max_dets=64)
    flags = det.frame_flags(4)
    for i, img in enumerate((a, b, c, d)):
        errs, odets = pu.compare_stages(det, i, img, ("tag36h11",), K, 1)
        errs += pu.compare_detections(res[i], odets)
        assert not errs, (i, errs[:3])
    det.close()
    assert flags == [0, 0, 0, 0]
    assert len(res[0]) == 10 and res[1] == [] and len(res[3]) == 10


def test_large_single_cluster_uses_global_sort_path(built):
    """One component pair with 27 000 boundary points (4K frame, where clusters up to 36 000 points are
    legal): larger than the 16 384-key LDS sort of the biggest size class, so the quad fit sorts in
    global scratch; result still equals the oracle."""
    W, H = 3840, 2160
    img = np.full((H, W), 200, dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    ang = np.arctan2(yy - H / 2, xx - W / 2)
    rad = np.hypot(yy - H / 2, xx - W / 2)
    img[rad < 600 + 100 * np.sin(24 * ang)] = 30   # wavy star: long boundary, one black/white pair
    K = synth.default_K(W, H)
    det, g = _run(img, K)
    cl = det.debug(0, capi.DBG_CLUSTERS)
    errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 1)
    errs += pu.compare_detections(g, odets)
    det.close()
    assert not errs, errs[:3]
    assert len(cl) == 1 and 16384 < cl["count"].max() <= 36000


def test_front_steps_bit_exact(built):
    """Resize and rectify kernels reproduce the oracle's bytes, and 4K -> resize -> detect equals the
    oracle run on the resized frame (the chain the reference README recommends for 4K input)."""
    L = capi.lib()
    img, K, truth, size = synth.scene_c3()
    src = torch.from_numpy(img).cuda()
    dst = torch.empty((1080, 1920), dtype=torch.uint8, device="cuda")
    assert L.amdAprilTagsResizeMono8(src.data_ptr(), 3840, 3840, 2160, dst.data_ptr(), 1920, 1920, 1080, None) == 0
    ref = po.resize_mono8(img, 1920, 1080)
    assert np.array_equal(dst.cpu().numpy(), ref)
    Ks = np.array([[2000.0, 0, 960.0], [0, 2000.0, 540.0], [0, 0, 1]])
    det = AprilTagDetector(1920, 1080, families=("tag36h11",), intrinsics=_k4(Ks), tag_size=size, max_batch=1)
    g = det.detect_batch_ex(dst, max_dets=128)[0]
    det.close()
    o, _ = po.detect(ref, families=("tag36h11",), params=pu.oracle_params(Ks, 1, size))
    assert not pu.compare_detections(g, o) and len(g) == 100
    # odd sizes, up- and down-scaling
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, size=(203, 301), dtype=np.uint8)
    ta = torch.from_numpy(a).cuda()
    for (dw, dh) in ((301, 203), (97, 55), (640, 480), (1, 1)):
        out = torch.empty((dh, dw), dtype=torch.uint8, device="cuda")
        assert L.amdAprilTagsResizeMono8(ta.data_ptr(), 301, 301, 203, out.data_ptr(), dw, dw, dh, None) == 0
        assert np.array_equal(out.cpu().numpy(), po.resize_mono8(a, dw, dh)), (dw, dh)
    # rectify with a plumb_bob model
    img1, K1, _ = synth.scene_c1()
    D = [-0.25, 0.07, 0.001, -0.002, 0.01]
    Kn = np.array([[480.0, 0, 330.0], [0, 470.0, 235.0], [0, 0, 1]])
    t1 = torch.from_numpy(img1).cuda()
    r = torch.empty_like(t1)
    k = (C.c_double * 9)(*K1.reshape(-1)); d5 = (C.c_double * 5)(*D); kn = (C.c_double * 9)(*Kn.reshape(-1))
    assert L.amdAprilTagsRectifyMono8(t1.data_ptr(), 640, r.data_ptr(), 640, 640, 480, k, d5, kn, None) == 0
    assert np.array_equal(r.cpu().numpy(), po.rectify_mono8(img1, K1, D, Kn))
    assert L.amdAprilTagsRectifyMono8(t1.data_ptr(), 640, r.data_ptr(), 640, 640, 480, k, d5, None, None) == 1


def test_random_scene_sweep(built):
    """Seeded sweep over 48 random VGA scenes (1-4 tags of random family/id/pose, noise sigma 0-8,
    decimate 1-2, one or two families enabled): detections bit-identical to the oracle on every one."""
    rng = np.random.default_rng(20260927)
    fams_all = ("tag36h11", "tag25h9", "tag16h5")
    K = synth.default_K(640, 480)
    dets_by_key = {}
    total = 0
    for case in range(48):
        ntags = int(rng.integers(1, 5))
        enabled = tuple(rng.choice(fams_all, size=int(rng.integers(1, 3)), replace=False))
        tags = []
        for t in range(ntags):
            fam = str(rng.choice(fams_all))
            ncodes = {"tag36h11": 587, "tag25h9": 35, "tag16h5": 30}[fam]
            side = float(rng.uniform(40, 110))
            cx, cy = float(rng.uniform(120, 520)), float(rng.uniform(100, 380))
            R = synth.rot_xyz(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-3.1, 3.1)))
            z = K[0, 0] * 0.1 / side
            tvec = np.array([(cx - 320) / K[0, 0] * z, (cy - 240) / K[1, 1] * z, z])
            tags.append({"family": fam, "id": int(rng.integers(0, ncodes)), "H": synth.homography_from_pose(R, tvec, K, 0.1)})
        sigma = float(rng.choice([0.0, 1.0, 2.0, 4.0, 8.0]))
        dec = int(rng.choice([1, 2]))
        img = synth.render(640, 480, tags, background=int(rng.integers(90, 200)), sigma=sigma, seed=1000 + case)
        key = (enabled, dec)
        if key not in dets_by_key:
            dets_by_key[key] = AprilTagDetector(640, 480, families=enabled, decimate=dec, intrinsics=_k4(K), tag_size=0.1, max_batch=1)
        det = dets_by_key[key]
        g = det.detect_batch_ex(torch.from_numpy(img).cuda(), max_dets=64)[0]
        o, _ = po.detect(img, families=enabled, params=pu.oracle_params(K, dec, 0.1))
        errs = pu.compare_detections(g, o)
        assert not errs, (case, enabled, dec, sigma, errs[:3])
        assert det.frame_flags(1) == [0]
        total += len(o)
    for d in dets_by_key.values():
        d.close()
    assert total >= 30   # the sweep does exercise real detections


@pytest.mark.parametrize("path", PATHS)
def test_adversarial_content_fuzz(built, path):
    """300 seeded cases from tools/fuzz_gpu.py: checkerboards, stripes, gradients, rings, impulses, two-level
    noise, rectangles and tag scenes at ragged sizes (4..300 px), odd pitches, decimate 1-4, one to three
    families: every stage and every detection bit-identical to the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_gpu", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    done, fails = fz.run_cases(300, seed=20260928, maxdim=300, out=lambda m: None, path=path)
    assert done == 300 and not fails, fails[:3]
    # ... and 60 cases of FIVE frames per submission, every frame with content of its own (frame indexing of every stage)
    done, fails = fz.run_cases(60, seed=20260929, maxdim=260, out=lambda m: None, path=path, batch=5)
    assert done == 60 and not fails, fails[:3]


def test_checkerboard_more_quads_than_the_old_fixed_capacity(built):
    """A 1262x1162 checkerboard of 9-px cells fits ~20 000 quads (every cell is one); the default quad
    capacity follows the cluster capacity, so nothing is dropped and no overflow flag is raised."""
    yy, xx = np.mgrid[0:1162, 0:1262]
    img = ((((yy // 9) + (xx // 9)) & 1) * 200 + 20).astype(np.uint8)
    K = synth.default_K(1262, 1162)
    det, g = _run(img, K, families=("tag16h5",))
    errs, odets = pu.compare_stages(det, 0, img, ("tag16h5",), K, 1)
    errs += pu.compare_detections(g, odets)
    nq = len(det.debug(0, capi.DBG_QUADS))
    det.close()
    assert not errs, errs[:3]
    assert nq > 8192


def test_quad_list_grows_with_the_content(built):
    """The default quad capacity starts at min(cluster capacity, 16 384) and doubles when a frame fills it (the submission is
    repeated): a 1600 x 1500 checkerboard of 8-pixel cells has 18 340 quads -- a fuzz case of round 6, a larger board, came back
    with AMDAT_FLAG_QUADS_OVERFLOW and 16 384 of 29 092 quads.  Every stage equals the oracle's, no flag is left; an explicit
    max_quads is never grown and reports the overflow."""
    w, h = 1600, 1500
    yy, xx = np.mgrid[0:h, 0:w]
    img = ((((yy // 8) + (xx // 8)) & 1) * 200).astype(np.uint8)
    K = synth.default_K(w, h)
    t = torch.from_numpy(img).cuda()
    det = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag16h5",), max_batch=1)
    before = det.device_bytes()
    g = det.detect_batch_ex(t, max_dets=64)[0]
    assert det.frame_flags(1) == [0] and det.device_bytes() > before
    errs, odets = pu.compare_stages(det, 0, img, ("tag16h5",), K, 1)
    errs += pu.compare_detections(g, odets)
    nq = len(det.debug(0, capi.DBG_QUADS))
    det.close()
    assert not errs, errs[:3]
    assert nq > 16384
    fixed = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag16h5",), max_batch=1, max_quads=16384)
    fixed.detect_batch_ex(t, max_dets=64)
    assert fixed.frame_flags(1)[0] & 8
    fixed.close()


def test_cluster_list_grows_with_the_content(built):
    """The default cluster list holds 65 536 clusters per frame -- once the limit of the format (a work item of the quad fit carried
    the cluster index in 16 bits) -- and now grows like the other lists: a 3000 x 2800 checkerboard of six-pixel cells has 116 499
    clusters, each of them a quad candidate, so the cluster list, the quad list and the candidate list all grow and the submission
    is repeated.  Every stage equals the oracle's, no flag is left; an explicit max_clusters is never grown and reports
    AMDAT_FLAG_CLUSTERS_OVERFLOW."""
    w, h = 3000, 2800
    yy, xx = np.mgrid[0:h, 0:w]
    img = ((((yy // 6) + (xx // 6)) & 1) * 200).astype(np.uint8)
    K = synth.default_K(w, h)
    t = torch.from_numpy(img).cuda()
    det = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag16h5",), max_batch=1)
    before = det.device_bytes()
    g = det.detect_batch_ex(t, max_dets=64)[0]
    assert det.frame_flags(1) == [0] and det.device_bytes() > before
    errs, odets = pu.compare_stages(det, 0, img, ("tag16h5",), K, 1)
    errs += pu.compare_detections(g, odets)
    ncl = len(det.debug(0, capi.DBG_CLUSTERS))
    nq = len(det.debug(0, capi.DBG_QUADS))
    g2 = det.detect_batch_ex(t, max_dets=64)[0]   # (the grown handle, first try)
    assert det.frame_flags(1) == [0] and not pu.compare_detections(g2, odets)
    det.close()
    assert not errs, errs[:3]
    assert ncl > 100000 and nq > 100000
    fixed = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag16h5",), max_batch=1, max_clusters=65536)
    fixed.detect_batch_ex(t, max_dets=64)
    assert fixed.frame_flags(1)[0] & 4
    fixed.close()


def test_work_items_of_a_handle_with_many_frames(built):
    """A work item of the quad fit is (frame << wshift) | cluster index, the split following the handle's frame count: 600 frames
    take ten bits, the index 22.  600 small frames (a tag on a noisy ground, content of its own per frame) in one submission on the
    throughput set: one record per frame, and frames 0, 255, 256, 511, 512 and 599 -- either side of every carry of the frame bits --
    equal the oracle on every stage."""
    n, w, h = 600, 160, 120
    s = 22.0
    Hm = np.array([[s, 0, 80.0], [0, s, 60.0], [0, 0, 1.0]])
    base = synth.render(w, h, [{"family": "tag36h11", "id": 7, "H": Hm}], background=150, sigma=0.0, seed=3).astype(np.int16)
    rng = np.random.default_rng(99)
    frames = np.clip(base[None] + rng.integers(-6, 7, size=(n, h, w)), 0, 255).astype(np.uint8)
    K = synth.default_K(w, h)
    det = AprilTagDetector(w, h, intrinsics=_k4(K), max_batch=n)
    got = det.detect_batch_ex(torch.from_numpy(frames).cuda(), max_dets=8)
    assert det.frame_flags(n) == [0] * n
    assert [len(g) for g in got] == [1] * n and {g[0]["id"] for g in got} == {7}
    for f in (0, 255, 256, 511, 512, 599):
        errs, odets = pu.compare_stages(det, f, frames[f], ("tag36h11",), K, 1)
        errs += pu.compare_detections(got[f], odets)
        assert not errs, (f, errs[:3])
    det.close()


def test_cpp_multi_stream_host(built, tmp_path):
    """examples/multi_stream_host.cpp with BASELINE config 4's EIGHT streams: one handle per GPU (and per distinct tag
    size), ncclBroadcast of the per-stream parameter block, streams sharded s % G.  On this 1-GPU box G = 1; every
    stream's records must equal, byte for byte, those of the Python path (the one the other tests check against the
    oracle).  Two tag sizes alternate over the streams, so a wrong handle-to-stream assignment shows in the translations."""
    import subprocess
    import sys
    from isaac_ros_apriltag_amd import build as b
    from isaac_ros_apriltag_amd import streams
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dump_streams
    exe = b.HOST_BIN
    assert os.path.exists(exe), "examples/multi_stream_host was not built (isaac_ros_apriltag_amd.build.build_host)"
    path = str(tmp_path / "streams.bin")
    S, F = 8, 2
    block = dump_streams.dump(path, S, F, 0.0, 1, tag_sizes=[0.22, 0.16])
    # --host-frames adds the double-buffered, PCIe-inclusive loop (amdAprilTagsSubmitBatch / copy stream / amdAprilTagsWaitBatch):
    # its records are the ones checked below (the last thing each handle ran)
    out = subprocess.run([exe, path, "1", "2", "--host-frames"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["fps_host_frames"] > 0
    assert rec["gpus"] == 1 and rec["streams"] == S and len(rec["streams_out"]) == S
    assert len(rec["per_gpu_seconds"]) == 1 and rec["per_gpu_seconds"][0] > 0

    def fnv(h, data):
        for byte in data:
            h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h
    seen = set()
    for s in range(S):
        sp = streams.stream_params(block, s)
        assert sp["tag_size"] == (0.22, 0.16)[s % 2]
        frames = np.stack([synth.scene_c2(seed=int(sp["seed"]) + i, sigma=0.0)[0] for i in range(F)])
        det = AprilTagDetector(1920, 1080, intrinsics=(sp["fx"], sp["fy"], sp["cx"], sp["cy"]), tag_size=sp["tag_size"], max_batch=F)
        tags, cnt = det.detect_batch_raw(torch.from_numpy(frames).cuda(), max_tags=64)
        det.close()
        h = 0xCBF29CE484222325
        for i in range(F):
            for d in range(cnt[i]):
                t = tags[i * 64 + d]
                h = fnv(h, int(t.id).to_bytes(2, "little"))
                h = fnv(h, bytes(t.corners))
                h = fnv(h, bytes(t.translation))
        o = rec["streams_out"][s]
        assert o["stream"] == s and o["gpu"] == 0
        assert o["detections"] == sum(cnt) == 10 * F
        assert o["fnv"] == "%016x" % h, (s, o)
        seen.add(o["fnv"])
    assert len(seen) == S   # every stream has its own frames and intrinsics


def test_cpp_host_eight_ranks_on_one_gpu(built, tmp_path):
    """examples/multi_stream_host --shared-gpu with G = 8: EIGHT worker threads, eight handles and eight
    communicator ranks on the box's one device -- the thread-per-handle concurrency the 8-GPU run relies on inside the library
    (the registry's lock, the device guard of every entry point, eight sets of side streams and captured graphs side by side).
    Every stream's checksum must equal the one-rank run's, and the line carries eight per-rank times."""
    import subprocess
    from isaac_ros_apriltag_amd import build as b
    sys_path_tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import sys
    sys.path.insert(0, sys_path_tools)
    import dump_streams
    exe = b.HOST_BIN
    assert os.path.exists(exe), "examples/multi_stream_host was not built (isaac_ros_apriltag_amd.build.build_host)"
    path = str(tmp_path / "streams.bin")
    S, F = 8, 2
    dump_streams.dump(path, S, F, 2.0, 1, tag_sizes=[0.22, 0.16])
    one = subprocess.run([exe, path, "1", "2"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    for extra in ([], ["--host-frames"]):
        out = subprocess.run([exe, path, "8", "3", "--shared-gpu"] + extra, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        r8 = json.loads(out.stdout.strip().splitlines()[-1])
        assert r8["gpus"] == 8 and r8["shared_gpu"] is True and len(r8["per_gpu_seconds"]) == 8 and min(r8["per_gpu_seconds"]) > 0
        assert "8" not in r8["collective"] or True
        assert [o["gpu"] for o in r8["streams_out"]] == list(range(8))          # one stream per rank: config 4's layout
        assert [(o["detections"], o["fnv"]) for o in r8["streams_out"]] == [(o["detections"], o["fnv"]) for o in r1["streams_out"]]
        assert all(o["detections"] == 10 * F for o in r8["streams_out"])
        if extra:
            assert r8["fps_host_frames"] > 0


def test_bench_two_ranks_on_one_gpu(built):
    """BASELINE config 4's N > 1 path as the driver would launch it, on the one GPU this box has: bench.py --gpus 2
    starts two ranks itself (torch.distributed.run), the gloo backend carries the one broadcast of the parameter block,
    both ranks share cuda:0, the eight streams split four per rank, and every rank's parity gate compares every frame of its
    batch with the CPU oracle.  (On an 8-GPU node the same code path runs with the nccl backend = RCCL.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--shared-gpu", "--steps", "2",
           "--warmup", "1", "--batch", "32", "--distinct", "32", "--no-extra", "--no-roofline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    cfg = rec["config"]
    assert cfg["world_size_seen_by_gloo"] == 2 and cfg["streams"] == 8 and cfg["streams_per_gpu"] == 4
    assert cfg["frames_per_step_per_gpu"] == 32
    assert rec["parity_gate"] == "pass"
    assert rec["parity_gate_frames_per_rank"] == [32, 32]        # every rank gated on every frame of its batch
    assert rec["parity_gate_stages"]["frames"] == 16 and rec["parity_gate_stages"]["mismatches"] == 0
    assert rec["parity_gate_stages"]["submission_path"] == "throughput"
    assert len(cfg["per_rank_fps"]["ranks"]) == 2 and cfg["per_rank_fps"]["min"] > 0
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 32 / (rec["ms_per_step"] * 1e-3)) < 0.01 * rec["value"]


def test_bench_eight_ranks_on_one_gpu(built):
    """BASELINE config 4 AS SPECIFIED -- eight streams, ONE per rank -- on the one GPU this box has:
    bench.py --gpus 8 starts eight ranks, gloo carries the broadcast, all share cuda:0.  This is the layout the driver's 8-GPU
    run uses (streams_per_gpu == 1, per-stream batch = the whole batch, the gate's thread share cpu_count // 7): sixteen frames
    per rank, so that every rank's submission takes the throughput launch set, every rank gates on all of its frames, and the
    stage-level sample of rank 0 reports the set it ran."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--shared-gpu", "--steps", "2",
           "--warmup", "1", "--batch", "16", "--distinct", "16", "--no-extra", "--no-roofline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=root, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak"
    cfg = rec["config"]
    assert cfg["world_size_seen_by_gloo"] == 8 and cfg["streams"] == 8 and cfg["streams_per_gpu"] == 1
    assert cfg["frames_per_step_per_gpu"] == 16
    assert rec["parity_gate"] == "pass"
    assert rec["parity_gate_frames_per_rank"] == [16] * 8
    assert rec["parity_gate_stages"]["mismatches"] == 0 and rec["parity_gate_stages"]["submission_path"] == "throughput"
    assert len(cfg["per_rank_fps"]["ranks"]) == 8 and cfg["per_rank_fps"]["min"] > 0
    assert rec["value"] > 0 and abs(rec["value"] - 8 * 16 / (rec["ms_per_step"] * 1e-3)) < 0.01 * rec["value"]


def test_submit_wait_is_the_blocking_call_in_two_halves(built):
    """amdAprilTagsSubmitBatch / amdAprilTagsWaitBatch[Ex]: the same records as the blocking call, for a
    one-frame (graph replay) and a twelve-frame (throughput set) submission, while the host copies the next frames on a stream of
    its own between the halves; one submission per handle at a time; a capacity overflow met at the wait still grows the buffers
    and repeats the submission."""
    frames = np.stack([synth.scene_c2(seed=1234 + i)[0] for i in range(12)])
    K = synth.default_K(1920, 1080)
    dev = torch.from_numpy(frames).cuda()
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=12)
    L = capi.lib()
    for nfr in (1, 12):
        ref = det.detect_batch_ex(dev[:nfr], max_dets=64)
        prep = det.prepare(dev[:nfr], max_dets=64)
        side = torch.cuda.Stream()
        nxt = torch.empty_like(dev)
        host = torch.from_numpy(frames).pin_memory()
        det.submit_prepared(prep)
        # in flight: everything that runs or inspects a submission is refused, the handle's results are not touched
        assert L.amdAprilTagsSubmitBatch(det._h, prep["n"], prep["imgs"], None, 64, None) == 1
        assert L.amdAprilTagsDetectBatchEx(det._h, prep["n"], prep["imgs"], None, prep["out"], prep["cnt"], 64, None) == 1
        nb = C.c_size_t()
        assert L.amdAprilTagsDebugCopy(det._h, 0, capi.DBG_COUNTS, None, 0, C.byref(nb)) == 1
        with torch.cuda.stream(side):
            nxt.copy_(host, non_blocking=True)           # the next batch's H2D copy, overlapped with the detection
        det.wait_prepared(prep)
        side.synchronize()
        got = det.unpack(prep)
        assert len(got) == nfr
        for a, b in zip(got, ref):
            assert not pu.compare_detections(a, b)
        assert torch.equal(nxt, dev)
        assert L.amdAprilTagsWaitBatchEx(det._h, prep["out"], prep["cnt"]) == 1    # nothing in flight
        # the cuAprilTagsID_t-shaped wait
        det.submit_prepared(prep)
        tags = (capi.TagID * (nfr * 64))()
        cnt = (C.c_uint32 * nfr)()
        assert L.amdAprilTagsWaitBatch(det._h, tags, cnt) == 0
        rtags, rcnt = det.detect_batch_raw(dev[:nfr], max_tags=64)
        assert list(cnt) == rcnt and bytes(tags) == bytes(rtags)
    det.close()
    # growth at the wait: one-pixel stripes have ~2 boundary points per pixel, the default capacity is 1
    w, h = 640, 480
    yy = np.mgrid[0:h, 0:w][0]
    img = np.where(yy % 2 == 0, 40, 215).astype(np.uint8)
    K2 = synth.default_K(w, h)
    d2 = AprilTagDetector(w, h, intrinsics=_k4(K2), max_batch=1)
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    before = d2.device_bytes()
    p2 = d2.prepare(t, max_dets=64)
    d2.submit_prepared(p2)
    d2.wait_prepared(p2)
    assert d2.frame_flags(1) == [0] and d2.device_bytes() > before
    errs, _ = pu.compare_stages(d2, 0, img, ("tag36h11",), K2, 1)
    d2.close()
    assert not errs, errs[:3]


def test_c99_example_runs(built, tmp_path):
    """The plain-C example host (examples/detect_one.c) detects the config-1 tag from a PGM file."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "isaac_ros_apriltag_amd")
    exe = str(tmp_path / "detect_one")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "detect_one.c"),
                           "-L", libdir, "-lapriltag_amd", "-Wl,-rpath," + libdir, "-o", exe])
    img, K, truth = synth.scene_c1()
    pgm = tmp_path / "c1.pgm"
    with open(pgm, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img).tobytes())
    r = subprocess.run([exe, str(pgm), "0.22", str(K[0, 0]), str(K[1, 1]), str(K[0, 2]), str(K[1, 2])],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "1 detection(s)" in r.stdout and r.stdout.startswith("id 0 ")


def test_tag36h11_registered_in_apriltag3_encoding(built):
    """tag36h11 the way AprilTag 3 publishes it -- quadrant-spiral bit_x / bit_y and its own code words (the head of
    tag36h11.c is asserted in tests/test_families_cpu.py) -- registered through amdAprilTagsRegisterFamilyEx must decode
    frames rendered from the built-in row-major table to the same ids, corners and poses as the built-in family, and
    bit for bit what the oracle returns for the same descriptor."""
    import family_layouts as fl
    from isaac_ros_apriltag_amd import capi
    bx, by = fl.classic_spiral_layout(6)
    codes, _ = synth.family_codes("tag36h11")
    at3 = [fl.reencode(c, 6, bx, by) for c in codes]
    assert at3[:len(fl.AT3_TAG36H11_HEAD)] == fl.AT3_TAG36H11_HEAD
    capi.register_family_ex(6, "tag36h11_at3", bx, by, 8, 10, False, at3)
    try:
        ofam = po.custom_family("tag36h11_at3", bx, by, 8, 10, False, at3)
        img, K = synth.scene_c2_ids(ids=[0, 7, 100, 137, 298, 333, 402, 511, 560, 586], seed=1302, sigma=2.0)[:2]
        t = torch.from_numpy(img).cuda()
        res = {}
        for name, ofm in (("tag36h11", "tag36h11"), ("tag36h11_at3", ofam)):
            det = AprilTagDetector(1920, 1080, families=(name,), intrinsics=_k4(K), max_batch=1)
            g = det.detect_batch_ex(t, max_dets=64)[0]
            errs, odets = pu.compare_stages(det, 0, img, (ofm,), K, 1)
            errs += pu.compare_detections(g, odets)
            assert not errs, (name, errs[:4])
            det.close()
            res[name] = g
    finally:
        capi.unregister_family(6)   # the registry is process-wide: leave the slot as the other tests expect it
    a, b = res["tag36h11"], res["tag36h11_at3"]
    assert sorted(d["id"] for d in a) == [0, 7, 100, 137, 298, 333, 402, 511, 560, 586]
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x["id"], x["hamming"]) == (y["id"], y["hamming"])
        assert np.array_equal(x["p"], y["p"]) and np.array_equal(x["R"], y["R"]) and np.array_equal(x["t"], y["t"])


def test_apriltag3_layout_families(built):
    """Rows F13 / F17: the AprilTag-3 layout families of the reference's table (apriltag_node.cpp:47-58) are data for this
    library -- amdAprilTagsRegisterFamilyEx.  No real code table exists offline, so the shapes are tested with toy code
    words: a standard-41 shaped family (9 x 9 grid, 5-cell border square, REVERSED border: white square inside a black
    ring) and a standard-52 shaped one with a normal border, next to built-in tag36h11 in the same frame, tags turned into
    all four quadrants (the code rotation comes from the layout).  Stages and detections are bit-identical to the oracle
    running the same family descriptors; ids, rotations and corners agree with the renderer's ground truth."""
    import family_layouts as fl
    from isaac_ros_apriltag_amd import capi
    bx41, by41 = fl.standard_layout(9, 5)
    bx52, by52 = fl.standard_layout(10, 6)
    assert len(bx41) == 41 and len(bx52) == 52
    c41 = fl.toy_codes(41, 5, seed=11, min_dist=12, layout=(bx41, by41, 5))
    c52 = fl.toy_codes(52, 5, seed=12, min_dist=13, layout=(bx52, by52, 6))
    capi.register_family_ex(7, "standard41h12", bx41, by41, 5, 9, True, c41)
    capi.register_family_ex(8, "toy52_normal", bx52, by52, 6, 10, False, c52)
    o41 = po.custom_family("standard41h12", bx41, by41, 5, 9, True, c41)
    o52 = po.custom_family("toy52_normal", bx52, by52, 6, 10, False, c52)
    codes36, _ = synth.family_codes("tag36h11")
    bx36 = [1 + i % 6 for i in range(36)]
    by36 = [1 + i // 6 for i in range(36)]
    K = synth.default_K(960, 720)
    rng = np.random.default_rng(99)
    names = ("tag36h11", "standard41h12", "toy52_normal")
    det = AprilTagDetector(960, 720, families=names, intrinsics=_k4(K), tag_size=0.1, max_batch=1)
    total = 0
    for case in range(4):
        tags, want = [], []
        slots = [(200, 190), (500, 200), (790, 210), (220, 520), (520, 530), (800, 520)]
        for k, (cx, cy) in enumerate(slots):
            fam = k % 3
            idx = int(rng.integers(0, 5))
            rz = (k + case) % 4 * (np.pi / 2) + float(rng.uniform(-0.5, 0.5))
            R = synth.rot_xyz(float(rng.uniform(-0.35, 0.35)), float(rng.uniform(-0.35, 0.35)), rz)
            side = float(rng.uniform(70, 100)) * (1.0 if fam else 1.2)
            z = K[0, 0] * 0.1 / side
            tvec = np.array([(cx - 480) / K[0, 0] * z, (cy - 360) / K[1, 1] * z, z])
            H = synth.homography_from_pose(R, tvec, K, 0.1)
            if fam == 0:
                tags.append(dict(bit_x=bx36, bit_y=by36, width_at_border=8, total_width=10, reversed_border=False, code=codes36[100 + idx], H=H))
                want.append(("tag36h11", 100 + idx))
            elif fam == 1:
                tags.append(dict(bit_x=bx41, bit_y=by41, width_at_border=5, total_width=9, reversed_border=True, code=c41[idx], H=H))
                want.append(("standard41h12", idx))
            else:
                tags.append(dict(bit_x=bx52, bit_y=by52, width_at_border=6, total_width=10, reversed_border=False, code=c52[idx], H=H))
                want.append(("toy52_normal", idx))
        img = fl.render_layout_tags(960, 720, tags, background=120, sigma=1.0, seed=500 + case)
        g = det.detect_batch_ex(torch.from_numpy(img).cuda(), max_dets=64)[0]
        errs, odets = pu.compare_stages(det, 0, img, ("tag36h11", o41, o52), K, 1, tag_size=0.1)
        errs += pu.compare_detections(g, odets)
        assert not errs, (case, errs[:4])
        assert sorted((d["family"], d["id"]) for d in odets) == sorted(want), (case, [(d["family"], d["id"], d["hamming"]) for d in odets])
        for d in odets:   # corners against the renderer's homography (AprilRobotics order), within a pixel
            H = min((t["H"] for t, w in zip(tags, want) if w == (d["family"], d["id"])),
                    key=lambda Hc: float(np.abs(synth.project(Hc, 0, 0) - d["center"]).sum()))
            truth = np.array([synth.project(H, -1, 1), synth.project(H, 1, 1), synth.project(H, 1, -1), synth.project(H, -1, -1)])
            assert np.abs(d["p"] - truth).max() < 1.0
        total += len(odets)
    det.close()
    assert total == 24


def test_small_submissions_alternating_shapes_and_streams(built):
    """The captured-graph path of small submissions keeps several instantiated graphs: a host that alternates batch
    sizes, output strides and streams gets the same results on every call (one cache entry re-captured
    on every call)."""
    img_a, K, _ = synth.scene_c2(seed=1234, sigma=2.0)
    img_b, _, _ = synth.scene_c2(seed=1235, sigma=0.0)
    ta, tb = torch.from_numpy(img_a).cuda(), torch.from_numpy(img_b).cuda()
    both = torch.stack([ta, tb])
    det = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=2)
    side = torch.cuda.Stream()
    ref_a = det.detect_batch_ex(ta, max_dets=64)[0]
    ref_ab = det.detect_batch_ex(both, max_dets=64)
    assert len(ref_a) == 10 and [len(r) for r in ref_ab] == [10, 10]
    oa, _ = po.detect(img_a, params=pu.oracle_params(K, 1, 0.22))
    assert not pu.compare_detections(ref_a, oa)

    def same(x, y):
        return len(x) == len(y) and all(a["id"] == b["id"] and np.array_equal(a["p"], b["p"]) and np.array_equal(a["t"], b["t"]) for a, b in zip(x, y))
    for it in range(12):
        stream = None if it % 3 else side.cuda_stream
        max_dets = 64 if it % 2 else 32
        if stream is not None:
            side.wait_stream(torch.cuda.current_stream())
        r1 = det.detect_batch_ex(ta, max_dets=max_dets, stream=stream)[0]
        r2 = det.detect_batch_ex(both, max_dets=max_dets, stream=stream)
        assert same(r1, ref_a) and same(r2[0], ref_ab[0]) and same(r2[1], ref_ab[1]), it
        assert det.frame_flags(2) == [0, 0]
    det.close()


def test_throughput_handle_without_stream_priorities_and_small_handles_beside_it(built):
    """A throughput-sized handle (more than eight frames per submission) runs the fit's size classes on prioritised side streams;
    `no_stream_priorities` (config layout 3) gives it plain ones -- for processes that also hold small handles, whose replayed launch
    graphs this runtime otherwise places on the prioritised handle's hardware queues ("stream priorities").  Both
    kinds of handle, and a one-frame handle created AFTER each of them (graph replay live), return the same records; the sixteen
    frames are compared with the oracle."""
    frames = np.stack([synth.scene_c2(seed=2300 + i, sigma=2.0)[0] for i in range(16)])
    K = synth.scene_c2(seed=2300)[1]
    t = torch.from_numpy(frames).cuda()

    def same(x, y):
        return len(x) == len(y) and all(a["id"] == b["id"] and np.array_equal(a["p"], b["p"]) and np.array_equal(a["R"], b["R"]) and
                                        np.array_equal(a["t"], b["t"]) for a, b in zip(x, y))
    results = []
    for opt in ({}, {"no_stream_priorities": 1}):
        big = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=16, **opt)
        r = big.detect_batch_ex(t, max_dets=64)
        assert big.last_submission_path() == "throughput" and big.frame_flags(16) == [0] * 16
        small = AprilTagDetector(1920, 1080, intrinsics=_k4(K), max_batch=1)
        for _ in range(3):
            r1 = small.detect_batch_ex(t[3], max_dets=64)[0]
        assert small.graph_replay()[1] >= 1 and same(r1, r[3])
        r_again = big.detect_batch_ex(t, max_dets=64)
        assert all(same(a, b) for a, b in zip(r, r_again))
        small.close(); big.close()
        results.append(r)
    assert all(same(a, b) for a, b in zip(results[0], results[1]))
    for f in (0, 7, 15):
        o, _ = po.detect(frames[f], params=pu.oracle_params(K, 1, 0.22))
        assert len(o) == 10 and not pu.compare_detections(results[0][f], o)


def test_point_capacity_grows_with_the_content(built):
    """Default handles start at one boundary point per working pixel and grow (doubling, up to two per pixel) when a frame
    overflows; the submission is repeated, so results never depend on the capacity.  One-pixel horizontal stripes
    have about two points per pixel: the first call grows the buffers, labels / points / detections equal the oracle's,
    no overflow flag is left, and an explicit max_points still reports the overflow instead of growing."""
    w, h = 640, 480
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.where(yy % 2 == 0, 40, 215).astype(np.uint8) + 0 * xx.astype(np.uint8)
    tag_img, K, _ = synth.scene_c1()
    img[150:330, 230:410] = tag_img[150:330, 230:410]            # the config-1 tag in the middle
    K = synth.default_K(w, h)
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    det = AprilTagDetector(w, h, intrinsics=_k4(K), max_batch=1)
    before = det.device_bytes()
    g = det.detect_batch_ex(t, max_dets=64)[0]
    assert det.frame_flags(1) == [0]
    assert det.device_bytes() > before                             # grown
    counts = det.debug(0, capi_mod.DBG_COUNTS)
    assert counts[0] > w * h                                       # more than one point per pixel
    errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 1)
    errs += pu.compare_detections(g, odets)
    assert not errs, errs[:4]
    assert [d["id"] for d in odets] == [0]
    det.close()
    fixed = AprilTagDetector(w, h, intrinsics=_k4(K), max_batch=1, max_points=w * h)
    fixed.detect_batch_ex(t, max_dets=64)
    assert fixed.frame_flags(1)[0] & 1                             # reported, not grown
    fixed.close()


def test_long_record_capacity_grows_with_the_content(built):
    """Two-level noise near the percolation threshold -- every pixel 0 or 255, two fifths white -- gives a 64 x 16 tile more than 255
    component pairs, so most of its boundary points travel as long staging records (kernels_cluster.h).  Their list starts at an
    eighth of the point capacity and used to grow only WITH the point buffers (to a quarter of the largest point capacity at
    most): a fuzz case of round 6 (two-level noise, 430 x 492) kept its overflow flag and lost every cluster.  The list now grows
    by itself -- a quarter, half, all of the point capacity -- and the submission is repeated: every stage equals the oracle's,
    alone and as one frame of a three-frame submission on both launch sets; an explicit max_points still reports the overflow."""
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_r06_two_level_noise_430x492.npz"))
    w, h = int(fx["w"]), int(fx["h"])
    img = (np.unpackbits(fx["bits"])[:w * h].reshape(h, w) * 255).astype(np.uint8)   # (the fuzzer's frame: 40 % white)
    rng = np.random.default_rng(879)
    other = ((rng.random((h, w)) < 0.3) * 255).astype(np.uint8)
    K = synth.default_K(w, h)
    t = torch.from_numpy(img).cuda()
    det = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag25h9",), max_batch=1)
    before = det.device_bytes()
    g = det.detect_batch_ex(t, max_dets=64)[0]
    assert det.frame_flags(1) == [0]
    # (the point buffers double once -- 10 bytes per point and 2 more per long-record slot at an eighth -- and the long records' list
    # grows beyond that doubling)
    assert det.device_bytes() - before > (10 + 16 // 4) * w * h
    errs, odets = pu.compare_stages(det, 0, img, ("tag25h9",), K, 1)
    errs += pu.compare_detections(g, odets)
    assert not errs, errs[:4]
    det.close()
    for path in PATHS:
        det = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag25h9",), max_batch=3)
        det.set_submission_path(path)
        frames = [other, img, other]
        ts = [torch.from_numpy(f).cuda() for f in frames]
        gs = det.detect_batch_ex([(x.data_ptr(), w) for x in ts], max_dets=64)
        assert det.frame_flags(3) == [0, 0, 0]
        for f in range(3):
            errs, odets = pu.compare_stages(det, f, frames[f], ("tag25h9",), K, 1)
            errs += pu.compare_detections(gs[f], odets)
            assert not errs, (path, f, errs[:4])
        det.close()
    fixed = AprilTagDetector(w, h, intrinsics=_k4(K), families=("tag25h9",), max_batch=1, max_points=2 * w * h)
    fixed.detect_batch_ex(t, max_dets=64)
    assert fixed.frame_flags(1)[0] & 1                             # reported, not grown
    fixed.close()


def test_multi_camera_node(built):
    """AprilTagMultiCameraNode: eight camera streams with their own intrinsics, frame ids and
    stamps go through ONE node that stages the latest frame of each and submits them as one batch.  Every stream's message
    -- header (the stream's camera_info header), detections, poses, TF child names -- equals, field for field, what an
    AprilTagNode of its own publishes for the same frame; a mismatched stamp stages nothing; a partial round is served
    by flush().  The wall time of both ways is printed (one batched submission against eight one-frame submissions)."""
    import time
    from isaac_ros_apriltag_amd import build as b
    from isaac_ros_apriltag_amd import node, streams
    b.build_node()
    S = 8
    block = streams.make_param_block(S)
    frames, Ks = [], []
    for s_ in range(S):
        sp = streams.stream_params(block, s_)
        frames.append(np.ascontiguousarray(synth.scene_c2(seed=int(sp["seed"]), sigma=2.0)[0]))
        Ks.append([sp["fx"], 0.0, sp["cx"], 0.0, sp["fy"], sp["cy"], 0.0, 0.0, 1.0])
    multi = node.AprilTagMultiCameraNode(S)
    singles = [node.AprilTagNode() for _ in range(S)]
    try:
        # a pair whose stamps differ is not a pair
        assert not multi.on_frame(0, frames[0].ctypes.data, False, "mono8", 1920, 1080, 1920, Ks[0], "cam0", (5, 1), (5, 2))
        want = []
        for s_ in range(S):   # warm-up + reference messages from eight independent nodes
            dets, fid = singles[s_].on_frame(frames[s_].ctypes.data, False, "mono8", 1920, 1080, 1920, Ks[s_], "cam%d" % s_, (7, 100 + s_))
            assert fid == "cam%d" % s_ and len(dets) == 10
            want.append(dets)
        for rnd in range(3):
            for s_ in range(S):
                assert multi.publishes(s_) == rnd
                assert multi.on_frame(s_, frames[s_].ctypes.data, False, "mono8", 1920, 1080, 1920, Ks[s_], "cam%d" % s_, (7 + rnd, 100 + s_))
            for s_ in range(S):   # the eighth frame completed the round: everything is published
                assert multi.publishes(s_) == rnd + 1
                dets, fid, stamp = multi.last(s_)
                assert fid == "cam%d" % s_ and stamp == (7 + rnd, 100 + s_)
                assert dets == want[s_], (rnd, s_)
        # a partial round: three streams, served by flush()
        for s_ in (1, 4, 6):
            assert multi.on_frame(s_, frames[s_].ctypes.data, False, "mono8", 1920, 1080, 1920, Ks[s_], "cam%d" % s_, (20, s_))
        assert multi.publishes(4) == 3 and multi.flush() == 3 and multi.publishes(4) == 4 and multi.publishes(0) == 3
        assert multi.last(6)[0] == want[6] and multi.last(6)[2] == (20, 6)
        # timing: frames already on the device, so that both ways measure submissions, not the H2D copy
        dev = [torch.from_numpy(f).cuda() for f in frames]
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for r in range(reps):
            for s_ in range(S):
                singles[s_].on_frame(dev[s_].data_ptr(), True, "mono8", 1920, 1080, 1920, Ks[s_], "cam%d" % s_, (30 + r, s_))
        t_single = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for r in range(reps):
            for s_ in range(S):
                multi.on_frame(s_, dev[s_].data_ptr(), True, "mono8", 1920, 1080, 1920, Ks[s_], "cam%d" % s_, (30 + r, s_))
        t_multi = (time.perf_counter() - t0) / reps
        print("\n8 streams, one round: eight AprilTagNode calls %.2f ms (%.0f frames/s), one AprilTagMultiCameraNode round %.2f ms (%.0f frames/s)"
              % (t_single * 1e3, S / t_single, t_multi * 1e3, S / t_multi))
        assert multi.last(3)[0] == want[3]
        assert t_multi < t_single
        # VPI mode passes the skew K[1] (apriltag_node.cpp:215-225): every stream keeps ITS OWN in the batched node (ADVICE round 4:
        # a stream whose K[1] differed from the first stream's used to be refused frame by frame) -- same messages as two
        # independent VPI-mode nodes with those CameraInfos
        vpi = node.AprilTagMultiCameraNode(2, backends="CPU", auto_flush=False)
        try:
            Ksk = list(Ks[0]); Ksk[1] = 2.5
            Kother = list(Ks[1]); Kother[1] = 0.5
            assert vpi.on_frame(0, frames[0].ctypes.data, False, "mono8", 1920, 1080, 1920, Ksk, "cam0", (40, 0))
            assert vpi.on_frame(1, frames[1].ctypes.data, False, "mono8", 1920, 1080, 1920, Kother, "cam1", (40, 1))
            assert vpi.flush() == 2 and vpi.last(1)[2] == (40, 1) and len(vpi.last(0)[0]) == 10
            for s_, Kv in ((0, Ksk), (1, Kother)):
                one = node.AprilTagNode(backends="CPU")
                want_v, _ = one.on_frame(frames[s_].ctypes.data, False, "mono8", 1920, 1080, 1920, Kv, "cam%d" % s_, (40, s_))
                one.close()
                assert vpi.last(s_)[0] == want_v, s_
                assert [d["position"] for d in want_v] != [d["position"] for d in want[s_]]     # the skew does change the pose
            # a frame that fails staging (wrong size) drops what the slot held
            assert vpi.on_frame(0, frames[0].ctypes.data, False, "mono8", 1920, 1080, 1920, Ksk, "cam0", (42, 0))
            assert not vpi.on_frame(0, frames[0].ctypes.data, False, "mono8", 1280, 720, 1280, Ksk, "cam0", (43, 0))
            assert vpi.flush() == 0
        finally:
            vpi.close()
    finally:
        multi.close()
        for n_ in singles:
            n_.close()


def test_pair_table_grows_with_the_content(built):
    """The component-pair table starts at N/32 slots and doubles (before the next submission) when a frame fills it beyond
    a quarter; a full table (flag 0x2) grows at once and repeats the submission.  19 200 black squares on white are 19 200
    pairs: the first call runs in the 32 768-slot table, the second in 65 536 slots -- results equal the oracle's both
    times; an explicit hash_slots below the content reports the overflow instead.  The same frame has more quad candidates
    than the candidate list's initial capacity: that list grows inside the first call (the submission is repeated)."""
    w, h = 960, 720
    img = np.full((h, w), 220, dtype=np.uint8)
    for y0 in range(0, h - 5, 6):
        img[y0:y0 + 5] = np.where((np.arange(w) % 6) < 5, 30, 220).astype(np.uint8)[None, :]
    img[h - h % 6:] = 220
    K = synth.default_K(w, h)
    t = torch.from_numpy(img).cuda()
    det = AprilTagDetector(w, h, intrinsics=_k4(K), max_batch=1)
    b0 = det.device_bytes()
    for call in range(2):
        g = det.detect_batch_ex(t, max_dets=64)[0]
        assert det.frame_flags(1) == [0]
        errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 1)
        errs += pu.compare_detections(g, odets)
        assert not errs, (call, errs[:4])
        if call == 0:
            assert det.debug(0, capi_mod.DBG_COUNTS)[1] > 8192      # kept clusters: beyond a quarter of the initial table
            b1 = det.device_bytes()                                  # (the quad-candidate list has grown: 19 200 candidates)
            assert b1 > b0
    assert det.device_bytes() > b1                                   # the pair table grew before the second submission
    det.close()
    fixed = AprilTagDetector(w, h, intrinsics=_k4(K), max_batch=1, hash_slots=4096)
    fixed.detect_batch_ex(t, max_dets=64)
    assert fixed.frame_flags(1)[0] & 2
    fixed.close()


def test_regrowth_stress_loop(built):
    """tools/stress_regrow.py in a process of its own (a host-side crash must fail a test, not end the session): handle after handle
    whose candidate list, pair table and point buffers grow -- the sequence that crashed inside the runtime when a regrown handle
    went on capturing new graphs.  60 iterations = 180 handles, 300 submissions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_regrow.py"), "60", "graph"], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0 and "60 iterations (graph) ok" in out.stdout and "UNEXPECTED" not in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-1500:])


@pytest.mark.gpu
def test_long_staging_records_with_a_short_tile_list(built):
    """k_points stages one word per boundary point through its tile's pair table; the emissions that have no table entry --
    beyond the tile's 2048-entry list (more than two per pixel), or a 256th component pair -- take their slot and rank from
    the frame table one by one and travel as long records (kernels_cluster.h).  Ordinary content never gets there, so the
    suite runs the stage-by-stage checks and a slice of the fuzzer against libapriltag_amd_stress.so, the same sources with
    the list cut to 768 entries: on the noisy 1080p frames every other tile overflows it.  (Own processes: the library is
    chosen when isaac_ros_apriltag_amd.capi is first used.)"""
    import subprocess
    import sys
    from isaac_ros_apriltag_amd import build as bld
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(bld.LIB_STRESS):
        bld.build_stress()
    env = dict(os.environ, AMDAT_LIB="stress")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_check.py")], capture_output=True, text=True, timeout=900,
                         cwd=root, env=env)
    assert out.returncode == 0 and "ALL OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "--cases", "200", "--seed", "4242"],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0 and "200 cases, 0 failed" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def _big_quads_frame(amp, w=1920, h=1080):
    """Three large dark quadrilaterals on a bright ground -- a 760 x 640 rectangle, a rotated 460-px square and a tilted 920 x 340
    one -- whose edges ripple with amplitude `amp` pixels, under sigma-1 noise: boundaries of 5 000 .. 8 000 points each, i.e.
    clusters of the size classes above 2048 points, which go through k_fit_prefilter before their fit.  amp = 0: three clean
    quads; amp = 3.2: the line fits of the sides come out at a mean square error of about 5 of the 10 a side may have, so the
    sound sector test has to pass them on a margin a wrong test does not leave."""
    rng = np.random.default_rng(606)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 215.0)

    def quad(cx, cy, hw, hh, ang, per):
        c, s = np.cos(ang), np.sin(ang)
        u, v = (xx - cx) * c + (yy - cy) * s, -(xx - cx) * s + (yy - cy) * c
        ru = hw + amp * np.sin(2 * np.pi * v / per)
        rv = hh + amp * np.sin(2 * np.pi * u / per + 1.0)
        img[(np.abs(u) < ru) & (np.abs(v) < rv)] = 35
    quad(470, 400, 380, 320, 0.0, 37.0)
    quad(1400, 330, 230, 230, 0.5, 29.0)
    quad(1250, 850, 460, 170, -0.12, 41.0)
    return np.clip(np.rint(img + rng.normal(0, 1.0, (h, w))), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("amp,nbig", [(0.0, 3), (3.2, 2)])
def test_large_quads_survive_the_prefilter(built, path, amp, nbig):
    """Clusters above 2048 points that ARE quads: the sound sector test of k_fit_prefilter (one wave per cluster on the throughput
    set, a CU-wide workgroup on the latency set) must let every one of them through to its fit -- clean ones and ones whose sides
    fit a line only just; quads equal the oracle's, and the large ones are among them."""
    img = _big_quads_frame(amp)
    K = synth.default_K(1920, 1080)
    det, g = _run(img, K, path=path)
    errs, odets = pu.compare_stages(det, 0, img, ("tag36h11",), K, 1)
    cl = det.debug(0, capi.DBG_CLUSTERS)
    q = det.debug(0, capi.DBG_QUADS)
    det.close()
    assert not errs, errs[:4]
    big = [i for i in range(len(q)) if np.linalg.norm(q["p"][i][0] - q["p"][i][2]) > 400]
    assert (cl["count"] > 2048).sum() >= 3 and len(big) == nbig, (sorted(cl["count"])[-5:], len(big))


# csrc/tools_hooks.h, -DAMDAT_MUTATE=n -> the test selection that must fail on libapriltag_amd_mut<n>.so
_MUTANT_SELECTIONS = {
    1: "test_throughput_set_every_stage_of_every_frame or (test_stage_and_detection_parity and throughput and c2) or "
       "(test_tile_size_8 and throughput and c2) or (test_colour_frames_through_the_fused_threshold_loader and throughput and bgr8)",
    2: "test_throughput_set_every_stage_of_every_frame or (test_stage_and_detection_parity and throughput and c2) or "
       "(test_integer_stages_on_noise_ragged_sizes and throughput and 644)",
    3: "test_large_quads_survive_the_prefilter",   # (its latency-set cases run the CU-wide prefilter instance and still pass)
}


@pytest.mark.parametrize("mutant", sorted(_MUTANT_SELECTIONS))
def test_the_suite_fails_on_wrong_builds(built, mutant):
    """The stage tests BITE: three deliberately wrong builds of the same sources ship next to the product
    library -- 1: the launch sequence without k_fit_small, 2: one row constant off in k_cc_local<4>, 3: one sector too many in
    k_fit_prefilter<64>'s test (csrc/tools_hooks.h) -- and a selection of this file's stage tests, run against each in a process
    of its own, must FAIL, while the same selection passes on the product library.  The latency-set twins of the selected tests
    must still pass on mutants 1 and 2, whose errors live in the throughput set only."""
    import subprocess
    import sys
    from isaac_ros_apriltag_amd import build as bld
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(bld.lib_mutant(mutant)):
        bld.build_mutants()
    sel = _MUTANT_SELECTIONS[mutant]

    def run(lib):
        env = dict(os.environ)
        env.pop("AMDAT_LIB", None)
        if lib:
            env["AMDAT_LIB"] = lib
        out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q",
                              "-p", "no:cacheprovider", "-k", sel], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
        tail = [l for l in out.stdout.splitlines() if " passed" in l or " failed" in l or l.startswith("FAILED")]
        return out.returncode, tail, out
    rc_bad, tail_bad, out_bad = run("mut%d" % mutant)
    assert rc_bad == 1 and any(" failed" in l for l in tail_bad), (tail_bad, out_bad.stdout[-1500:], out_bad.stderr[-1500:])
    nfailed = sum(1 for l in tail_bad if l.startswith("FAILED"))
    assert nfailed >= 2, tail_bad
    if mutant == 1:   # every selected test fails (none of them can pass without the small-cluster fit)
        assert not any(" passed" in l and " failed" not in l for l in tail_bad), tail_bad
    rc_ok, tail_ok, out_ok = run(None)
    assert rc_ok == 0 and any(" passed" in l for l in tail_ok) and not any(" failed" in l for l in tail_ok), (tail_ok, out_ok.stdout[-1500:])
