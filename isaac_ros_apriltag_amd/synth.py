"""Deterministic synthetic AprilTag frames with analytic ground truth.

Replaces the reference's Git-LFS image fixture (isaac_ros_apriltag/test/test_cases/apriltag0/image.png
is an LFS pointer) and generates the inputs of every BASELINE.json config.  The pixels are rendered
by csrc/synth_render.c (host C, seeded counter-based noise, integer arithmetic) so the build
container and the GPU box produce identical bytes.

Conventions (AprilRobotics): tag coordinates x right, y down, black-border outer edge = [-1,1]^2;
pose = tag frame in the camera optical frame (z forward, y down); `size` = border edge length [m].
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "synth_render.c")
_LIB_PATH = os.path.join(_HERE, "libapriltag_synth.so")


class _Tag(C.Structure):
    _fields_ = [("code", C.c_uint64), ("d", C.c_int32), ("pad", C.c_int32), ("H", C.c_double * 9)]


def build():
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu99", "-ffp-contract=off", "-shared", "-o", _LIB_PATH,
                           _SRC, "-lm"])


_lib = None


def _get_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.synth_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                      C.POINTER(_Tag), C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.synth_render.restype = C.c_int
        _lib.synth_family_codes.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.synth_family_codes.restype = C.POINTER(C.c_uint64)
    return _lib


def family_codes(name):
    n, d = C.c_int(), C.c_int()
    ptr = _get_lib().synth_family_codes(name.encode(), C.byref(n), C.byref(d))
    if n.value == 0:
        raise ValueError("unknown family %r" % name)
    return [int(ptr[i]) for i in range(n.value)], d.value


class _Rng:
    """splitmix64 stream (pure Python ints) so scene layouts are identical everywhere."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        x = self.s
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return x ^ (x >> 31)

    def uniform(self, a, b):
        return a + (b - a) * (self.u64() >> 11) / float(1 << 53)


def homography_from_pose(R, t, K, size):
    """H mapping tag coords [-1,1]^2 to pixels for a tag of edge `size` with pose (R, t)."""
    R = np.asarray(R, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    M = np.stack([R[:, 0] * size / 2.0, R[:, 1] * size / 2.0, t], axis=1)
    H = np.asarray(K, dtype=np.float64) @ M
    return H / H[2, 2]


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def project(H, x, y):
    v = np.asarray(H) @ np.array([x, y, 1.0])
    return v[:2] / v[2]


def truth_from_H(family, tag_id, H, R=None, t=None):
    """Ground-truth record in AprilRobotics corner order: H(-1,1), H(1,1), H(1,-1), H(-1,-1)."""
    p = np.array([project(H, -1, 1), project(H, 1, 1), project(H, 1, -1), project(H, -1, -1)])
    return {"family": family, "id": tag_id, "H": np.asarray(H), "center": project(H, 0, 0), "p": p, "R": R, "t": t}


def render(width, height, tags, background=150, sigma=0.0, seed=0, ss=4, black=25, white=230, pitch=None):
    """tags: list of dicts {family, id, H}. Returns mono8 ndarray (height, width)."""
    lib = _get_lib()
    pitch = pitch or width
    buf = np.zeros((height, pitch), dtype=np.uint8)
    arr = (_Tag * max(len(tags), 1))()
    for i, tg in enumerate(tags):
        codes, d = family_codes(tg["family"])
        arr[i].code = tg["code"] if "code" in tg else codes[tg["id"]]
        arr[i].d = d
        for k, v in enumerate(np.asarray(tg["H"], dtype=np.float64).reshape(-1)):
            arr[i].H[k] = float(v)
    rc = lib.synth_render(buf.ctypes.data, width, height, pitch, int(background), int(round(sigma * 256)),
                          int(seed) & 0xFFFFFFFFFFFFFFFF, arr, len(tags), ss, black, white)
    if rc != 0:
        raise RuntimeError("synth_render failed")
    return buf[:, :width] if pitch != width else buf


# --------------------------------------------------------------------------------------------
# Scenes for the BASELINE.json configs
# --------------------------------------------------------------------------------------------
def default_K(width, height):
    return np.array([[1000.0, 0, width / 2.0], [0, 1000.0, height / 2.0], [0, 0, 1]])


def scene_pol_golden():
    """Re-synthesis of the reference's apriltag0 fixture: 1920x1080, tag36h11 id 0, rotated 180 deg
    in-plane, pose from isaac_ros_apriltag/test/isaac_ros_apriltag_pol_test.py:156-175 and K from
    test/test_cases/apriltag0/camera_info.json."""
    K = np.array([[434.943999, 0, 651.073921], [0, 431.741273, 441.878037], [0, 0, 1]])
    R = rot_xyz(0, 0, math.pi)
    R = np.round(R)  # exact 180 degrees
    t = np.array([0.255342, 0.098358, 0.403961])
    H = homography_from_pose(R, t, K, 0.22)
    tags = [{"family": "tag36h11", "id": 0, "H": H}]
    img = render(1920, 1080, tags, background=150, sigma=0.0, seed=0)
    return img, K, [truth_from_H("tag36h11", 0, H, R, t)]


def scene_c1():
    """Config 1: 640x480, one tag36h11 id 0, border side 120 px centred (320,240), no noise."""
    s = 60.0
    H = np.array([[s, 0, 320.0], [0, s, 240.0], [0, 0, 1.0]])
    tags = [{"family": "tag36h11", "id": 0, "H": H}]
    img = render(640, 480, tags, background=160, sigma=0.0, seed=1)
    return img, default_K(640, 480), [truth_from_H("tag36h11", 0, H)]


def _grid_scene(width, height, families_ids, cols, rows, seed, side_lo, side_hi, rot_max_deg, tilt_max_deg,
                sigma, size=0.22):
    rng = _Rng(seed)
    K = default_K(width, height)
    tags, truth = [], []
    cw, ch = width / cols, height / rows
    for idx, (fam, tid) in enumerate(families_ids):
        gx, gy = idx % cols, idx // cols
        side = rng.uniform(side_lo, side_hi)
        jx = rng.uniform(-0.12, 0.12) * cw
        jy = rng.uniform(-0.12, 0.12) * ch
        cxp, cyp = (gx + 0.5) * cw + jx, (gy + 0.5) * ch + jy
        rz = math.radians(rng.uniform(-rot_max_deg, rot_max_deg))
        rx = math.radians(rng.uniform(-tilt_max_deg, tilt_max_deg))
        ry = math.radians(rng.uniform(-tilt_max_deg, tilt_max_deg))
        R = rot_xyz(rx, ry, rz)
        z = K[0, 0] * size / side
        t = np.array([(cxp - K[0, 2]) / K[0, 0] * z, (cyp - K[1, 2]) / K[1, 1] * z, z])
        H = homography_from_pose(R, t, K, size)
        tags.append({"family": fam, "id": tid, "H": H})
        truth.append(truth_from_H(fam, tid, H, R, t))
    img = render(width, height, tags, background=150, sigma=sigma, seed=seed)
    return img, K, truth


def scene_c2(seed=1234, sigma=2.0):
    """Config 2: 1920x1080, tag36h11 ids 0-9 on a jittered 5x2 grid, side U[96,192] px,
    in-plane rotation U[-30,30] deg, tilt <= 25 deg, background 150 + noise sigma."""
    return _grid_scene(1920, 1080, [("tag36h11", i) for i in range(10)], 5, 2, seed, 96, 192, 30, 25, sigma)


def scene_c2_ids(ids, seed=1234, sigma=2.0):
    """Config 2's layout with the ten given tag36h11 ids (golden vectors and tests beyond ids 0-9, e.g. 100..586)."""
    assert len(ids) == 10
    return _grid_scene(1920, 1080, [("tag36h11", int(i)) for i in ids], 5, 2, seed, 96, 192, 30, 25, sigma)


def scene_c5(seed=1234, sigma=2.0):
    """Config 5: config-2 layout with 5 tag36h11 ids and 5 tag25h9 ids."""
    ids = [("tag36h11", i) for i in range(5)] + [("tag25h9", i) for i in range(5)]
    return _grid_scene(1920, 1080, ids, 5, 2, seed, 96, 192, 30, 25, sigma)


def scene_c3(seed=77, sigma=2.0):
    """Config 3: 3840x2160, 10x10 board of tag36h11 ids 0-99, side 160 px,
    gap 40 px, board tilted 10 deg about the vertical axis (long lens, f = 4000 px, so that the whole
    board stays in view)."""
    width, height = 3840, 2160
    K = np.array([[4000.0, 0, width / 2.0], [0, 4000.0, height / 2.0], [0, 0, 1]])
    z = 4.0
    scale_m = z / K[0, 0]          # metres per pixel at depth z
    pitch_m, size = 200.0 * scale_m, 160.0 * scale_m
    Rb = rot_xyz(0, math.radians(10.0), 0)
    tags, truth = [], []
    for r in range(10):
        for c in range(10):
            off = np.array([(c - 4.5) * pitch_m, (r - 4.5) * pitch_m, 0.0])
            t = np.array([0.0, 0.0, z]) + Rb @ off
            H = homography_from_pose(Rb, t, K, size)
            tid = r * 10 + c
            tags.append({"family": "tag36h11", "id": tid, "H": H})
            truth.append(truth_from_H("tag36h11", tid, H, Rb, t))
    img = render(width, height, tags, background=150, sigma=sigma, seed=seed)
    return img, K, truth, size
