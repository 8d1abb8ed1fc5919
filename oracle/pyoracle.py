"""ctypes binding of oracle/libapriltag_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (isaac_ros_apriltag_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libapriltag_oracle.so")


class Family(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("nbits", C.c_uint32), ("d", C.c_uint32),
                ("width_at_border", C.c_uint32), ("total_width", C.c_uint32),
                ("reversed_border", C.c_int32), ("ncodes", C.c_uint32),
                ("codes", C.POINTER(C.c_uint64)), ("bit_x", C.c_int8 * 64), ("bit_y", C.c_int8 * 64)]


class Params(C.Structure):
    _fields_ = [("decimate", C.c_int32), ("tile_size", C.c_int32), ("min_white_black_diff", C.c_int32),
                ("min_component_size", C.c_int32), ("min_cluster_points", C.c_int32),
                ("max_nmaxima", C.c_int32), ("cos_critical_rad", C.c_double),
                ("max_line_fit_mse", C.c_double), ("refine_edges", C.c_int32),
                ("decode_sharpening", C.c_double), ("max_hamming", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("tag_size", C.c_double), ("skew", C.c_double), ("variant", C.c_int32)]


class Detection(C.Structure):
    _fields_ = [("family", C.c_int32), ("id", C.c_int32), ("hamming", C.c_int32),
                ("decision_margin", C.c_float), ("H", C.c_double * 9), ("c", C.c_double * 2),
                ("p", (C.c_double * 2) * 4), ("R", C.c_double * 9), ("t", C.c_double * 3)]


class Quad(C.Structure):
    _fields_ = [("p", (C.c_float * 2) * 4), ("reversed_border", C.c_int32), ("key", C.c_uint64)]


class Cluster(C.Structure):
    _fields_ = [("key", C.c_uint64), ("start", C.c_uint32), ("count", C.c_uint32)]


class Dump(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("gray", C.POINTER(C.c_uint8)),
                ("thr", C.POINTER(C.c_uint8)), ("label", C.POINTER(C.c_uint32)),
                ("csize", C.POINTER(C.c_uint32)), ("nclusters", C.c_uint32),
                ("clusters", C.POINTER(Cluster)), ("npoints", C.c_uint32),
                ("points", C.POINTER(C.c_uint32)), ("nquads", C.c_uint32), ("quads", C.POINTER(Quad)),
                ("ndet", C.c_uint32), ("dets", C.POINTER(Detection))]


def build():
    """(Re)build the oracle shared library with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def use_library(path):
    """Binds another build of the same source (bench.py's -O3 -march=native CPU leg); call before lib()."""
    global _lib, _LIB_PATH
    _LIB_PATH = path
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ato_default_params.argtypes = [C.POINTER(Params)]
        _lib.ato_builtin_family.argtypes = [C.c_char_p, C.POINTER(Family)]
        _lib.ato_builtin_family.restype = C.c_int
        _lib.ato_custom_family.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.c_uint32, C.c_uint32,
                                           C.c_int, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(Family)]
        _lib.ato_custom_family.restype = C.c_int
        _lib.ato_detect.argtypes = [C.POINTER(Params), C.POINTER(Family), C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.POINTER(Detection), C.c_int, C.POINTER(Dump)]
        _lib.ato_detect.restype = C.c_int
        _lib.ato_dump_free.argtypes = [C.POINTER(Dump)]
        _lib.ato_threshold.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.ato_connected_components.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ato_decimate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.ato_resize_mono8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.ato_rectify_mono8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.ato_pose_from_homography.argtypes = [C.POINTER(C.c_double)] + [C.c_double] * 5 + \
            [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.ato_pose_from_homography_ex.argtypes = [C.POINTER(C.c_double)] + [C.c_double] * 6 + \
            [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _lib


def default_params(**kw):
    p = Params()
    lib().ato_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def family(name):
    f = Family()
    if lib().ato_builtin_family(name.encode(), C.byref(f)) != 0:
        raise ValueError("unknown family %r" % name)
    return f


def custom_family(name, bit_x, bit_y, width_at_border, total_width, reversed_border, codes):
    """An AprilTag-3 style family given as data (bit i of the layout <-> bit nbits-1-i of a code)."""
    n = len(bit_x)
    bx = (C.c_int8 * n)(*[int(v) for v in bit_x])
    by = (C.c_int8 * n)(*[int(v) for v in bit_y])
    cc = (C.c_uint64 * len(codes))(*[int(c) for c in codes])
    f = Family()
    if lib().ato_custom_family(name.encode(), n, bx, by, width_at_border, total_width, int(bool(reversed_border)), cc, len(codes),
                               C.byref(f)) != 0:
        raise ValueError("layout rejected (not closed under rotation, or out of range)")
    f._keep = (bx, by, cc)   # the struct points into cc
    return f


def family_codes(name):
    f = family(name)
    return [int(f.codes[i]) for i in range(f.ncodes)], int(f.d)


def _det_to_dict(d, fam_names):
    return {
        "family": fam_names[d.family], "id": int(d.id), "hamming": int(d.hamming),
        "decision_margin": float(d.decision_margin),
        "H": np.array(list(d.H), dtype=np.float64).reshape(3, 3),
        "center": np.array(list(d.c), dtype=np.float64),
        "p": np.array([[d.p[i][0], d.p[i][1]] for i in range(4)], dtype=np.float64),
        "R": np.array(list(d.R), dtype=np.float64).reshape(3, 3),
        "t": np.array(list(d.t), dtype=np.float64),
    }


def threshold(gray, tile=4, min_diff=5):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    out = np.empty_like(gray)
    lib().ato_threshold(gray.ctypes.data, gray.shape[1], gray.shape[0], tile, min_diff, out.ctypes.data)
    return out


def connected_components(thr):
    thr = np.ascontiguousarray(thr, dtype=np.uint8)
    label = np.empty(thr.shape, dtype=np.uint32)
    csize = np.empty(thr.shape, dtype=np.uint32)
    lib().ato_connected_components(thr.ctypes.data, thr.shape[1], thr.shape[0], label.ctypes.data,
                                   csize.ctypes.data)
    return label, csize


def decimate(img, f):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((1 + (h - 1) // f, 1 + (w - 1) // f), dtype=np.uint8)
    sw, sh = C.c_int(), C.c_int()
    lib().ato_decimate(img.ctypes.data, w, h, img.strides[0], f, out.ctypes.data, C.byref(sw), C.byref(sh))
    return out


def resize_mono8(img, dw, dh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((dh, dw), dtype=np.uint8)
    lib().ato_resize_mono8(img.ctypes.data, img.strides[0], img.shape[1], img.shape[0], out.ctypes.data, dw, dw, dh)
    return out


def rectify_mono8(img, K, D, Knew):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    k = (C.c_double * 9)(*np.asarray(K, dtype=np.float64).reshape(-1))
    d = (C.c_double * 5)(*np.asarray(D, dtype=np.float64).reshape(-1))
    kn = (C.c_double * 9)(*np.asarray(Knew, dtype=np.float64).reshape(-1))
    lib().ato_rectify_mono8(img.ctypes.data, img.strides[0], out.ctypes.data, out.strides[0], img.shape[1], img.shape[0], k, d, kn)
    return out


VAR_SEQ_MOMENTS, VAR_ATAN_NORMAL, VAR_SVD_POLAR, VAR_FLOAT_DOT = 1, 2, 4, 8
VAR_AT3_BIT_ORDER = 32   # (16 and 64 were FLOAT_COS and TRIG_RZ: upstream's forms are the definition now)
VAR_FAST_PATHS = 128     # same statements through cheaper code (radix sort of the keys, quick_decode table, no exact sums beside SEQ_MOMENTS)


def pose_from_homography(H, fx, fy, cx, cy, tag_size, skew=0.0, variant=0):
    Hc = (C.c_double * 9)(*np.asarray(H, dtype=np.float64).reshape(-1))
    R = (C.c_double * 9)()
    t = (C.c_double * 3)()
    lib().ato_pose_from_homography_ex(Hc, fx, fy, cx, cy, skew, tag_size, variant, R, t)
    return np.array(list(R)).reshape(3, 3), np.array(list(t))


def detect(img, families=("tag36h11",), params=None, max_det=1024, want_dump=False):
    """Runs the full CPU restatement on one mono8 frame.

    Returns (detections, dump) where dump is None or a dict of numpy arrays (copies).
    """
    img = np.asarray(img, dtype=np.uint8)
    assert img.ndim == 2 and img.strides[1] == 1
    h, w = img.shape
    prm = params if params is not None else default_params()
    fam_objs = [f if isinstance(f, Family) else family(f) for f in families]   # names of built-ins or custom_family() objects
    fams = (Family * len(families))(*fam_objs)
    families = [f.name.decode() if isinstance(f, Family) else f for f in families]
    out = (Detection * max_det)()
    dump = Dump() if want_dump else None
    n = lib().ato_detect(C.byref(prm), fams, len(families), img.ctypes.data, w, h, img.strides[0], out, max_det,
                         C.byref(dump) if want_dump else None)
    if n < 0:
        raise RuntimeError("ato_detect failed with %d" % n)
    dets = [_det_to_dict(out[i], list(families)) for i in range(n)]
    d = None
    if want_dump:
        ww, hh = dump.w, dump.h
        npx = ww * hh
        d = {
            "w": ww, "h": hh,
            "gray": np.ctypeslib.as_array(dump.gray, (npx,)).reshape(hh, ww).copy(),
            "thr": np.ctypeslib.as_array(dump.thr, (npx,)).reshape(hh, ww).copy(),
            "label": np.ctypeslib.as_array(dump.label, (npx,)).reshape(hh, ww).copy(),
            "csize": np.ctypeslib.as_array(dump.csize, (npx,)).reshape(hh, ww).copy(),
            "clusters": [(int(dump.clusters[i].key), int(dump.clusters[i].start), int(dump.clusters[i].count))
                         for i in range(dump.nclusters)],
            "points": (np.ctypeslib.as_array(dump.points, (max(dump.npoints, 1),))[:dump.npoints].copy()),
            "quads": [{"key": int(dump.quads[i].key), "reversed_border": int(dump.quads[i].reversed_border),
                       "p": np.array([[dump.quads[i].p[k][0], dump.quads[i].p[k][1]] for k in range(4)],
                                     dtype=np.float32)} for i in range(dump.nquads)],
            "ndet_total": int(dump.ndet),
        }
        lib().ato_dump_free(C.byref(dump))
    return dets, d
