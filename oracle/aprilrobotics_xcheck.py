"""Optional cross-check against a REAL AprilRobotics libapriltag.so, if the host has one.

TEST INFRASTRUCTURE ONLY (see oracle/apriltag_oracle.h).  AprilRobotics' apriltag is neither part of the
reference tree nor installed in the build image, so nothing depends on this module:
find_library() returns None there and every caller skips.  Where a libapriltag.so (3.x) exists -- set
APRILTAG_LIB or have it on the loader path -- this binds the handful of C entry points apriltag_detect() needs,
runs it on the same frames and reports (a) its speed as a second CPU row for bench.py and (b) how far the
restatement's ids and corners are from it (tests/test_aprilrobotics_xcheck_cpu.py states the rounding).

UNTESTED HERE: no libapriltag.so was available to run this against; the struct layouts below are those of the
public apriltag 3.x headers (apriltag.h, common/image_types.h, common/zarray.h, common/matd.h).
"""
import ctypes as C
import ctypes.util
import os
import time

import numpy as np


class image_u8_t(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32), ("buf", C.POINTER(C.c_uint8))]


class zarray_t(C.Structure):
    _fields_ = [("el_sz", C.c_size_t), ("size", C.c_int), ("alloc", C.c_int), ("data", C.c_void_p)]


class matd_t(C.Structure):
    _fields_ = [("nrows", C.c_uint), ("ncols", C.c_uint), ("data", C.c_double * 9)]   # flexible array; 3x3 here


class apriltag_detection_t(C.Structure):
    _fields_ = [("family", C.c_void_p), ("id", C.c_int), ("hamming", C.c_int), ("decision_margin", C.c_float),
                ("H", C.POINTER(matd_t)), ("c", C.c_double * 2), ("p", (C.c_double * 2) * 4)]


class apriltag_quad_thresh_params(C.Structure):
    _fields_ = [("min_cluster_pixels", C.c_int), ("max_nmaxima", C.c_int), ("critical_rad", C.c_float),
                ("cos_critical_rad", C.c_float), ("max_line_fit_mse", C.c_float), ("min_white_black_diff", C.c_int),
                ("deglitch", C.c_int)]


class apriltag_detector_head(C.Structure):
    """Leading user-settable fields of apriltag_detector_t (apriltag.h)."""
    _fields_ = [("nthreads", C.c_int), ("quad_decimate", C.c_float), ("quad_sigma", C.c_float),
                ("refine_edges", C.c_bool), ("decode_sharpening", C.c_double), ("debug", C.c_bool),
                ("qtp", apriltag_quad_thresh_params)]


def find_library():
    cand = [os.environ.get("APRILTAG_LIB"), ctypes.util.find_library("apriltag"), "libapriltag.so", "libapriltag.so.3"]
    for c in cand:
        if not c:
            continue
        try:
            return C.CDLL(c)
        except OSError:
            continue
    return None


def detect_all(lib, frames, decimate, nthreads=1):
    """Runs apriltag_detect (tag36h11, quad_sigma 0, refine_edges on) on each mono8 frame.
    Returns (list of per-frame [(id, hamming, corners[4][2])], seconds)."""
    lib.apriltag_detector_create.restype = C.POINTER(apriltag_detector_head)
    lib.tag36h11_create.restype = C.c_void_p
    lib.apriltag_detector_add_family_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.apriltag_detector_detect.restype = C.POINTER(zarray_t)
    lib.apriltag_detector_detect.argtypes = [C.c_void_p, C.POINTER(image_u8_t)]
    lib.apriltag_detections_destroy.argtypes = [C.POINTER(zarray_t)]
    lib.apriltag_detector_destroy.argtypes = [C.c_void_p]
    lib.tag36h11_destroy.argtypes = [C.c_void_p]
    td = lib.apriltag_detector_create()
    tf = lib.tag36h11_create()
    lib.apriltag_detector_add_family_bits(td, tf, 2)
    td.contents.nthreads = nthreads
    td.contents.quad_decimate = float(decimate)
    td.contents.quad_sigma = 0.0
    td.contents.refine_edges = True
    td.contents.decode_sharpening = 0.25
    out = []
    t0 = time.perf_counter()
    for fr in frames:
        fr = np.ascontiguousarray(fr, dtype=np.uint8)
        im = image_u8_t(fr.shape[1], fr.shape[0], fr.strides[0], fr.ctypes.data_as(C.POINTER(C.c_uint8)))
        za = lib.apriltag_detector_detect(td, C.byref(im))
        dets = []
        ptrs = C.cast(za.contents.data, C.POINTER(C.POINTER(apriltag_detection_t)))
        for i in range(za.contents.size):
            d = ptrs[i].contents
            dets.append((int(d.id), int(d.hamming), np.array([[d.p[k][0], d.p[k][1]] for k in range(4)])))
        dets.sort(key=lambda x: x[0])
        out.append(dets)
        lib.apriltag_detections_destroy(za)
    dt = time.perf_counter() - t0
    lib.apriltag_detector_destroy(td)
    lib.tag36h11_destroy(tf)
    return out, dt


def compare(real, restated):
    """real: [(id, hamming, corners)], restated: the oracle's detection dicts.  Returns (ids equal, max corner
    difference in px over tags present in both)."""
    rid = sorted(d[0] for d in real)
    oid = sorted(d["id"] for d in restated)
    byid = {d["id"]: d["p"] for d in restated}
    worst = 0.0
    for tid, _, p in real:
        if tid in byid:
            worst = max(worst, float(np.abs(p - byid[tid]).max()))
    return rid == oid, worst


def time_and_compare(frames, restated_by_frame, decimate):
    """bench.py's optional second CPU row.  None when no libapriltag is present."""
    lib = find_library()
    if lib is None:
        return None
    real, dt = detect_all(lib, frames, decimate)
    ids_ok, worst = True, 0.0
    for i, r in enumerate(real):
        if i in restated_by_frame:
            ok, w = compare(r, restated_by_frame[i])
            ids_ok &= ok
            worst = max(worst, w)
    return {"value": round(len(frames) / dt, 2), "unit": "frames/s", "cores": 1, "kind": "reference-algorithm (AprilRobotics libapriltag)",
            "ids_equal_to_restatement": bool(ids_ok), "max_corner_diff_px": worst}
