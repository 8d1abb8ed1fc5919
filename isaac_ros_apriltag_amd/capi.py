"""ctypes binding of libapriltag_amd.so -- the C ABI declared in include/apriltag_amd.h.

The library is the product: there is no CPU fallback.  Importing this module without the built
library, or creating a detector without a HIP device, raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libapriltag_amd.so")

NUM_STAGES = 12
FAMILY_ENUM = {"tag36h11": 0, "tag25h9": 1, "tag16h5": 2}
SLOT_TAG36H10, SLOT_CUSTOM0 = 3, 4   # registrable slots: 3 (tag36h10: no built-in table) and 4..8
(DBG_GRAY, DBG_THRESH, DBG_LABEL, DBG_CSIZE, DBG_CLUSTERS, DBG_POINTS, DBG_QUADS, DBG_COUNTS) = range(8)

STATUS = {0: "AMDAT_SUCCESS", 1: "AMDAT_INVALID_ARGUMENT", 2: "AMDAT_UNSUPPORTED", 3: "AMDAT_HIP_ERROR",
          4: "AMDAT_SIZE_MISMATCH", 5: "AMDAT_OUT_OF_MEMORY", 6: "AMDAT_BATCH_TOO_LARGE"}


class Intrinsics(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class ImageInput(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("dev_ptr", C.c_void_p), ("pitch", C.c_size_t)]


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class TagID(C.Structure):
    _fields_ = [("id", C.c_uint16), ("corners", Float2 * 4), ("hamming_error", C.c_uint16),
                ("orientation", C.c_float * 9), ("translation", C.c_float * 3), ("family", C.c_uint16),
                ("reserved", C.c_uint16), ("decision_margin", C.c_float), ("center", Float2)]


class DetectionEx(C.Structure):
    _fields_ = [("family", C.c_int32), ("id", C.c_int32), ("hamming", C.c_int32), ("decision_margin", C.c_float),
                ("H", C.c_double * 9), ("c", C.c_double * 2), ("p", (C.c_double * 2) * 4), ("R", C.c_double * 9),
                ("t", C.c_double * 3)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("tile_size", C.c_uint32), ("decimate", C.c_uint32),
                ("num_families", C.c_uint32), ("families", C.c_int * 4), ("intrinsics", Intrinsics),
                ("tag_size", C.c_float), ("max_batch", C.c_uint32), ("refine_edges", C.c_uint32),
                ("max_hamming", C.c_uint32), ("decode_sharpening", C.c_float), ("max_points", C.c_uint32),
                ("hash_slots", C.c_uint32), ("max_clusters", C.c_uint32), ("max_quads", C.c_uint32),
                ("max_detections", C.c_uint32), ("device", C.c_int32), ("skew", C.c_float), ("corner_convention", C.c_uint32),
                ("no_graph_replay", C.c_uint32), ("no_stream_priorities", C.c_uint32)]


# every symbol include/apriltag_amd.h declares
EXPORTS = ["amdAprilTagsDefaultConfig", "amdCreateAprilTagsDetector", "amdCreateAprilTagsDetectorEx",
           "amdAprilTagsDestroy", "amdAprilTagsDetect", "amdAprilTagsDetectBatch", "amdAprilTagsDetectBatchEx",
           "amdAprilTagsSubmitBatch", "amdAprilTagsWaitBatch", "amdAprilTagsWaitBatchEx", "amdAprilTagsSetFrameSkews",
           "amdAprilTagsGetFrameFlags", "amdAprilTagsConvertToMono8", "amdAprilTagsRegisterFamily",
           "amdAprilTagsRegisterFamilyEx", "amdAprilTagsUnregisterFamily",
           "amdAprilTagsFamilyInfo", "amdAprilTagsFamilyFromName", "amdAprilTagsStageName",
           "amdAprilTagsSetProfiling", "amdAprilTagsGetStageMs", "amdAprilTagsThresholdOnly",
           "amdAprilTagsDebugCopy", "amdAprilTagsDebugMath", "amdAprilTagsDeviceAlloc", "amdAprilTagsDeviceFree",
           "amdAprilTagsCopyToDevice", "amdAprilTagsResizeMono8", "amdAprilTagsRectifyMono8",
           "amdAprilTagsGetDeviceBytes", "amdAprilTagsDebugSetSubmissionPath", "amdAprilTagsDebugLastSubmissionPath", "amdAprilTagsDebugLateWaits",
           "amdAprilTagsEncodingFromName", "amdAprilTagsDetectColor", "amdAprilTagsDetectBatchColor", "amdAprilTagsDetectBatchColorEx",
           "amdAprilTagsSubmitBatchColor", "amdAprilTagsThresholdOnlyColor", "amdAprilTagsCopyToDeviceAsync", "amdAprilTagsStreamCreate",
           "amdAprilTagsStreamDestroy", "amdAprilTagsDebugGraphReplay", "amdAprilTagsConfigLayoutVersion"]
PATH_AUTO, PATH_LATENCY, PATH_THROUGHPUT = 0, 1, 2
ENCODINGS = {"mono8": 0, "rgb8": 1, "bgr8": 2, "rgba8": 3, "bgra8": 4}   # amdAprilTagsEncoding
ENC_CHANNELS = {"mono8": 1, "rgb8": 3, "bgr8": 3, "rgba8": 4, "bgra8": 4}

_lib = None


def lib():
    """Loads libapriltag_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libapriltag_amd.so is missing: run `python -m isaac_ros_apriltag_amd.build` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.amdAprilTagsDefaultConfig.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32]
    L.amdAprilTagsDefaultConfig.restype = None
    L.amdCreateAprilTagsDetector.argtypes = [C.POINTER(H), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                             C.POINTER(Intrinsics), C.c_float]
    L.amdCreateAprilTagsDetectorEx.argtypes = [C.POINTER(H), C.POINTER(Config)]
    L.amdAprilTagsDestroy.argtypes = [H]
    L.amdAprilTagsDetect.argtypes = [H, C.POINTER(ImageInput), C.POINTER(TagID), C.POINTER(C.c_uint32), C.c_uint32, H]
    L.amdAprilTagsDetectBatch.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.POINTER(Intrinsics),
                                          C.POINTER(TagID), C.POINTER(C.c_uint32), C.c_uint32, H]
    L.amdAprilTagsDetectBatchEx.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.POINTER(Intrinsics),
                                            C.POINTER(DetectionEx), C.POINTER(C.c_uint32), C.c_uint32, H]
    L.amdAprilTagsSubmitBatch.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.POINTER(Intrinsics), C.c_uint32, H]
    L.amdAprilTagsEncodingFromName.argtypes = [C.c_char_p]
    L.amdAprilTagsDetectColor.argtypes = [H, C.POINTER(ImageInput), C.c_int, C.POINTER(TagID), C.POINTER(C.c_uint32), C.c_uint32, H]
    L.amdAprilTagsDetectBatchColor.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.c_int, C.POINTER(Intrinsics),
                                               C.POINTER(TagID), C.POINTER(C.c_uint32), C.c_uint32, H]
    L.amdAprilTagsDetectBatchColorEx.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.c_int, C.POINTER(Intrinsics),
                                                 C.POINTER(DetectionEx), C.POINTER(C.c_uint32), C.c_uint32, H]
    L.amdAprilTagsSubmitBatchColor.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.c_int, C.POINTER(Intrinsics), C.c_uint32, H]
    L.amdAprilTagsThresholdOnlyColor.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), C.c_int, H]
    L.amdAprilTagsWaitBatch.argtypes = [H, C.POINTER(TagID), C.POINTER(C.c_uint32)]
    L.amdAprilTagsWaitBatchEx.argtypes = [H, C.POINTER(DetectionEx), C.POINTER(C.c_uint32)]
    L.amdAprilTagsSetFrameSkews.argtypes = [H, C.c_uint32, C.POINTER(C.c_float)]
    L.amdAprilTagsGetFrameFlags.argtypes = [H, C.POINTER(C.c_uint32), C.c_uint32]
    L.amdAprilTagsConvertToMono8.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                             C.c_size_t, H]
    L.amdAprilTagsRegisterFamily.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32]
    L.amdAprilTagsRegisterFamilyEx.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.c_uint32,
                                               C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.c_uint32]
    L.amdAprilTagsFamilyInfo.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint64))]
    L.amdAprilTagsUnregisterFamily.argtypes = [C.c_int]
    L.amdAprilTagsFamilyFromName.argtypes = [C.c_char_p]
    L.amdAprilTagsStageName.argtypes = [C.c_uint32]
    L.amdAprilTagsStageName.restype = C.c_char_p
    L.amdAprilTagsSetProfiling.argtypes = [H, C.c_int]
    L.amdAprilTagsGetStageMs.argtypes = [H, C.POINTER(C.c_float)]
    L.amdAprilTagsThresholdOnly.argtypes = [H, C.c_uint32, C.POINTER(ImageInput), H]
    L.amdAprilTagsDebugCopy.argtypes = [H, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amdAprilTagsDeviceAlloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.amdAprilTagsDeviceFree.argtypes = [C.c_void_p]
    L.amdAprilTagsCopyToDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, H]
    L.amdAprilTagsCopyToDeviceAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, H]
    L.amdAprilTagsStreamCreate.argtypes = [C.POINTER(H)]
    L.amdAprilTagsStreamDestroy.argtypes = [H]
    L.amdAprilTagsResizeMono8.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32,
                                          C.c_uint32, H]
    L.amdAprilTagsRectifyMono8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), H]
    L.amdAprilTagsGetDeviceBytes.argtypes = [H, C.POINTER(C.c_size_t)]
    L.amdAprilTagsDebugSetSubmissionPath.argtypes = [H, C.c_int]
    L.amdAprilTagsDebugLastSubmissionPath.argtypes = [H]
    L.amdAprilTagsDebugLateWaits.argtypes = [H]
    L.amdAprilTagsDebugGraphReplay.argtypes = [H, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.amdAprilTagsDebugMath.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int or name in ("amdAprilTagsFamilyFromName", "amdAprilTagsEncodingFromName"):
            fn.restype = C.c_int
    _lib = L
    return L


class AprilTagsError(RuntimeError):
    def __init__(self, where, code):
        super().__init__("%s failed: %s (error code %d)" % (where, STATUS.get(code, "?"), code))
        self.code = code


def _check(where, rc):
    if rc != 0:
        raise AprilTagsError(where, rc)


def stage_names():
    return [lib().amdAprilTagsStageName(i).decode() for i in range(NUM_STAGES)]


def family_info(name_or_enum):
    L = lib()
    fam = name_or_enum if isinstance(name_or_enum, int) else L.amdAprilTagsFamilyFromName(name_or_enum.encode())
    if fam < 0:
        raise ValueError("unknown family %r" % (name_or_enum,))
    nm, d, n, codes = C.c_char_p(), C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint64)()
    _check("amdAprilTagsFamilyInfo", L.amdAprilTagsFamilyInfo(fam, C.byref(nm), C.byref(d), C.byref(n), C.byref(codes)))
    return {"enum": fam, "name": nm.value.decode(), "d": d.value, "codes": [int(codes[i]) for i in range(n.value)]}


def register_family(slot, name, d, codes):
    """Classic d x d family (row-major codes) in a registrable slot."""
    cc = (C.c_uint64 * len(codes))(*[int(c) for c in codes])
    _check("amdAprilTagsRegisterFamily", lib().amdAprilTagsRegisterFamily(slot, name.encode(), d, cc, len(codes)))


def register_family_ex(slot, name, bit_x, bit_y, width_at_border, total_width, reversed_border, codes):
    """AprilTag-3 style layout (bit i at cell (bit_x[i], bit_y[i]) in border coordinates)."""
    n = len(bit_x)
    bx = (C.c_int8 * n)(*[int(v) for v in bit_x])
    by = (C.c_int8 * n)(*[int(v) for v in bit_y])
    cc = (C.c_uint64 * len(codes))(*[int(c) for c in codes])
    _check("amdAprilTagsRegisterFamilyEx", lib().amdAprilTagsRegisterFamilyEx(slot, name.encode(), n, bx, by, width_at_border, total_width,
                                                                             int(bool(reversed_border)), cc, len(codes)))


def unregister_family(slot):
    _check("amdAprilTagsUnregisterFamily", lib().amdAprilTagsUnregisterFamily(slot))


def debug_math(op, a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty_like(a)
    _check("amdAprilTagsDebugMath", lib().amdAprilTagsDebugMath(op, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data))
    return out
