"""ctypes view of libapriltag_node.so -- the ROS-free C++ mirror of the reference node shell
(include/apriltag_node_shell.hpp).  Used by the tests to drive the node logic the way the reference's
gtest / launch tests drive AprilTagNode."""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))


class ShellDetection(C.Structure):
    _fields_ = [("id", C.c_int32), ("family", C.c_char * 32), ("center", C.c_double * 2),
                ("corners", (C.c_double * 2) * 4), ("position", C.c_double * 3),
                ("orientation_xyzw", C.c_double * 4), ("child_frame_id", C.c_char * 48)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB_NODE
        if not os.path.exists(path):
            _build.build_node()
        L = C.CDLL(path)
        L.node_shell_create.restype = C.c_void_p
        L.node_shell_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.node_shell_create_strict.restype = C.c_void_p
        L.node_shell_create_strict.argtypes = L.node_shell_create.argtypes
        L.node_shell_destroy.argtypes = [C.c_void_p]
        L.node_shell_on_frame.restype = C.c_int
        L.node_shell_on_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_double), C.c_char_p, C.c_int32, C.c_uint32, C.c_int32, C.c_uint32,
                                          C.POINTER(ShellDetection), C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.node_shell_multi_create.restype = C.c_void_p
        L.node_shell_multi_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.node_shell_multi_destroy.argtypes = [C.c_void_p]
        L.node_shell_multi_on_frame.restype = C.c_int
        L.node_shell_multi_on_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.POINTER(C.c_double), C.c_char_p, C.c_int32, C.c_uint32, C.c_int32, C.c_uint32, C.c_char_p, C.c_size_t]
        L.node_shell_multi_flush.restype = C.c_int
        L.node_shell_multi_flush.argtypes = [C.c_void_p]
        L.node_shell_multi_publishes.restype = C.c_int
        L.node_shell_multi_publishes.argtypes = [C.c_void_p, C.c_int]
        L.node_shell_multi_last.restype = C.c_int
        L.node_shell_multi_last.argtypes = [C.c_void_p, C.c_int, C.POINTER(ShellDetection), C.c_int, C.c_char_p, C.c_size_t,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _unpack(out, n):
    dets = []
    for i in range(n):
        d = out[i]
        dets.append({"id": d.id, "family": d.family.decode(), "center": list(d.center),
                     "corners": [[d.corners[c][0], d.corners[c][1]] for c in range(4)],
                     "position": list(d.position), "orientation_xyzw": list(d.orientation_xyzw),
                     "child_frame_id": d.child_frame_id.decode()})
    return dets


class AprilTagMultiCameraNode:
    """S camera streams on one GPU, one detector submission per round (include/apriltag_node_shell.hpp)."""

    def __init__(self, num_streams, max_tags=64, size=0.22, tile_size=4, tag_family="tag36h11", backends="CUDA", decimate=1,
                 auto_flush=True):
        err = C.create_string_buffer(1024)
        self._L = lib()
        self._h = self._L.node_shell_multi_create(num_streams, max_tags, size, tile_size, tag_family.encode(), backends.encode(),
                                                  decimate, 1 if auto_flush else 0, err, 1024)
        if not self._h:
            raise RuntimeError(err.value.decode())
        self.max_tags, self.num_streams = max_tags, num_streams

    def close(self):
        if getattr(self, "_h", None):
            self._L.node_shell_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def on_frame(self, stream, data_ptr, is_device, encoding, width, height, step, K9, frame_id="tf_camera", stamp=(1, 0), info_stamp=None):
        info_stamp = stamp if info_stamp is None else info_stamp
        err = C.create_string_buffer(1024)
        k = (C.c_double * 9)(*[float(v) for v in K9])
        rc = self._L.node_shell_multi_on_frame(self._h, stream, data_ptr, 1 if is_device else 0, encoding.encode(), width, height, step, k,
                                               frame_id.encode(), stamp[0], stamp[1], info_stamp[0], info_stamp[1], err, 1024)
        if rc == -2:
            raise RuntimeError(err.value.decode())
        return rc == 1

    def flush(self):
        return self._L.node_shell_multi_flush(self._h)

    def publishes(self, stream):
        return self._L.node_shell_multi_publishes(self._h, stream)

    def last(self, stream):
        """(detections, header frame_id, (sec, nanosec)) of the last message published for `stream`."""
        out = (ShellDetection * self.max_tags)()
        fid = C.create_string_buffer(128)
        sec, nsec = C.c_int32(), C.c_uint32()
        n = self._L.node_shell_multi_last(self._h, stream, out, self.max_tags, fid, 128, C.byref(sec), C.byref(nsec))
        return _unpack(out, min(n, self.max_tags)), fid.value.decode(), (sec.value, nsec.value)


class AprilTagNode:
    """Parameters and defaults of the reference node (apriltag_node.cpp:564-568)."""

    def __init__(self, max_tags=64, size=0.22, tile_size=4, tag_family="tag36h11", backends="CUDA", decimate=1,
                 strict_cuapriltags_encodings=False):
        err = C.create_string_buffer(1024)
        self._L = lib()
        create = self._L.node_shell_create_strict if strict_cuapriltags_encodings else self._L.node_shell_create
        self._h = create(max_tags, size, tile_size, tag_family.encode(), backends.encode(), decimate, err, 1024)
        if not self._h:
            raise RuntimeError(err.value.decode())
        self.max_tags = max_tags

    def close(self):
        if getattr(self, "_h", None):
            self._L.node_shell_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def on_frame(self, data_ptr, is_device, encoding, width, height, step, K9, frame_id="tf_camera", stamp=(1, 0),
                 info_stamp=None):
        info_stamp = stamp if info_stamp is None else info_stamp
        out = (ShellDetection * self.max_tags)()
        fid = C.create_string_buffer(128)
        err = C.create_string_buffer(1024)
        k = (C.c_double * 9)(*[float(v) for v in K9])
        n = self._L.node_shell_on_frame(self._h, data_ptr, 1 if is_device else 0, encoding.encode(), width, height, step, k,
                                        frame_id.encode(), stamp[0], stamp[1], info_stamp[0], info_stamp[1], out, self.max_tags,
                                        fid, 128, err, 1024)
        if n == -2:
            raise RuntimeError(err.value.decode())
        if n < 0:
            return None, None
        dets = []
        for i in range(n):
            d = out[i]
            dets.append({"id": d.id, "family": d.family.decode(), "center": list(d.center),
                         "corners": [[d.corners[c][0], d.corners[c][1]] for c in range(4)],
                         "position": list(d.position), "orientation_xyzw": list(d.orientation_xyzw),
                         "child_frame_id": d.child_frame_id.decode()})
        return dets, fid.value.decode()
