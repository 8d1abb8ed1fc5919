"""Python host-side view of the detector handle (thin; all work happens behind the C ABI).

Mirrors the call shape the reference node uses on cuAprilTags (reference
isaac_ros_apriltag/src/apriltag_node.cpp:450-452 create, :491-493 detect, :556 destroy): create once
per (width, height, family, intrinsics, size); detect is host-synchronous; destroy in close().
"""
import ctypes as C

import numpy as np

from . import capi


def _as_images(frames, width, height, channels=1):
    """frames: torch uint8 CUDA tensor [n,H,W] / [H,W] (colour: [n,H,W,C] / [H,W,C], interleaved), or a list of such tensors,
    or a list of (dev_ptr, pitch) pairs.  Returns (ctypes array, keepalive)."""
    items = []
    if hasattr(frames, "data_ptr"):
        t = frames
        if t.dim() == (2 if channels == 1 else 3):
            t = t.unsqueeze(0)
        if channels == 1:
            assert t.dim() == 3 and t.stride(2) == 1, "expected [n,H,W] mono8 with unit pixel stride"
        else:
            assert t.dim() == 4 and t.shape[3] == channels and t.stride(3) == 1 and t.stride(2) == channels, "expected [n,H,W,C] interleaved"
        for i in range(t.shape[0]):
            items.append((t[i].data_ptr(), t.stride(1)))
    else:
        for f in frames:
            if hasattr(f, "data_ptr"):
                if channels == 1:
                    assert f.dim() == 2 and f.stride(1) == 1
                else:
                    assert f.dim() == 3 and f.shape[2] == channels and f.stride(2) == 1 and f.stride(1) == channels
                items.append((f.data_ptr(), f.stride(0)))
            else:
                items.append((int(f[0]), int(f[1])))
    arr = (capi.ImageInput * len(items))()
    for i, (ptr, pitch) in enumerate(items):
        arr[i].width, arr[i].height, arr[i].dev_ptr, arr[i].pitch = width, height, ptr, pitch
    return arr, frames


class AprilTagDetector:
    def __init__(self, width, height, families=("tag36h11",), decimate=1, intrinsics=None, tag_size=0.22, max_batch=1,
                 tile_size=4, device=-1, refine_edges=True, **caps):
        L = capi.lib()
        cfg = capi.Config()
        L.amdAprilTagsDefaultConfig(C.byref(cfg), width, height)
        cfg.tile_size = tile_size
        cfg.decimate = decimate
        cfg.num_families = len(families)
        for i, f in enumerate(families):
            e = f if isinstance(f, int) else L.amdAprilTagsFamilyFromName(f.encode())
            if e < 0:
                raise capi.AprilTagsError("family lookup %r" % (f,), 2)
            cfg.families[i] = e
        if intrinsics is not None:
            cfg.intrinsics = capi.Intrinsics(*[float(v) for v in intrinsics])
        cfg.tag_size = tag_size
        cfg.max_batch = max_batch
        cfg.refine_edges = 1 if refine_edges else 0
        cfg.device = device
        for k, v in caps.items():
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
        self.families = [f if isinstance(f, str) else capi.family_info(f)["name"] for f in families]
        self.width, self.height, self.decimate, self.max_batch = width, height, decimate, max_batch
        self._h = C.c_void_p()
        capi._check("amdCreateAprilTagsDetectorEx", L.amdCreateAprilTagsDetectorEx(C.byref(self._h), C.byref(cfg)))
        self._L = L

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.amdAprilTagsDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- detection --------------------------------------------------------------------------------
    def detect_batch_ex(self, frames, max_dets=64, intrinsics=None, stream=None, encoding="mono8"):
        """encoding != "mono8": frames are interleaved colour ([n,H,W,C]) and go through amdAprilTagsDetectBatchColorEx."""
        imgs, keep = _as_images(frames, self.width, self.height, capi.ENC_CHANNELS[encoding])
        n = len(imgs)
        out = (capi.DetectionEx * (n * max_dets))()
        cnt = (C.c_uint32 * n)()
        intr = None
        if intrinsics is not None:
            intr = (capi.Intrinsics * n)(*[capi.Intrinsics(*[float(v) for v in k]) for k in intrinsics])
        if encoding == "mono8":
            capi._check("amdAprilTagsDetectBatchEx",
                        self._L.amdAprilTagsDetectBatchEx(self._h, n, imgs, intr, out, cnt, max_dets, stream))
        else:
            capi._check("amdAprilTagsDetectBatchColorEx",
                        self._L.amdAprilTagsDetectBatchColorEx(self._h, n, imgs, capi.ENCODINGS[encoding], intr, out, cnt, max_dets, stream))
        res = []
        for f in range(n):
            dets = []
            for i in range(cnt[f]):
                d = out[f * max_dets + i]
                dets.append({"family": self.families[d.family], "id": int(d.id), "hamming": int(d.hamming),
                             "decision_margin": float(d.decision_margin),
                             "H": np.array(list(d.H)).reshape(3, 3), "center": np.array(list(d.c)),
                             "p": np.array([[d.p[k][0], d.p[k][1]] for k in range(4)]),
                             "R": np.array(list(d.R)).reshape(3, 3), "t": np.array(list(d.t))})
            res.append(dets)
        return res

    # ---- prepared submissions: argument marshalling done once, the timed call is only the C ABI call ----
    def prepare(self, frames, max_dets=64, intrinsics=None, encoding="mono8"):
        imgs, keep = _as_images(frames, self.width, self.height, capi.ENC_CHANNELS[encoding])
        n = len(imgs)
        intr = None
        if intrinsics is not None:
            assert len(intrinsics) == n
            intr = (capi.Intrinsics * n)(*[capi.Intrinsics(*[float(v) for v in k]) for k in intrinsics])
        return {"imgs": imgs, "keep": keep, "n": n, "max_dets": max_dets, "intr": intr, "enc": capi.ENCODINGS[encoding],
                "out": (capi.DetectionEx * (n * max_dets))(), "cnt": (C.c_uint32 * n)()}

    def run_prepared(self, prep, stream=None):
        """One blocking amdAprilTagsDetectBatchEx call; results stay in prep['out'] / prep['cnt']."""
        if prep.get("enc", 0):
            capi._check("amdAprilTagsDetectBatchColorEx",
                        self._L.amdAprilTagsDetectBatchColorEx(self._h, prep["n"], prep["imgs"], prep["enc"], prep.get("intr"), prep["out"],
                                                               prep["cnt"], prep["max_dets"], stream))
            return
        capi._check("amdAprilTagsDetectBatchEx",
                    self._L.amdAprilTagsDetectBatchEx(self._h, prep["n"], prep["imgs"], prep.get("intr"), prep["out"],
                                                      prep["cnt"], prep["max_dets"], stream))

    def submit_prepared(self, prep, stream=None):
        """amdAprilTagsSubmitBatch: enqueues and returns; wait_prepared() collects (results in prep['out'] / prep['cnt'])."""
        if prep.get("enc", 0):
            capi._check("amdAprilTagsSubmitBatchColor",
                        self._L.amdAprilTagsSubmitBatchColor(self._h, prep["n"], prep["imgs"], prep["enc"], prep.get("intr"), prep["max_dets"], stream))
            return
        capi._check("amdAprilTagsSubmitBatch",
                    self._L.amdAprilTagsSubmitBatch(self._h, prep["n"], prep["imgs"], prep.get("intr"), prep["max_dets"], stream))

    def wait_prepared(self, prep):
        capi._check("amdAprilTagsWaitBatchEx", self._L.amdAprilTagsWaitBatchEx(self._h, prep["out"], prep["cnt"]))

    def unpack(self, prep):
        res = []
        out, cnt, max_dets = prep["out"], prep["cnt"], prep["max_dets"]
        for f in range(prep["n"]):
            dets = []
            for i in range(cnt[f]):
                d = out[f * max_dets + i]
                dets.append({"family": self.families[d.family], "id": int(d.id), "hamming": int(d.hamming),
                             "decision_margin": float(d.decision_margin),
                             "H": np.array(list(d.H)).reshape(3, 3), "center": np.array(list(d.c)),
                             "p": np.array([[d.p[k][0], d.p[k][1]] for k in range(4)]),
                             "R": np.array(list(d.R)).reshape(3, 3), "t": np.array(list(d.t))})
            res.append(dets)
        return res

    def detect_batch_raw(self, frames, max_tags=64, intrinsics=None, stream=None):
        """Returns (TagID ctypes array of n*max_tags, counts) -- the cuAprilTagsID_t-shaped records."""
        imgs, keep = _as_images(frames, self.width, self.height)
        n = len(imgs)
        out = (capi.TagID * (n * max_tags))()
        cnt = (C.c_uint32 * n)()
        intr = None
        if intrinsics is not None:
            intr = (capi.Intrinsics * n)(*[capi.Intrinsics(*[float(v) for v in k]) for k in intrinsics])
        capi._check("amdAprilTagsDetectBatch",
                    self._L.amdAprilTagsDetectBatch(self._h, n, imgs, intr, out, cnt, max_tags, stream))
        return out, [int(c) for c in cnt]

    def threshold_only(self, frames, stream=None, encoding="mono8"):
        imgs, keep = _as_images(frames, self.width, self.height, capi.ENC_CHANNELS[encoding])
        if encoding == "mono8":
            capi._check("amdAprilTagsThresholdOnly", self._L.amdAprilTagsThresholdOnly(self._h, len(imgs), imgs, stream))
        else:
            capi._check("amdAprilTagsThresholdOnlyColor",
                        self._L.amdAprilTagsThresholdOnlyColor(self._h, len(imgs), imgs, capi.ENCODINGS[encoding], stream))

    # ---- measurement / inspection ---------------------------------------------------------------------
    def set_profiling(self, enable=True):
        """True/1: HIP events per stage; 2: plus cycle counters inside the quad-fit kernel."""
        capi._check("amdAprilTagsSetProfiling", self._L.amdAprilTagsSetProfiling(self._h, int(enable)))

    def set_submission_path(self, path):
        """'auto' (by submission size), 'latency' or 'throughput': pins the launch set (parity tests; results never depend on it)."""
        code = {"auto": capi.PATH_AUTO, "latency": capi.PATH_LATENCY, "throughput": capi.PATH_THROUGHPUT}[path] if isinstance(path, str) else int(path)
        capi._check("amdAprilTagsDebugSetSubmissionPath", self._L.amdAprilTagsDebugSetSubmissionPath(self._h, code))

    def late_waits(self):
        return int(self._L.amdAprilTagsDebugLateWaits(self._h))

    def graph_replay(self):
        """(still capturing new launch graphs?, live cache entries, retired graphs) -- include/apriltag_amd_debug.h."""
        live, ret = C.c_uint32(), C.c_uint32()
        on = self._L.amdAprilTagsDebugGraphReplay(self._h, C.byref(live), C.byref(ret))
        return bool(on), int(live.value), int(ret.value)

    def last_submission_path(self):
        return {capi.PATH_AUTO: "none", capi.PATH_LATENCY: "latency", capi.PATH_THROUGHPUT: "throughput"}[
            self._L.amdAprilTagsDebugLastSubmissionPath(self._h)]

    def stage_ms(self):
        ms = (C.c_float * capi.NUM_STAGES)()
        capi._check("amdAprilTagsGetStageMs", self._L.amdAprilTagsGetStageMs(self._h, ms))
        return dict(zip(capi.stage_names(), [float(v) for v in ms]))

    def frame_flags(self, n):
        fl = (C.c_uint32 * n)()
        capi._check("amdAprilTagsGetFrameFlags", self._L.amdAprilTagsGetFrameFlags(self._h, fl, n))
        return [int(v) for v in fl]

    def mean_counts(self, n, sample=16):
        """Mean per-frame content counters of the last submission over `sample` evenly spaced frames."""
        idx = sorted(set(int(i) for i in np.linspace(0, n - 1, min(sample, n))))
        c = np.array([self.debug(i, capi.DBG_COUNTS)[:5] for i in idx], dtype=np.float64).mean(axis=0)
        return {"npoints_raw": float(c[0]), "nclusters": float(c[1]), "npoints_kept": float(c[2]), "nquads": float(c[3]),
                "ndets_raw": float(c[4])}

    def device_bytes(self):
        """Device memory the handle owns."""
        nb = C.c_size_t()
        capi._check("amdAprilTagsGetDeviceBytes", self._L.amdAprilTagsGetDeviceBytes(self._h, C.byref(nb)))
        return int(nb.value)

    def debug(self, frame, what):
        nbytes = C.c_size_t()
        capi._check("amdAprilTagsDebugCopy", self._L.amdAprilTagsDebugCopy(self._h, frame, what, None, 0, C.byref(nbytes)))
        buf = np.empty(max(nbytes.value, 1), dtype=np.uint8)
        capi._check("amdAprilTagsDebugCopy",
                    self._L.amdAprilTagsDebugCopy(self._h, frame, what, buf.ctypes.data, buf.size, C.byref(nbytes)))
        buf = buf[:nbytes.value]
        if what in (capi.DBG_GRAY, capi.DBG_THRESH):
            return buf
        if what in (capi.DBG_LABEL, capi.DBG_CSIZE, capi.DBG_POINTS, capi.DBG_COUNTS):
            return buf.view(np.uint32)
        if what == capi.DBG_CLUSTERS:
            return buf.view(np.dtype([("key", "<u8"), ("start", "<u4"), ("count", "<u4")]))
        if what == capi.DBG_QUADS:
            return buf.view(np.dtype([("p", "<f4", (4, 2)), ("reversed_border", "<i4"), ("pad", "<u4"), ("key", "<u8")]))
        return buf
