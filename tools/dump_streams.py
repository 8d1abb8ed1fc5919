#!/usr/bin/env python3
"""Writes the input file of examples/multi_stream_host.cpp: S camera streams x F frames of the config-2 generator with
the per-stream parameter block of isaac_ros_apriltag_amd/streams.py (the same block bench.py broadcasts).
Usage: python tools/dump_streams.py out.bin [streams=8] [frames=4] [sigma=2] [decimate=1]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import streams, synth  # noqa: E402


def dump(path, nstreams=8, nframes=4, sigma=2.0, decimate=1, width=1920, height=1080, tag_sizes=None):
    """tag_sizes: optional per-stream tag sizes (cycled); default = one size for all streams."""
    block = streams.make_param_block(nstreams, width, height, decimate)
    if tag_sizes:
        for s in range(nstreams):
            block[s][streams.PARAM_FIELDS.index("tag_size")] = tag_sizes[s % len(tag_sizes)]
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", 0x31535441, nstreams, nframes, width, height, decimate))
        for s in range(nstreams):
            sp = streams.stream_params(block, s)
            f.write(struct.pack("<5d", sp["fx"], sp["fy"], sp["cx"], sp["cy"], sp["tag_size"]))
        for s in range(nstreams):
            seed = int(streams.stream_params(block, s)["seed"])
            for i in range(nframes):
                img = synth.scene_c2(seed=seed + i, sigma=sigma)[0]
                assert img.shape == (height, width)
                f.write(np.ascontiguousarray(img).tobytes())
    return block


if __name__ == "__main__":
    a = sys.argv[1:]
    dump(a[0], *(int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 4, float(a[3]) if len(a) > 3 else 2.0,
                 int(a[4]) if len(a) > 4 else 1))
