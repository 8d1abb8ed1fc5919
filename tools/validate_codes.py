"""Validates the recalled tag-family tables against the published lexicode-generator property.

The AprilTag family generator walks v <- v + 982451653 (mod 2^nbits) from a seed and appends every
accepted word, so every code of a genuine table equals code[0] + k*P (mod 2^nbits) with k strictly
increasing in id.  A mis-remembered digit breaks the relation (k becomes a random nbits-bit number).
Run: python tools/validate_codes.py   (exit code 0 = all tables consistent)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_ros_apriltag_amd import synth  # noqa: E402

P = 982451653


def stride_indices(codes, nbits):
    mask = (1 << nbits) - 1
    inv = pow(P, -1, 1 << nbits)
    return [((c - codes[0]) * inv) & mask for c in codes]


def main():
    ok = True
    for name, nbits in (("tag36h11", 36), ("tag25h9", 25), ("tag16h5", 16)):
        codes, d = synth.family_codes(name)
        ks = stride_indices(codes, nbits)
        mono = all(b > a for a, b in zip(ks, ks[1:]))
        print(name, "ncodes", len(codes), "monotone stride indices:", mono, ks[:12], "...")
        ok &= mono
    codes, _ = synth.family_codes("synth36h11")
    ks = stride_indices(codes, 36)
    print("synth36h11 (stand-in) ncodes", len(codes), "prefix equals tag36h11:",
          codes[:27] == synth.family_codes("tag36h11")[0])
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
