"""Host-side checks of two pieces of index / floating-point algebra the quad-fit kernel relies on
(isaac_ros_apriltag_amd/csrc/kernels_quad.h, common.h); the device versions are exercised by tests/test_gpu_parity.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sqrt_u18_sequence_is_exact_for_every_argument(tmp_path):
    """The f32-seeded square root of the line-fit weights equals IEEE sqrt for all 2^18 integer arguments even when the
    hardware seed is off by up to 8 ulp (the device check, amdAprilTagsDebugMath op 5, covers the real seed)."""
    exe = str(tmp_path / "sqrt_u18_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", os.path.join(HERE, "aux_c", "sqrt_u18_check.c"), "-o", exe, "-lm"])
    assert subprocess.check_output([exe]).decode().strip() == "0"


def _kp(i):
    return i + (i >> 5)


@pytest.mark.parametrize("lpow", range(3, 15))
def test_skewed_key_index_shortcut(lpow):
    """bitonic_pass_r addresses the skewed LDS key array (one free slot per 32 keys) as FQ_KP(first key of the run) + a
    uniform per-element offset instead of FQ_KP(index) per key; the two must agree for every level, step, pass width
    and group (mirrors the index code of bitonic_pass_r)."""
    for lk in range(1, lpow + 1):
        for s in range(0, lk):
            for R in range(1, 5):
                if s + R > lk:
                    continue
                M = 1 << R
                lsp = lk - s - R
                spm = (1 << lsp) - 1
                soff = [_kp(e << lsp) for e in range(M)]
                for g in range((1 << lpow) >> R):
                    if s == 0:
                        blk, off = g >> lsp, g & spm
                        lo = (blk << lk) + off
                        hi = (blk << lk) + (1 << lk) - 1 - off
                        hi0 = hi - ((M // 2 - 1) << lsp)
                        idx = [0] * M
                        phys = [0] * M
                        for e in range(M // 2):
                            idx[e] = lo + (e << lsp)
                            idx[M - 1 - e] = hi - (e << lsp)
                            phys[e] = _kp(lo) + soff[e]
                            phys[M - 1 - e] = _kp(hi0) + soff[M // 2 - 1 - e]
                    else:
                        base = ((g >> lsp) << (lsp + R)) + (g & spm)
                        idx = [base + (e << lsp) for e in range(M)]
                        phys = [_kp(base) + soff[e] for e in range(M)]
                    assert phys == [_kp(i) for i in idx], (lpow, lk, s, R, g)


def test_skew_spreads_every_lane_stride_over_the_banks():
    """Walks give lane t the keys E*t .. E*t+E-1; with the skew the 32 lanes of an access group spread over at least 23
    of the 32 eight-byte bank pairs for every stride E the kernel uses (E = 31, which the skew would fold back onto one
    bank pair, is bumped to 32 by the kernel), so E needs no rounding to an odd number."""
    for E in range(1, 33):
        if (E & 31) == 31:
            assert len({_kp(E * t) % 32 for t in range(32)}) <= 2   # the reason for the bump
            continue
        for k in range(E):
            banks = {_kp(E * t + k) % 32 for t in range(32)}
            assert len(banks) >= 23, (E, k, len(banks))
