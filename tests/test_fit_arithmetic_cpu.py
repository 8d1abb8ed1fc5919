"""Host-side checks of the index and floating-point algebra the kernels rely on (sqrt of the line-fit weights, skewed key array, frame-interleaved block order, one-word staging record, register sort network)
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


# ---- host models of this round's index algebra (device versions: tests/test_gpu_parity.py, tools/fuzz_gpu.py) -------------
def _frame_block(lin, bpf, n, G):
    """at_frame_block (csrc/common.h): block -> (frame, block of the frame), the frames of a group of G taking turns."""
    if G <= 1:
        return lin // bpf, lin % bpf
    ngroups = (n + G - 1) // G
    group = min(lin // (bpf * G), ngroups - 1)
    rem = lin - group * bpf * G
    gsize = n - group * G if group == ngroups - 1 else G
    blk = rem // gsize
    return group * G + (rem - blk * gsize), blk


@pytest.mark.parametrize("bpf,n,G", [(1, 1, 256), (7, 1, 256), (5, 3, 2), (510, 8, 256), (13, 20, 16), (3, 33, 16), (2040, 5, 4),
                                     (9, 256, 256), (4, 17, 1), (6, 300, 256)])
def test_frame_interleaved_block_order_is_a_bijection(bpf, n, G):
    """Every (frame, block) pair is produced exactly once by the 1-D grid of bpf * n blocks, whatever the group size -- also
    when the last group is short -- and inside a full group consecutive blocks belong to different frames."""
    seen = set()
    for lin in range(bpf * n):
        f, b = _frame_block(lin, bpf, n, G)
        assert 0 <= f < n and 0 <= b < bpf
        seen.add((f, b))
    assert len(seen) == bpf * n
    if G > 1 and n >= 2:
        g = min(G, n)
        assert len({_frame_block(lin, bpf, n, G)[0] for lin in range(g)}) == g   # the first g blocks: g different frames


def _pack_point(x, y, gx, gy):
    return (x << 18) | (y << 4) | ((gx // 255 + 1) << 2) | (gy // 255 + 1)


def test_staging_word_round_trip():
    """k_points stages ONE word per boundary point -- table entry (8 bits) | rank in the block's group (11) | pixel of the
    64 x 16 tile (10) | direction (2) | sign of the value step (1) -- and k_scatter rebuilds the packed point from the tile
    origin: the rebuilt point equals the one the old 8-byte record carried, for every field combination that can occur."""
    import random
    rng = random.Random(3)
    for _ in range(20000):
        e, rk, pix, d, neg = rng.randrange(255), rng.randrange(2048), rng.randrange(1024), rng.randrange(4), rng.randrange(2)
        X0, Y0 = 64 * rng.randrange(60), 16 * rng.randrange(135)
        w = e | (rk << 8) | (pix << 19) | (d << 29) | (neg << 31)
        assert w != 0xFFFFFFFF and (w & 255) != 255
        # k_scatter's decode
        e2, rk2, pix2, d2 = w & 255, (w >> 8) & 2047, (w >> 19) & 1023, (w >> 29) & 3
        assert (e2, rk2, pix2, d2, w >> 31) == (e, rk, pix, d, neg)
        ly, plx = pix2 >> 6, pix2 & 63
        ddx = -1 if d2 == 2 else (0 if d2 == 1 else 1)
        ddy = 0 if d2 == 0 else 1
        step = -255 if (w >> 31) else 255
        rebuilt = _pack_point(2 * (X0 + plx) + ddx, 2 * (Y0 + ly) + ddy, ddx * step, ddy * step)
        # k_points' emit with the pixel values: v0 white <=> step negative
        v0, v1 = (255, 0) if neg else (0, 255)
        direct = _pack_point(2 * (X0 + plx) + ddx, 2 * (Y0 + ly) + ddy, ddx * (v1 - v0), ddy * (v1 - v0))
        assert rebuilt == direct


@pytest.mark.parametrize("K", [1, 2, 4])
def test_register_sort_network_sorts(K):
    """fq_wave_sort<K> (kernels_quad.h): lane l holds keys l K .. l K + K - 1; flips meet index i with i ^ (2^lk - 1) -- across
    lanes that is lane ^ (2^lk / K - 1), register K - 1 - j -- half-cleaners meet i with i ^ S; the lower lane of an exchange
    keeps the minimum.  The model runs the same steps on random keys (with +infinity pads) and must leave them sorted."""
    import random
    rng = random.Random(K)
    LK = {1: 0, 2: 1, 4: 2}[K]
    for trial in range(40):
        n = 64 * K
        real = rng.randrange(1, n + 1)
        keys = [rng.randrange(1 << 20) for _ in range(real)] + [float("inf")] * (n - real)
        v = [[keys[l * K + j] for j in range(K)] for l in range(64)]
        for lk in range(1, LK + 7):
            if (1 << lk) <= K:
                for l in range(64):
                    for j in range(K):
                        j2 = j ^ ((1 << lk) - 1)
                        if j < j2:
                            a, b = v[l][j], v[l][j2]
                            v[l][j], v[l][j2] = min(a, b), max(a, b)
            else:
                lx = (1 << (lk - LK)) - 1
                old = [row[:] for row in v]
                for l in range(64):
                    keep_min = (l & (1 << (lk - LK - 1))) == 0
                    for j in range(K):
                        o = old[l ^ lx][K - 1 - j]
                        v[l][j] = min(old[l][j], o) if keep_min else max(old[l][j], o)
            S = (1 << lk) >> 2
            while S >= 1:
                if S < K:
                    for l in range(64):
                        for j in range(K):
                            if (j & S) == 0:
                                a, b = v[l][j], v[l][j + S]
                                v[l][j], v[l][j + S] = min(a, b), max(a, b)
                else:
                    lx = S // K
                    old = [row[:] for row in v]
                    for l in range(64):
                        keep_min = (l & lx) == 0
                        for j in range(K):
                            o = old[l ^ lx][j]
                            v[l][j] = min(old[l][j], o) if keep_min else max(old[l][j], o)
                S >>= 1
        flat = [v[l][j] for l in range(64) for j in range(K)]
        assert flat == sorted(keys)


def _pidx(a, b):
    return ((a * (19 - a)) >> 1) + b - a - 1


def test_pair_and_corner_choice_tables():
    """Index algebra of the quad fit's tail (kernels_quad.h: FQ_PIDX, fq_pair_of, make_combo_pairs): the triangular index
    of the 45 pairs a < b < 10, its closed-form inverse, and the packed pair indices of the 210 corner choices in upstream's
    loop order m0 < m1 < m2 < m3 (lexicographic: the first of equal errors wins)."""
    pairs = [(a, b) for a in range(9) for b in range(a + 1, 10)]
    assert [_pidx(a, b) for a, b in pairs] == list(range(45))
    for t, (a, b) in enumerate(pairs):   # fq_pair_of
        a2 = sum(t >= s for s in (9, 17, 24, 30, 35, 39, 42, 44))
        b2 = t - ((a2 * (19 - a2)) >> 1) + a2 + 1
        assert (a2, b2) == (a, b)
    combos = [(m0, m1, m2, m3) for m0 in range(7) for m1 in range(m0 + 1, 8) for m2 in range(m1 + 1, 9) for m3 in range(m2 + 1, 10)]
    assert len(combos) == 210 and combos == sorted(combos)
    for m0, m1, m2, m3 in combos:
        cp = _pidx(m0, m1) | (_pidx(m1, m2) << 6) | (_pidx(m2, m3) << 12) | (_pidx(m0, m3) << 18) | (m3 << 24)
        assert (cp & 63, (cp >> 6) & 63, (cp >> 12) & 63, (cp >> 18) & 63, cp >> 24) == (_pidx(m0, m1), _pidx(m1, m2), _pidx(m2, m3), _pidx(m0, m3), m3)
        assert max(_pidx(m0, m1), _pidx(m1, m2), _pidx(m2, m3), _pidx(m0, m3)) < 45 and cp < (1 << 28)


def test_packed_gradient_sign_sums():
    """k_fit_small reduces the border-direction sums in 32 bits: the signs of the two gradient components of a cluster's
    points travel packed as sgn(gx) * 65536 + sgn(gy) and are separated after the wave sum (up to 256 points)."""
    import random
    rng = random.Random(3)
    for _ in range(2000):
        n = rng.randint(1, 256)
        gx = [rng.choice((-1, 0, 1)) for _ in range(n)]
        gy = [rng.choice((-1, 0, 1)) for _ in range(n)]
        if rng.random() < 0.1:
            gx = [rng.choice((-1, 1))] * n
            gy = [rng.choice((-1, 1))] * n
        sg = sum(a * 65536 + b for a, b in zip(gx, gy))
        sg32 = (sg + (1 << 31)) % (1 << 32) - (1 << 31)            # int32 arithmetic
        low = sg32 & 0xFFFF
        sgy = low - 65536 if low >= 32768 else low                  # (int)(short)
        sgx = (sg32 - sgy) >> 16
        assert (sgx, sgy) == (sum(gx), sum(gy))
