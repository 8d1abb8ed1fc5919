"""CPU models of the wave-mask arithmetic of k_cc_local (isaac_ros_apriltag_amd/csrc/kernels_cc.h).

The kernel states the link rules of the tile-local connected components on 64-bit masks of a row (bit = column) and
keeps run lengths and list entries in packed forms; the border pass (cc_row_requests, same file) states the SAME rules
per pixel on the threshold image.  These tests restate both in Python and check them against each other on random rows,
plus the bounds the kernel's buffer sizes rely on.  No GPU, no library: pure algebra of the formulas in the source."""
import random

M64 = (1 << 64) - 1


def _masks(row):
    w = sum(1 << x for x, v in enumerate(row) if v == 255)
    b = sum(1 << x for x, v in enumerate(row) if v == 0)
    return w, b


def _src_masks(X0, W):
    src = sum(1 << i for i in range(64) if 1 <= X0 + i <= W - 2)
    lsrc = sum(1 << i for i in range(64) if X0 + i - 1 >= 1)
    rsrc = sum(1 << i for i in range(64) if X0 + i + 1 <= W - 2)
    return src, lsrc, rsrc


def _link_masks(row, up, X0, W):
    """pass 2 of k_cc_local: the three request masks of a row from the class masks of the row and of the row above"""
    W_, B_ = _masks(row)
    Wu, Bu = _masks(up)
    SRC, LSRC, RSRC = _src_masks(X0, W)
    sl = lambda m: (m << 1) & M64
    m0 = ((W_ & Wu & ~(sl(W_) & sl(Wu) & LSRC)) | (B_ & Bu & ~(sl(B_) & sl(Bu) & LSRC))) & SRC
    m1 = W_ & sl(Wu) & ~Wu & ~(sl(W_) & LSRC) & SRC
    m2 = W_ & (Wu >> 1) & ~((Wu | (W_ >> 1)) & RSRC) & SRC
    return m0 & M64, m1 & M64, m2 & M64


def _pixel_rules(row, up, X0, W, lane):
    """cc_row_requests' statements for the pixel in column `lane` of the tile (neighbours outside the tile count as class 127:
    the tile kernel links inside its tile only, the border pass owns the rest)"""
    at = lambda r, i: r[i] if 0 <= i < 64 else 127
    gx = X0 + lane
    if gx < 1 or gx > W - 2:
        return False, False, False
    v = row[lane]
    if v == 127:
        return False, False, False
    vl, vu, vul = at(row, lane - 1), at(up, lane), at(up, lane - 1)
    left_src = gx - 1 >= 1
    up_link = vu == v and not (left_src and vl == v and vul == v)
    ul = ur = False
    if v == 255:
        ul = vul == 255 and vu != 255 and not (left_src and vl == 255)
        vur, vr = at(up, lane + 1), at(row, lane + 1)
        right_src = gx + 1 <= W - 2
        ur = vur == 255 and not (right_src and (vu == 255 or vr == 255))
    return up_link, ul, ur


def _random_row(rng, W, X0, p127):
    row = []
    for i in range(64):
        if X0 + i >= W:
            row.append(127)      # the tile load pads columns beyond the image
        else:
            r = rng.random()
            row.append(127 if r < p127 else (255 if r < p127 + (1 - p127) / 2 else 0))
    return row


def test_link_masks_equal_the_per_pixel_rules():
    rng = random.Random(11)
    for trial in range(3000):
        W = rng.choice([1920, 1280, 644, 130, 70, 66, 65])
        X0 = 64 * rng.randrange((W + 63) // 64)
        p127 = rng.choice([0.0, 0.1, 0.5])
        row, up = _random_row(rng, W, X0, p127), _random_row(rng, W, X0, p127)
        m0, m1, m2 = _link_masks(row, up, X0, W)
        for lane in range(64):
            a, b, c = _pixel_rules(row, up, X0, W, lane)
            assert ((m0 >> lane) & 1, (m1 >> lane) & 1, (m2 >> lane) & 1) == (int(a), int(b), int(c)), (trial, lane)
        # the tile's first row: the row above is "no class" everywhere and every rule comes out empty without a row test
        assert _link_masks(row, [127] * 64, X0, W) == (0, 0, 0)


def test_request_list_holds_every_link_once_within_two_entries_per_pixel():
    """One entry per linked pixel (its up link, else up-left, else up-right) and a second entry for a pixel that has an
    up-right link besides: together exactly the links of the three masks, and never more than two per pixel -- the bound
    the list's LDS size is taken from (UREQ = rows x 2 x 64)."""
    rng = random.Random(12)
    for _ in range(3000):
        W = rng.choice([1920, 70, 65])
        X0 = 64 * rng.randrange((W + 63) // 64)
        row, up = _random_row(rng, W, X0, 0.05), _random_row(rng, W, X0, 0.05)
        m0, m1, m2 = _link_masks(row, up, X0, W)
        assert m0 & m1 == 0                       # up and up-left exclude each other
        mp, ms = m0 | m1 | m2, m2 & (m0 | m1)
        entries = []
        for lane in range(64):
            if (mp >> lane) & 1:
                entries.append((lane, 0 if (m0 >> lane) & 1 else (1 if (m1 >> lane) & 1 else 2)))
        for lane in range(64):
            if (ms >> lane) & 1:
                entries.append((lane, 2))
        want = [(l, t) for t, m in enumerate((m0, m1, m2)) for l in range(64) if (m >> l) & 1]
        assert sorted(entries) == sorted(want)
        assert len(entries) <= 2 * 64
        # the union's partner from an entry: byte offsets, partner = pixel - one row (256 B) - 4 (up-left) / + 4 (up-right)
        for lane, t in entries:
            p4 = (64 + lane) << 2                 # a pixel of tile row 1
            partner = p4 - 4 * 64 - ((t & 1) << 2) + ((t & 2) << 1)
            assert partner == ((lane + (0, -1, 1)[t]) << 2)


def test_run_labels_lengths_and_perimeter_flags():
    """pass 1: label = first pixel of the run (runs of every class, 'no class' included), the LAST pixel of a black or white
    run carries the run's length; pass 3 flags a run that starts in column 0 (length == lane + 1) or ends in column 63."""
    rng = random.Random(13)
    for _ in range(3000):
        W = rng.choice([1920, 130, 70, 65])
        X0 = 64 * rng.randrange((W + 63) // 64)
        row = _random_row(rng, W, X0, rng.choice([0.0, 0.2]))
        W_, B_ = _masks(row)
        N_ = ~(W_ | B_) & M64
        SRC, _, _ = _src_masks(X0, W)
        sl = lambda m: (m << 1) & M64
        L = ((W_ & sl(W_)) | (B_ & sl(B_)) | (N_ & sl(N_))) & SRC
        ends = ((~(L >> 1) & M64) | (1 << 63)) & ~N_ & M64
        total = 0
        for lane in range(64):
            below = M64 if lane == 63 else ((2 << lane) - 1)
            m = ~L & below & M64
            s = m.bit_length() - 1             # 63 - clz
            # reference: walk left while the pixel continues its left neighbour's run
            s_ref = lane
            while (L >> s_ref) & 1:
                s_ref -= 1
            assert s == s_ref
            if (ends >> lane) & 1:
                ln = lane + 1 - s
                assert 1 <= ln <= 64 and row[lane] != 127
                assert all(row[i] == row[lane] for i in range(s, lane + 1))
                assert lane == 63 or not ((L >> (lane + 1)) & 1)
                total += ln
                flag = ln == lane + 1 or lane == 63
                assert flag == (s == 0 or lane == 63)
        assert total == sum(1 for v in row if v != 127)   # every classed pixel is counted exactly once


def test_byte_offset_labels_split_into_row_and_column():
    """labels are byte offsets into the tile's parent array while the tile is worked on: offset = (row * 64 + column) * 4;
    the write pass takes row = offset >> 8 and column = (offset >> 2) & 63, and a strip's row k is the strip base OR k * 256."""
    for wv in range(4):
        for lane in range(64):
            base = (wv * 16 * 64 + lane) << 2
            for k in range(16):
                off = base | (k * 64 * 4)
                assert off == base + k * 256 == ((wv * 16 + k) * 64 + lane) << 2
                assert (off >> 8, (off >> 2) & 63) == (wv * 16 + k, lane)
    for wv in range(16):                          # sixteen waves per tile: four rows per wave
        for lane in (0, 63):
            base = (wv * 4 * 64 + lane) << 2
            for k in range(4):
                assert base | (k * 256) == ((wv * 4 + k) * 64 + lane) << 2


def test_staging_word_from_the_list_record():
    """k_points forms the staging word in its second pass from the list record (pixel | direction << 10): the record's low 12
    bits shifted by 19 ARE the word's pixel and direction fields."""
    rng = random.Random(14)
    for _ in range(5000):
        pix, d, e, rk, neg = rng.randrange(1024), rng.randrange(4), rng.randrange(255), rng.randrange(2048), rng.randrange(2)
        rec = pix | (d << 10)
        s0 = (neg << 31) | rng.randrange(1 << 21)
        w = e | (rk << 8) | ((rec & 0xFFF) << 19) | (s0 & 0x80000000)
        assert w == e | (rk << 8) | (pix << 19) | (d << 29) | (neg << 31)


def test_points_class_bits_make_the_emission_test_one_xor_and_one_compare():
    """k_points (kernels_cluster.h) keeps one word per tile pixel: representative | class bits (white 0x80000000, black
    0x40000000; 0 = not in a counted component).  `(a ^ b) > 0xBFFFFFFF` must hold exactly for a counted white and a
    counted black pixel, whatever their representatives (below 2^30)."""
    rng = random.Random(15)
    WHITE, BLACK, REP = 0x80000000, 0x40000000, 0x3FFFFFFF
    words = []
    for _ in range(300):
        r = rng.randrange(1 << 30) if rng.random() < 0.8 else rng.choice([0, 1, REP])
        words += [("w", WHITE | r), ("b", BLACK | r)]
    words += [("n", 0)] * 20
    for ca, a in words:
        for cb, b in rng.sample(words, 40):
            want = {ca, cb} == {"w", "b"}
            assert (((a ^ b) & 0xFFFFFFFF) > 0xBFFFFFFF) == want
            if ca != "n":
                assert a & REP == a & ~(WHITE | BLACK) & 0xFFFFFFFF   # the representative comes back with one mask
