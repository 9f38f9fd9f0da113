"""Shared helpers for the parity tests (the oracle is the checker, never the thing under test)."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bit-exact float comparison (NaNs compare equal to NaNs)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def to_dev(torch, dev, arr):
    """numpy (any dtype, incl. structured) -> device byte/typed tensor."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype.fields is not None:
        return torch.from_numpy(arr.view(np.uint8).reshape(len(arr), arr.dtype.itemsize)).to(dev)
    return torch.from_numpy(arr).to(dev)


def make_pair(S, gpu, scene):
    torch, dev, ctx = gpu
    n = len(scene["sift"])
    d_sift = to_dev(torch, dev, scene["sift"])
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift)
    return pair, d_sift


def read_pnm_grey(path):
    """Binary PGM / PPM -> float32 grey image exactly as cv::imread(path, 0).convertTo(CV_32FC1) delivers it to the
    reference program (src/main.cpp:249-252): 8-bit fixed-point RGB -> grey (4899 R + 9617 G + 1868 B + 8192) >> 14.
    Python twin of ReadPNM in cuda-sfm_amd/host/sfm_io.h."""
    if path.endswith(".bz2"):                      # frames 4..35 of the dino ring are kept compressed
        import bz2
        b = bz2.decompress(open(path, "rb").read())
    else:
        b = open(path, "rb").read()
    toks, i = [], 0
    while len(toks) < 4:
        while b[i:i + 1].isspace():
            i += 1
        if b[i:i + 1] == b"#":
            while b[i:i + 1] != b"\n":
                i += 1
            continue
        j = i
        while not b[j:j + 1].isspace():
            j += 1
        toks.append(b[i:j]); i = j
    i += 1
    kind, w, h, mx = toks[0], int(toks[1]), int(toks[2]), int(toks[3])
    assert kind in (b"P5", b"P6") and mx == 255
    if kind == b"P5":
        return np.frombuffer(b, np.uint8, w * h, i).reshape(h, w).astype(np.float32)
    c = np.frombuffer(b, np.uint8, w * h * 3, i).reshape(h, w, 3).astype(np.int64)
    return ((c[..., 0] * 4899 + c[..., 1] * 9617 + c[..., 2] * 1868 + 8192) >> 14).astype(np.float32)


def dino_frame(k):
    """Path of frame k of the reference's data/dino sequence as kept under tests/golden/dino (8-bit grey, 0..3 plain PGM,
    4..35 bzip2)."""
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dino")
    plain = os.path.join(d, f"dino_grey_{k:03d}.pgm")
    return plain if os.path.exists(plain) else plain + ".bz2"


DINO_K = np.array([[2360.0, 0.0, 360.0], [0.0, 2360.0, 288.0], [0.0, 0.0, 1.0]], np.float32)          # src/main.cpp:292-297
DINO_KINV = np.array([[1.0 / 2360, 0.0, -360.0 / 2360], [0.0, 1.0 / 2360, -288.0 / 2360], [0.0, 0.0, 1.0]], np.float32)
DINO_SIFT = dict(num_octaves=5, init_blur=1.5, thresh=1.0, lowest_scale=0.0, scale_up=False)       # src/main.cpp:267-277


def lattice_points(rng, n, scale):
    """Unit-z point pairs whose second image sits on a 2^-12 lattice (scaled with the coordinate range), so that products
    with few-bit coefficients are exact in float arithmetic."""
    lattice = 2.0 ** -12 * 2.0 ** np.ceil(np.log2(scale))
    X0 = np.ones((3, n), np.float32); X1 = np.ones((3, n), np.float32)
    X1[:2] = (np.round(rng.uniform(-scale, scale, (2, n)) / lattice) * lattice).astype(np.float32)
    X0[:2] = (X1[:2] + rng.normal(size=(2, n)) * 0.01 * scale).astype(np.float32)
    X1[:2, 3] = X1[:2, 2]                                             # two points in one cell, one of them a duplicate
    return X0, X1


def crafted_candidates(rng, X1, n, H):
    """Candidate matrices for sfm_ransac_score_candidates: ordinary ones, and every family the zero-divisor guard of the
    pre-filter has to get right -- first two rows vanishing EXACTLY at a point of the set (well conditioned, nearly
    parallel, parallel = a whole line of zero divisors, identically zero), the same one ulp off, degenerate and
    non-finite matrices."""
    coef = np.array([-1.5, -1.0, -0.75, -0.5, -0.25, 0.25, 0.5, 0.75, 1.0, 1.5])
    Es = np.zeros((H, 3, 3), np.float32)
    for h in range(H):
        kind = h % 16
        j = int(rng.integers(n))
        px, py = float(X1[0, j]), float(X1[1, j])
        if kind < 6 or kind == 15:                                   # ordinary: random, entries up to ~1.9
            M = rng.normal(size=(3, 3)) * rng.choice([0.05, 1.0], size=(3, 3))
            Es[h] = (M * rng.uniform(0.2, 1.9) / np.abs(M).max()).astype(np.float32)
            continue
        a, b = rng.choice(coef, 2)
        if kind in (6, 7):
            c, d = rng.choice(coef, 2)
        elif kind in (8, 9):
            c, d = a + 2.0 ** -int(rng.integers(4, 12)), b
        elif kind in (10, 11):
            c, d = 0.5 * a, 0.5 * b                                   # parallel rows: x2 on the line a x + b y + e2 = 0 has da = 0
        elif kind == 12:
            c, d = 0.0, 0.0
        elif kind == 13:                                              # zero 2 x 2 part: da = e2^2 + e5^2 for every point
            Es[h] = np.array([[0, 0, rng.choice([0.0, 1e-30, 0.25])], [0, 0, 0], rng.uniform(-1, 1, 3)], np.float32)
            continue
        else:                                                         # non-finite / beyond the tame bound / all zero
            Es[h] = [np.full((3, 3), np.nan), np.full((3, 3), 3.0), np.zeros((3, 3)), np.full((3, 3), np.inf)][int(rng.integers(4))]
            continue
        E = np.array([[a, b, -(a * px + b * py)], [c, d, -(c * px + d * py)], rng.uniform(-1, 1, 3) * rng.choice([1e-3, 1.0])], np.float64)
        if np.abs(E).max() > 2.0:
            E /= 2.0
        E32 = E.astype(np.float32)
        if kind % 2 == 1:                                            # one ulp off: the exact zero is gone
            E32[0, 2] = np.nextafter(E32[0, 2], np.float32(9), dtype=np.float32)
        Es[h] = E32
    return Es
