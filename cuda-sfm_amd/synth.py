"""Portable synthetic inputs for the two-view path.

Everything is derived from SplitMix64 + Box-Muller in numpy uint64/float64 arithmetic, so the same
seed gives the same bytes on every machine (no std::*_distribution / rand()).  Used by bench.py,
the smoke test and the parity tests; the oracle and the HIP path both consume these arrays.
"""
import numpy as np

SEED = 0x5EED5F3D
_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

SIFT_DTYPE = np.dtype([
    ("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("sharpness", "<f4"),
    ("edgeness", "<f4"), ("orientation", "<f4"), ("score", "<f4"), ("ambiguity", "<f4"),
    ("match", "<i4"), ("match_xpos", "<f4"), ("match_ypos", "<f4"), ("match_error", "<f4"),
    ("subsampling", "<f4"), ("empty", "<f4", (3,)), ("data", "<f4", (128,)),
])


def splitmix64(seed, n, stream=0):
    """n 64-bit outputs of SplitMix64 started at seed (+ a stream offset)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(stream) * np.uint64(0xD1342543DE82EF95)
        z = base + _G * (np.arange(1, n + 1, dtype=np.uint64))
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n, stream=0):
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, n, stream=0):
    m = (n + 1) // 2
    u1 = 1.0 - uniform01(seed, m, 2 * stream + 1000)       # (0, 1]
    u2 = uniform01(seed, m, 2 * stream + 1001)
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
    return out[:n]


def camera(width=720, height=576, focal=2360.0):
    """K and K^-1 exactly as the reference's caller builds them (src/main.cpp:292-297), float32."""
    K = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1]], np.float64).astype(np.float32)
    Kinv = np.array([[1.0 / focal, 0, -(width / 2.0) / focal],
                     [0, 1.0 / focal, -(height / 2.0) / focal],
                     [0, 0, 1]], np.float64).astype(np.float32)
    return K, Kinv


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def two_view_scene(n, seed=SEED, noise_px=0.5, outlier_frac=0.3, width=720, height=576, focal=2360.0):
    """N correspondences of a rigid scene packed into SiftPoint records (xpos, ypos, match_xpos,
    match_ypos), so that fillXU is exercised.  Points uniform in [-1,1]^2 x [4,8] (cam-1 frame),
    cam-2 = Ry(10 deg) Rx(3 deg), t = normalize(1, 0.1, 0.2), sigma = noise_px, outliers: x2 uniform
    in the image."""
    K, Kinv = camera(width, height, focal)
    K64 = K.astype(np.float64)
    P = np.stack([2 * uniform01(seed, n, 1) - 1, 2 * uniform01(seed, n, 2) - 1, 4 + 4 * uniform01(seed, n, 3)], 1)
    R = _rot_y(np.deg2rad(10.0)) @ _rot_x(np.deg2rad(3.0))
    t = np.array([1.0, 0.1, 0.2]); t /= np.linalg.norm(t)
    p1 = (K64 @ P.T).T
    x1 = p1[:, :2] / p1[:, 2:]
    P2 = (R @ P.T).T + t
    p2 = (K64 @ P2.T).T
    x2 = p2[:, :2] / p2[:, 2:]
    x1 = x1 + noise_px * np.stack([normal(seed, n, 4), normal(seed, n, 5)], 1)
    x2 = x2 + noise_px * np.stack([normal(seed, n, 6), normal(seed, n, 7)], 1)
    is_out = uniform01(seed, n, 8) < outlier_frac
    ox = np.stack([width * uniform01(seed, n, 9), height * uniform01(seed, n, 10)], 1)
    x2 = np.where(is_out[:, None], ox, x2)
    sift = np.zeros(n, SIFT_DTYPE)
    sift["xpos"] = x1[:, 0].astype(np.float32)
    sift["ypos"] = x1[:, 1].astype(np.float32)
    sift["match_xpos"] = x2[:, 0].astype(np.float32)
    sift["match_ypos"] = x2[:, 1].astype(np.float32)
    sift["match"] = np.arange(n, dtype=np.int32)
    return {"K": K, "Kinv": Kinv, "sift": sift, "R": R, "t": t, "points3d": P, "outlier": is_out}


def normalized_points(scene):
    """X = Kinv [x; y; 1] in float64 -> float32 (3 x N each); convenience for set_points callers.
    (The parity tests use fillXU / the oracle's fill_xu instead, which round differently.)"""
    s = scene["sift"]
    n = len(s)
    Ki = scene["Kinv"].astype(np.float64)
    U0 = np.stack([s["xpos"], s["ypos"], np.ones(n, np.float32)]).astype(np.float64)
    U1 = np.stack([s["match_xpos"], s["match_ypos"], np.ones(n, np.float32)]).astype(np.float64)
    return (Ki @ U0).astype(np.float32), (Ki @ U1).astype(np.float32)


def descriptors(n, seed=SEED, noise=0.05, sparsity=0.0):
    """Two descriptor sets shaped like CudaSift output (non-negative, clipped at 0.2, unit L2;
    reference CudaSift/cudaSiftD.cu:390-409): set 2 = permuted set 1 + noise.  Returns
    (d1, d2, perm) with d2[i] ~ d1[perm[i]].  sparsity > 0 zeroes that fraction of the bins, which makes
    unrelated descriptors less correlated (lower ambiguity, as real SIFT on textured images)."""
    def finish(d):
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        d = np.minimum(d, 0.2)
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        return d
    raw = np.abs(normal(seed, n * 128, 20).reshape(n, 128))
    if sparsity > 0.0:
        raw = raw * (uniform01(seed, n * 128, 23).reshape(n, 128) >= sparsity) + 1e-3
    d1 = finish(raw)
    key = splitmix64(seed, n, 21)
    perm = np.argsort(key, kind="stable")
    d2 = finish(np.abs(d1[perm] + noise * normal(seed, n * 128, 22).reshape(n, 128)))
    return np.ascontiguousarray(d1, np.float32), np.ascontiguousarray(d2, np.float32), perm


def homography_scene(n, seed=SEED, noise_px=0.7, outlier_frac=0.35, width=1920, height=1080):
    """Matched SiftPoint records related by one plane-induced homography (input of FindHomography):
    xpos/ypos -> match_xpos/match_ypos, score/ambiguity set so that about 10 % fail the default gate."""
    x = (width * uniform01(seed, n, 40)).astype(np.float64)
    y = (height * uniform01(seed, n, 41)).astype(np.float64)
    H = np.array([[0.96, 0.05, 31.0], [-0.04, 1.03, -18.0], [2.0e-5, -1.0e-5, 1.0]])
    w = H[2, 0] * x + H[2, 1] * y + 1.0
    x2 = (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / w + noise_px * normal(seed, n, 42)
    y2 = (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / w + noise_px * normal(seed, n, 43)
    out = uniform01(seed, n, 44) < outlier_frac
    x2 = np.where(out, width * uniform01(seed, n, 45), x2)
    y2 = np.where(out, height * uniform01(seed, n, 46), y2)
    s = np.zeros(n, SIFT_DTYPE)
    s["xpos"], s["ypos"] = x.astype(np.float32), y.astype(np.float32)
    s["match_xpos"], s["match_ypos"] = x2.astype(np.float32), y2.astype(np.float32)
    s["score"] = (0.80 + 0.2 * uniform01(seed, n, 47)).astype(np.float32)
    s["ambiguity"] = (0.5 + 0.5 * uniform01(seed, n, 48)).astype(np.float32)
    s["match"] = np.arange(n, dtype=np.int32)
    return {"sift": s, "H": H.astype(np.float32), "outlier": out}


def image(width=640, height=480, seed=SEED, blobs=400, shift=(0.0, 0.0)):
    """8-bit-valued float image (what cv::imread(...,0).convertTo(CV_32FC1) hands to CudaImage): random
    Gaussian blobs of both polarities and a few oriented bars on a ramp -- rich in DoG extrema at every
    octave.  `shift` translates the content (second view of the same scene)."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    xx = xx - shift[0]; yy = yy - shift[1]
    img = 110.0 + 0.03 * xx + 0.02 * yy
    cx = width * uniform01(seed, blobs, 50); cy = height * uniform01(seed, blobs, 51)
    sg = 1.2 * np.exp(3.0 * uniform01(seed, blobs, 52))                # 1.2 .. 24 px
    am = 90.0 * (uniform01(seed, blobs, 53) - 0.5)
    el = 0.6 + 0.8 * uniform01(seed, blobs, 54); th = np.pi * uniform01(seed, blobs, 55)
    for i in range(blobs):
        r = int(4 * sg[i] * max(el[i], 1.0)) + 2
        x0, x1 = max(0, int(cx[i] + shift[0]) - r), min(width, int(cx[i] + shift[0]) + r + 1)
        y0, y1 = max(0, int(cy[i] + shift[1]) - r), min(height, int(cy[i] + shift[1]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx[i]; dy = yy[y0:y1, x0:x1] - cy[i]
        u = np.cos(th[i]) * dx + np.sin(th[i]) * dy; v = -np.sin(th[i]) * dx + np.cos(th[i]) * dy
        img[y0:y1, x0:x1] += am[i] * np.exp(-0.5 * ((u / (sg[i] * el[i])) ** 2 + (v / sg[i]) ** 2))
    # hard-edged discs and rotated boxes: sharp structure for the two finest octaves
    nd = 3 * blobs
    dx_ = width * uniform01(seed, nd, 56); dy_ = height * uniform01(seed, nd, 57)
    rad = 1.5 * np.exp(2.2 * uniform01(seed, nd, 58))                   # 1.5 .. 13.5 px
    amp = 30.0 + 70.0 * uniform01(seed, nd, 59)
    amp = np.where(uniform01(seed, nd, 60) < 0.5, -amp, amp)
    box = uniform01(seed, nd, 61) < 0.4; ang = np.pi * uniform01(seed, nd, 62)
    for i in range(nd):
        r = int(1.5 * rad[i]) + 3
        x0, x1 = max(0, int(dx_[i] + shift[0]) - r), min(width, int(dx_[i] + shift[0]) + r + 1)
        y0, y1 = max(0, int(dy_[i] + shift[1]) - r), min(height, int(dy_[i] + shift[1]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        ex = xx[y0:y1, x0:x1] - dx_[i]; ey = yy[y0:y1, x0:x1] - dy_[i]
        if box[i]:
            u = np.cos(ang[i]) * ex + np.sin(ang[i]) * ey; v = -np.sin(ang[i]) * ex + np.cos(ang[i]) * ey
            d = np.maximum(np.abs(u), np.abs(v) * 1.4) - rad[i]
        else:
            d = np.hypot(ex, ey) - rad[i]
        img[y0:y1, x0:x1] += amp[i] * np.clip(0.5 - d, 0.0, 1.0)           # one-pixel anti-aliased edge
    return np.clip(np.rint(img), 0, 255).astype(np.float32)


def stereo_pair(width=640, height=480, seed=SEED, disparities=(6.0, 10.0, 14.0, 18.0, 22.0, 9.0, 16.0, 12.0), blobs=400):
    """Two views of a scene made of fronto-parallel textured facets at different depths, camera moved
    along +x without rotation: every facet (a vertical strip) moves left by its own disparity
    d = f * tx / Z (integer pixels, so view 2 is an exact resampling).  Returns (img1, img2, strip_of_column, disparities).
    The textbook essential matrix of this motion is [t]_x with t = (1, 0, 0)."""
    base = image(width + 64, height, seed=seed, blobs=blobs * (width + 64) // width)
    img1 = np.ascontiguousarray(base[:, :width])
    n = len(disparities)
    edges = np.linspace(0, width, n + 1).astype(int)
    strip = np.zeros(width, np.int32)
    img2 = np.zeros_like(img1)
    for i in range(n):
        strip[edges[i]:edges[i + 1]] = i
        d = int(disparities[i])
        # the facet covers columns [edges[i], edges[i+1]) of view 1 and appears d pixels to the left in view 2
        lo, hi = max(edges[i] - d, 0), edges[i + 1] - d
        img2[:, lo:hi] = base[:, lo + d:hi + d]
    # columns never written (right of the last facet) keep the texture of view 1 shifted by the last disparity
    d = int(disparities[-1])
    img2[:, width - d:] = base[:, width:width + d]
    return img1, np.ascontiguousarray(img2), strip, np.array(disparities, np.float32)


def sift_records(desc, seed=SEED, width=720, height=576, stream=30):
    """Wrap a descriptor matrix into SiftPoint records with random keypoint positions."""
    n = desc.shape[0]
    s = np.zeros(n, SIFT_DTYPE)
    s["xpos"] = (width * uniform01(seed, n, stream)).astype(np.float32)
    s["ypos"] = (height * uniform01(seed, n, stream + 1)).astype(np.float32)
    s["data"] = desc
    s["match"] = -1
    return s
