"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (cuda-sfm_amd/) never does.  See oracle/sfm_oracle.h for what is restated and
how it is pinned to the reference.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsfm_oracle.so")
_REF = os.path.join(_HERE, "_ref")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
u8p = C.POINTER(C.c_uint8)

SIFT_DTYPE = np.dtype([
    ("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("sharpness", "<f4"),
    ("edgeness", "<f4"), ("orientation", "<f4"), ("score", "<f4"), ("ambiguity", "<f4"),
    ("match", "<i4"), ("match_xpos", "<f4"), ("match_ypos", "<f4"), ("match_error", "<f4"),
    ("subsampling", "<f4"), ("empty", "<f4", (3,)), ("data", "<f4", (128,)),
])
assert SIFT_DTYPE.itemsize == 576

POSE_REFERENCE = 0
POSE_CORRECT = 1


def _fp(a):
    return a.ctypes.data_as(f32p)


def _ip(a):
    return a.ctypes.data_as(i32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _load():
    if not os.path.exists(_LIB):
        raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(_LIB)
    L.orc_det_ref.restype = C.c_float
    L.orc_det3.restype = C.c_float
    L.orc_residual.restype = C.c_float
    L.orc_residual.argtypes = [f32p] + [C.c_float] * 6
    L.orc_hash32.restype = C.c_uint32
    L.orc_hash32.argtypes = [C.c_uint32]
    L.orc_pack_key.restype = C.c_uint64
    L.orc_pack_key.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_ransac_range.restype = C.c_uint64
    L.orc_ransac_range.argtypes = [f32p, f32p, C.c_int, C.c_uint32, C.c_uint32, i32p, C.c_uint32,
                                   C.c_float, C.c_int, i32p, f32p, C.c_int]
    L.orc_ransac_range_fast.restype = C.c_uint64
    L.orc_ransac_range_fast.argtypes = L.orc_ransac_range.argtypes
    L.orc_count_inliers_fast.restype = C.c_int
    L.orc_count_inliers_fast.argtypes = [f32p, f32p, f32p, C.c_int, C.c_float]
    L.orc_count_inliers.restype = C.c_int
    L.orc_count_inliers.argtypes = [f32p, f32p, f32p, C.c_int, C.c_float, u8p]
    L.orc_choose_pose.restype = C.c_int
    L.orc_inv4.restype = C.c_int
    L.orc_tri_A.argtypes = [C.c_float] * 4 + [f32p, f32p, f32p]
    return L


_L = _load()


def svd3(a):
    a = _f32(a).reshape(9)
    u, s, v = (np.empty(9, np.float32) for _ in range(3))
    _L.orc_svd3(_fp(a), _fp(u), _fp(s), _fp(v))
    return u.reshape(3, 3), s.reshape(3, 3), v.reshape(3, 3)


def _mm(fn, a, b):
    a = _f32(a).reshape(9); b = _f32(b).reshape(9)
    m = np.empty(9, np.float32)
    fn(_fp(a), _fp(b), _fp(m))
    return m.reshape(3, 3)


def multAB(a, b): return _mm(_L.orc_multAB, a, b)
def multAtB(a, b): return _mm(_L.orc_multAtB, a, b)
def multABt(a, b): return _mm(_L.orc_multABt, a, b)


def det_ref(a):
    a = _f32(a).reshape(9)
    return float(_L.orc_det_ref(_fp(a)))


def det3(a):
    a = _f32(a).reshape(9)
    return float(_L.orc_det3(_fp(a)))


def fill_xu(pts, kinv):
    pts = np.ascontiguousarray(pts, dtype=SIFT_DTYPE)
    n = pts.shape[0]
    kinv = _f32(kinv).reshape(9)
    out = [np.empty((3, n), np.float32) for _ in range(4)]
    _L.orc_fill_xu(pts.ctypes.data_as(C.c_void_p), C.c_int(n), _fp(kinv), *[_fp(o) for o in out])
    return out  # U0, U1, X0, X1


def sample8(seed, hyp, n):
    idx = np.empty(8, np.int32)
    _L.orc_sample8(C.c_uint32(seed), C.c_uint32(hyp), C.c_int(n), _ip(idx))
    return idx


def build_A(X0, X1, idx):
    X0 = _f32(X0); X1 = _f32(X1); idx = np.ascontiguousarray(idx, np.int32)
    A = np.empty(72, np.float32)
    _L.orc_build_A(_fp(X0), _fp(X1), C.c_int(X0.shape[1]), _ip(idx), _fp(A))
    return A.reshape(8, 9)


def AtA9(A):
    A = _f32(A).reshape(72)
    S = np.empty(81, np.float32)
    _L.orc_AtA9(_fp(A), _fp(S))
    return S.reshape(9, 9)


def jacobi9(S, sweeps):
    S = _f32(S).reshape(81).copy()
    V = np.eye(9, dtype=np.float32).reshape(81).copy()
    _L.orc_jacobi9(_fp(S), _fp(V), C.c_int(sweeps))
    return S.reshape(9, 9), V.reshape(9, 9)


def nullvec9(A, sweeps):
    A = _f32(A).reshape(72)
    e = np.empty(9, np.float32)
    _L.orc_nullvec9(_fp(A), C.c_int(sweeps), _fp(e))
    return e


def normalizeE(E):
    E = _f32(E).reshape(9).copy()
    _L.orc_normalizeE(_fp(E))
    return E.reshape(3, 3)


def residual(E, x1, x2):
    E = _f32(E).reshape(9)
    return float(_L.orc_residual(_fp(E), *[C.c_float(float(v)) for v in (*x1, *x2)]))


def count_inliers(E, X0, X1, thr, want_mask=True):
    E = _f32(E).reshape(9); X0 = _f32(X0); X1 = _f32(X1)
    n = X0.shape[1]
    mask = np.empty(n, np.uint8) if want_mask else None
    c = _L.orc_count_inliers(_fp(E), _fp(X0), _fp(X1), C.c_int(n), C.c_float(thr),
                             mask.ctypes.data_as(u8p) if want_mask else None)
    return int(c), mask


def hypothesis_E(X0, X1, idx, sweeps):
    X0 = _f32(X0); X1 = _f32(X1); idx = np.ascontiguousarray(idx, np.int32)
    E = np.empty(9, np.float32)
    _L.orc_hypothesis_E(_fp(X0), _fp(X1), C.c_int(X0.shape[1]), _ip(idx), C.c_int(sweeps), _fp(E))
    return E.reshape(3, 3)


def pack_key(count, hyp):
    return int(_L.orc_pack_key(C.c_uint32(count), C.c_uint32(hyp)))


def unpack_key(key):
    return int(key >> 32), int(0xFFFFFFFF - (key & 0xFFFFFFFF))


def ransac_range(X0, X1, h0, count, thr, sweeps, seed=0, indices=None, want_counts=True,
                 want_E=False, nthreads=0):
    """Scores hypotheses [h0, h0+count); returns (key, counts|None, Ecand|None)."""
    X0 = _f32(X0); X1 = _f32(X1)
    n = X0.shape[1]
    counts = np.empty(count, np.int32) if want_counts else None
    Ec = np.empty((count, 9), np.float32) if want_E else None
    ind = None
    if indices is not None:
        indices = np.ascontiguousarray(indices, np.int32)
        ind = _ip(indices)
    key = _L.orc_ransac_range(_fp(X0), _fp(X1), n, h0, count, ind, seed, thr, sweeps,
                              _ip(counts) if want_counts else None,
                              _fp(Ec) if want_E else None, nthreads)
    return int(key), counts, Ec


def ransac_range_fast(X0, X1, h0, count, thr, sweeps, seed=0, indices=None, want_counts=True,
                      want_E=False, nthreads=0):
    """ransac_range through the vectorised scoring loop of sfm_oracle_fast.c (bench.py's cpu_baseline): same results."""
    X0 = _f32(X0); X1 = _f32(X1)
    n = X0.shape[1]
    counts = np.empty(count, np.int32) if want_counts else None
    Ec = np.empty((count, 9), np.float32) if want_E else None
    ind = None
    if indices is not None:
        indices = np.ascontiguousarray(indices, np.int32)
        ind = _ip(indices)
    key = _L.orc_ransac_range_fast(_fp(X0), _fp(X1), n, h0, count, ind, seed, thr, sweeps,
                                   _ip(counts) if want_counts else None,
                                   _fp(Ec) if want_E else None, nthreads)
    return int(key), counts, Ec


def count_inliers_fast(E, X0, X1, thr):
    E = _f32(E).reshape(9); X0 = _f32(X0); X1 = _f32(X1)
    return int(_L.orc_count_inliers_fast(_fp(E), _fp(X0), _fp(X1), X0.shape[1], thr))


def pose_candidates(E, mode=POSE_REFERENCE):
    E = _f32(E).reshape(9)
    P = np.empty(64, np.float32)
    _L.orc_pose_candidates(_fp(E), C.c_int(mode), _fp(P))
    return P.reshape(4, 4, 4)


def tri_A(x1, y1, x2, y2, m1, m2):
    m1 = _f32(m1).reshape(16); m2 = _f32(m2).reshape(16)
    A = np.empty(16, np.float32)
    _L.orc_tri_A(x1, y1, x2, y2, _fp(m1), _fp(m2), _fp(A))
    return A.reshape(4, 4)


def nullvec4(A, sweeps):
    A = _f32(A).reshape(16)
    v = np.empty(4, np.float32)
    _L.orc_nullvec4(_fp(A), C.c_int(sweeps), _fp(v))
    return v


def normalize_pt(v):
    v = _f32(v).reshape(4)
    o = np.empty(4, np.float32)
    _L.orc_normalize_pt(_fp(v), _fp(o))
    return o


def inv4(m):
    m = _f32(m).reshape(16)
    o = np.zeros(16, np.float32)
    ok = _L.orc_inv4(_fp(m), _fp(o))
    return bool(ok), o.reshape(4, 4)


def choose_pose(X0, X1, P, mode=POSE_REFERENCE, sweeps=8):
    X0 = _f32(X0); X1 = _f32(X1); P = _f32(P).reshape(64)
    Pinv = np.empty(64, np.float32); d1 = np.empty(16, np.float32); d2 = np.empty(16, np.float32)
    ind = _L.orc_choose_pose(_fp(X0), _fp(X1), C.c_int(X0.shape[1]), _fp(P), C.c_int(mode),
                             C.c_int(sweeps), _fp(Pinv), _fp(d1), _fp(d2))
    return int(ind), Pinv.reshape(4, 4, 4), d1.reshape(4, 4), d2.reshape(4, 4)


def triangulate(X0, X1, Pm, sweeps=8):
    X0 = _f32(X0); X1 = _f32(X1); Pm = _f32(Pm).reshape(16)
    n = X0.shape[1]
    out = np.empty((4, n), np.float32)
    _L.orc_triangulate(_fp(X0), _fp(X1), C.c_int(n), _fp(Pm), C.c_int(sweeps), _fp(out))
    return out


def match_desc(d1, d2, nthreads=0):
    d1 = _f32(d1); d2 = _f32(d2)
    n1, n2 = d1.shape[0], d2.shape[0]
    best = np.empty(n1, np.float32); sec = np.empty(n1, np.float32); idx = np.empty(n1, np.int32)
    _L.orc_match_desc(_fp(d1), C.c_int(n1), C.c_int(d1.shape[1]), _fp(d2), C.c_int(n2),
                      C.c_int(d2.shape[1]), _fp(best), _fp(sec), _ip(idx), C.c_int(nthreads))
    return best, sec, idx


def match_second_ref(d1, d2, nthreads=0):
    """FindMaxCorr10's own bookkeeping (matching.cu:361-390): (best, its approximate second best, index) -- see sfm_oracle.c."""
    d1 = _f32(d1); d2 = _f32(d2)
    n1, n2 = d1.shape[0], d2.shape[0]
    best = np.empty(n1, np.float32); sec = np.empty(n1, np.float32); idx = np.empty(n1, np.int32)
    _L.orc_match_second_ref(_fp(d1), C.c_int(n1), C.c_int(d1.shape[1]), _fp(d2), C.c_int(n2), C.c_int(d2.shape[1]),
                            _fp(best), _fp(sec), _ip(idx), C.c_int(nthreads))
    return best, sec, idx


def match_sift(s1, s2, nthreads=0):
    s1 = np.ascontiguousarray(s1, dtype=SIFT_DTYPE).copy()
    s2 = np.ascontiguousarray(s2, dtype=SIFT_DTYPE)
    _L.orc_match_sift(s1.ctypes.data_as(C.c_void_p), C.c_int(len(s1)),
                      s2.ctypes.data_as(C.c_void_p), C.c_int(len(s2)), C.c_int(nthreads))
    return s1


def homography4(coord, pts):
    coord = _f32(coord); pts = np.ascontiguousarray(pts, np.int32)
    h = np.empty(8, np.float32)
    _L.orc_homography4(_fp(coord), C.c_int(coord.shape[1]), _ip(pts), _fp(h))
    return h


def homography_count(h, coord, n, thresh2):
    coord = _f32(coord); h = _f32(h).reshape(8)
    _L.orc_homography_count.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_float]
    return int(_L.orc_homography_count(_fp(h), _fp(coord), coord.shape[1], int(n), float(thresh2)))


def find_homography(sift, num_loops=1000, min_score=0.85, max_ambiguity=0.95, thresh=5.0, seed=0, want_all=False):
    """FindHomography end to end (matching.cu:1000-1087) -> (H 3x3, numMatches[, counts, homo 8 x L])."""
    sift = np.ascontiguousarray(sift)
    L = (int(num_loops) + 15) // 16 * 16
    H = np.empty(9, np.float32)
    counts = np.empty(L, np.int32); homo = np.empty((8, L), np.float32)
    _L.orc_find_homography.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32, f32p, i32p, f32p]
    nm = int(_L.orc_find_homography(sift.ctypes.data_as(C.c_void_p), len(sift), int(num_loops), float(min_score), float(max_ambiguity),
                                    float(thresh), int(seed), _fp(H), _ip(counts), _fp(homo)))
    return (H.reshape(3, 3), nm, counts, homo) if want_all else (H.reshape(3, 3), nm)


def improve_homography(sift, H, num_loops, min_score, max_ambiguity, thresh):
    """numpy restatement of ImproveHomography (CudaSift/geomFuncs.cpp:6-72): double normal equations,
    float residuals, 0/1 weights, Cholesky solve.  The reference solves with OpenCV's
    cv::solve(DECOMP_CHOLESKY) (absent here -> PARITY UNPINNED at that call; compared to ~1e-9).
    Returns (numfit, H 3x3 float32, match_error float32[n])."""
    f32, f64 = np.float32, np.float64
    x, y = sift["xpos"].astype(f32), sift["ypos"].astype(f32)
    mx, my = sift["match_xpos"].astype(f32), sift["match_ypos"].astype(f32)
    H = np.asarray(H, f32).reshape(9)
    A = (H[:8] / H[8]).astype(f64)
    limit = f32(thresh) * f32(thresh)
    gate = ~((sift["score"] < f32(min_score)) | (sift["ambiguity"] > f32(max_ambiguity)))

    def errs_exact(A):
        xd, yd = x.astype(f64), y.astype(f64)
        den = (A[6] * xd + A[7] * yd + 1.0).astype(f32)
        dx = ((A[0] * xd + A[1] * yd + A[2]) / den.astype(f64) - mx.astype(f64)).astype(f32)
        dy = ((A[3] * xd + A[4] * yd + A[5]) / den.astype(f64) - my.astype(f64)).astype(f32)
        err = (dx * dx + dy * dy).astype(f32)
        return err

    for _ in range(int(num_loops)):
        err = errs_exact(A)
        w = ((err < limit) & gate).astype(f64)
        xd, yd = x.astype(f64), y.astype(f64)
        z, o = np.zeros_like(xd), np.ones_like(xd)
        Y1 = np.stack([xd, yd, o, z, z, z, (-x * mx).astype(f64), (-y * mx).astype(f64)])
        Y2 = np.stack([z, z, z, xd, yd, o, (-x * my).astype(f64), (-y * my).astype(f64)])
        M = (Y1 * w) @ Y1.T + (Y2 * w) @ Y2.T
        X = (Y1 * w) @ mx.astype(f64) + (Y2 * w) @ my.astype(f64)
        try:
            L = np.linalg.cholesky(M)
            A = np.linalg.solve(L.T, np.linalg.solve(L, X))
        except np.linalg.LinAlgError:
            A = np.zeros(8)                                      # cv::solve zeroes dst on failure
    err = errs_exact(A)
    out = np.append(A, 1.0).astype(f32).reshape(3, 3)
    return int((err < limit).sum()), out, np.sqrt(err).astype(f32)


# ---- ExtractSift (sift_oracle.c) --------------------------------------------------------------------
def _img(a):
    a = np.ascontiguousarray(a, np.float32)
    assert a.ndim == 2
    return a


def sift_tables(num_octaves):
    """(laplace kernel table float[8*12*16], scaledown taps[5]) as the reference's host code builds them."""
    kt = np.zeros(8 * 12 * 16, np.float32); k5 = np.zeros(5, np.float32)
    _L.orc_sift_laplace_kernels.argtypes = [C.c_int, C.c_float, f32p]
    _L.orc_sift_laplace_kernels(int(num_octaves), 0.0, _fp(kt))
    _L.orc_sift_scaledown_kernel.argtypes = [C.c_float, f32p]
    _L.orc_sift_scaledown_kernel(0.5, _fp(k5))
    return kt, k5


def sift_lowpass_taps(scale):
    k = np.zeros(9, np.float32)
    _L.orc_sift_lowpass_kernel.argtypes = [C.c_float, f32p]
    _L.orc_sift_lowpass_kernel(float(scale), _fp(k))
    return k


def sift_lowpass(img, taps):
    img = _img(img); out = np.zeros_like(img)
    _L.orc_sift_lowpass.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]
    _L.orc_sift_lowpass(_fp(img), img.shape[1], img.shape[0], img.shape[1], _fp(out), img.shape[1], _fp(_f32(taps)))
    return out


def sift_scaledown(img, taps):
    img = _img(img); out = np.zeros((img.shape[0] // 2, img.shape[1] // 2), np.float32)
    _L.orc_sift_scaledown.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]
    _L.orc_sift_scaledown(_fp(img), img.shape[1], img.shape[0], img.shape[1], _fp(out), out.shape[1], _fp(_f32(taps)))
    return out


def sift_scaleup(img):
    img = _img(img); out = np.zeros((img.shape[0] * 2, img.shape[1] * 2), np.float32)
    _L.orc_sift_scaleup.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
    _L.orc_sift_scaleup(_fp(img), img.shape[1], img.shape[0], img.shape[1], _fp(out), out.shape[1])
    return out


def sift_laplace(img, kern8x16):
    """-> DoG planes (7, h, w)"""
    img = _img(img); h, w = img.shape
    out = np.zeros((7, h, w), np.float32)
    _L.orc_sift_laplace.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]
    _L.orc_sift_laplace(_fp(img), w, h, w, _fp(out), w, _fp(_f32(kern8x16)))
    return out


def sift_find_points(dog, subsampling, lowest_scale, thresh, max_pts=32768, factor=0.2, edge_limit=10.0):
    dog = np.ascontiguousarray(dog, np.float32); _, h, w = dog.shape
    pts = np.zeros(max_pts, SIFT_DTYPE); cnt = C.c_int(0)
    _L.orc_sift_find_points.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_void_p, C.POINTER(C.c_int), C.c_int]
    _L.orc_sift_find_points(_fp(dog), w, h, w, float(subsampling), float(lowest_scale), float(thresh), float(factor),
                            float(edge_limit), pts.ctypes.data_as(C.c_void_p), C.byref(cnt), int(max_pts))
    return pts[:min(cnt.value, max_pts)], cnt.value


def sift_orientation(img, x, y, scale):
    img = _img(img); ori = np.zeros(2, np.float32)
    _L.orc_sift_orientation.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, f32p]
    two = _L.orc_sift_orientation(_fp(img), img.shape[1], img.shape[0], img.shape[1], float(x), float(y), float(scale), _fp(ori))
    return ori[:2] if two else ori[:1]


def sift_descriptor(img, x, y, scale, orientation):
    img = _img(img); d = np.zeros(128, np.float32)
    _L.orc_sift_descriptor.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, f32p]
    _L.orc_sift_descriptor(_fp(img), img.shape[1], img.shape[0], img.shape[1], float(x), float(y), float(scale), float(orientation), _fp(d))
    return d


def sift_math(name, *args):
    """elementary functions of the extractor: exp2f, expf, atan2f, fast_atan2f, tex (img, x, y), sincosf"""
    if name == "sincosf":
        s, c = C.c_float(), C.c_float()
        _L.orc_sift_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _L.orc_sift_sincosf(float(args[0]), C.byref(s), C.byref(c))
        return np.float32(s.value), np.float32(c.value)
    if name == "tex":
        img = _img(args[0])
        _L.orc_sift_tex.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]; _L.orc_sift_tex.restype = C.c_float
        return np.float32(_L.orc_sift_tex(_fp(img), img.shape[1], img.shape[1], img.shape[0], float(args[1]), float(args[2])))
    f = getattr(_L, "orc_sift_" + name)
    f.argtypes = [C.c_float] * len(args); f.restype = C.c_float
    return np.float32(f(*[float(a) for a in args]))


def extract_sift(image, num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=0.0, scale_up=False, max_pts=32768):
    """ExtractSift (cudaSiftH.cu:72-147) -> (records[:numPts], numPts, stored) ; stored >= numPts counts the
    records that carry a descriptor (incl. the finest octave's secondary orientations)."""
    image = _img(image); h, w = image.shape
    pts = np.zeros(max_pts, SIFT_DTYPE); stored = C.c_int(0)
    _L.orc_extract_sift.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int,
                                    C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    n = _L.orc_extract_sift(_fp(image), w, h, w, int(num_octaves), float(init_blur), float(thresh), float(lowest_scale),
                            int(bool(scale_up)), pts.ctypes.data_as(C.c_void_p), int(max_pts), C.byref(stored))
    return pts[:stored.value], int(n), int(stored.value)


# ---- in-place builds of the reference (oracle/_ref), optional ---------------------------------
def ref_available(name):
    return os.path.exists(os.path.join(_REF, name))


def ref_lib(name):
    path = os.path.join(_REF, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built (needs /root/reference; run make -C oracle)")
    L = C.CDLL(path)
    if hasattr(L, "ref_det"):
        L.ref_det.restype = C.c_float
    return L
