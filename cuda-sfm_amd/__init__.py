"""cuda-sfm_amd -- MI355X-native two-view geometric estimation (match -> estimateE -> pose -> triangulate).

Python is only the harness here (tests, bench, torch.distributed plumbing).  The product is the
C-ABI library ``lib/libsfm_amd.so`` (hand-written HIP for gfx950, declared in ``include/sfm_amd.h``)
and the C++ facade in ``host/`` that keeps the reference's ``SfM::Image_pair`` / ``MatchSiftData``
call surface (reference: SfM/sfm.h:20-60, CudaSift/cudaSift.h:35-43).

There is NO CPU fallback: importing this package without the built HIP library raises.
The directory name contains a hyphen, so import it through the ``cuda_sfm_amd`` shim at the
repository root.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two flavours of one source tree (Makefile): the product library, and the lab bench (-DSFM_AB=1: the A/B switches behind
# sfm_ransac_params.reserved[], the recorded slower kernel variants, the probe / trace hooks of include/sfm_amd_ab.h).  The lab
# bench is what `import cuda_sfm_amd_ab` binds -- tests/ and profiles/ only.
AB = __name__.endswith("_ab")
# SFM_AMD_LIB_DIR: another build of the same ABI (profiles/: same-box A/B of two commits); never set by tests or bench.py itself
LIB_PATH = os.path.join(os.environ.get("SFM_AMD_LIB_DIR") or os.path.join(_HERE, "lib"), "libsfm_amd_ab.so" if AB else "libsfm_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()' "
        "or `make`); this package has no CPU fallback")

# One HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64, and a second
# runtime initialised next to it cannot see the GPU.  Import torch FIRST so that the library's
# libamdhip64.so.7 dependency binds to the copy torch already loaded (torch is this harness's device
# memory / stream / torch.distributed plumbing anyway).  Stand-alone C++ users link the system runtime.
import torch  # noqa: E402,F401

_lib = C.CDLL(LIB_PATH)
ABI_VERSION = 3                 # SFM_ABI_VERSION of include/sfm_amd.h this binding was written against
if _lib.sfm_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} reports ABI version {_lib.sfm_abi_version()}, this binding needs {ABI_VERSION}: rebuild it (`make`)")

# ---- constants (include/sfm_amd.h) --------------------------------------------------------------
OK, E_INVALID, E_HIP, E_NOMEM, E_STATE, E_SINGULAR = 0, -1, -2, -3, -4, -5
KERNEL_AUTO, KERNEL_SPLIT, KERNEL_FUSED, KERNEL_PREFILTER = 0, 1, 2, 4
if AB:
    KERNEL_MFMA = 3             # include/sfm_amd_ab.h
QUIRK_MATCH_TAIL = 1
QUIRK_MATCH_AMBIGUITY = 2
PREFILTER_PER_HYPOTHESIS, PREFILTER_PER_TILE = 2, 3      # sfm_ransac_last_prefilter_rule
MATCH_AUTO, MATCH_EXACT, MATCH_PREFILTER, MATCH_FUSED = 0, 1, 2, 3
POSE_REFERENCE, POSE_CORRECT = 0, 1
(BUF_X0, BUF_X1, BUF_U0, BUF_U1, BUF_E, BUF_P, BUF_PINV, BUF_POINTS, BUF_COUNTS, BUF_MASK, BUF_KEY,
 BUF_ECAND, BUF_PIND) = range(13)

SIFT_DTYPE = np.dtype([
    ("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("sharpness", "<f4"),
    ("edgeness", "<f4"), ("orientation", "<f4"), ("score", "<f4"), ("ambiguity", "<f4"),
    ("match", "<i4"), ("match_xpos", "<f4"), ("match_ypos", "<f4"), ("match_error", "<f4"),
    ("subsampling", "<f4"), ("empty", "<f4", (3,)), ("data", "<f4", (128,)),
])
assert SIFT_DTYPE.itemsize == 576

EXPORTS = [
    "sfm_abi_version", "sfm_last_error", "sfm_ctx_create", "sfm_ctx_destroy", "sfm_ctx_retain", "sfm_ctx_release", "sfm_ctx_set_stream", "sfm_ctx_set_quirks", "sfm_ctx_set_match_kernel", "sfm_ctx_last_match_kernel",
    "sfm_ctx_synchronize", "sfm_ctx_own_stream", "sfm_ctx_get_stream", "sfm_ctx_get_device", "sfm_ctx_timer_start", "sfm_ctx_timer_stop", "sfm_ctx_kernel_timing",
    "sfm_ctx_kernel_timing_read", "sfm_device_alloc", "sfm_device_free", "sfm_copy_to_device", "sfm_copy_to_host",
    "sfm_copy_to_host_2d", "sfm_copy_to_device_2d", "sfm_find_homography", "sfm_sift_temp_layout", "sfm_extract_sift", "sfm_extract_sift_begin", "sfm_extract_sift_end", "sfm_match", "sfm_match_soa",
    "sfm_pair_create", "sfm_pair_destroy", "sfm_pair_reset", "sfm_get_result", "sfm_fill_xu", "sfm_set_points", "sfm_ransac_default_params",
    "sfm_ransac_permutation_indices", "sfm_estimate_E", "sfm_ransac_score", "sfm_ransac_score_candidates", "sfm_ransac_score_into", "sfm_ransac_finalize",
    "sfm_ransac_finalize_key", "sfm_ransac_finalize_key_on", "sfm_ransac_score_into_slot", "sfm_estimate_E_pipelined", "sfm_pair_flush", "sfm_ransac_export_key", "sfm_pose_candidates", "sfm_choose_pose", "sfm_triangulate", "sfm_pose_chain",
    "sfm_pair_device_ptr", "sfm_pair_ld", "sfm_pair_num_points", "sfm_get_XU", "sfm_get_E", "sfm_get_best",
    "sfm_get_key", "sfm_get_inlier_counts", "sfm_get_inlier_mask", "sfm_get_E_candidates",
    "sfm_get_pose_candidates", "sfm_get_pose_inverses", "sfm_get_pose_index", "sfm_get_points", "sfm_copy_points_to_vbo",
    "sfm_ransac_last_launch", "sfm_ransac_last_clock", "sfm_ransac_last_prefilter_rule", "sfm_process_pairs", "sfm_ctx_last_pairs_batched", "sfm_extract_views", "sfm_extract_views_u8",
]
AB_EXPORTS = ["sfm_ransac_last_phases", "sfm_ransac_last_trace", "sfm_prefilter_probe", "sfm_prefilter_band_probe"]      # include/sfm_amd_ab.h
if AB:
    EXPORTS = EXPORTS + AB_EXPORTS


class RansacParams(C.Structure):
    """sfm_ransac_params (include/sfm_amd.h)."""
    _fields_ = [
        ("num_hypotheses", C.c_uint32), ("hyp_begin", C.c_uint32), ("hyp_count", C.c_uint32),
        ("seed", C.c_uint32), ("d_indices", C.c_void_p), ("threshold", C.c_float),
        ("jacobi_sweeps", C.c_int32), ("kernel", C.c_int32), ("reserved", C.c_int32 * 4),
    ]


_vp = C.c_void_p
_lib.sfm_last_error.restype = C.c_char_p
_lib.sfm_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
_lib.sfm_ctx_destroy.argtypes = [_vp]
_lib.sfm_ctx_set_stream.argtypes = [_vp, _vp]
_lib.sfm_ctx_synchronize.argtypes = [_vp]
_lib.sfm_ctx_timer_start.argtypes = [_vp]
_lib.sfm_ctx_timer_stop.argtypes = [_vp, C.POINTER(C.c_float)]
_lib.sfm_ctx_kernel_timing.argtypes = [_vp, C.c_int]
_lib.sfm_ctx_kernel_timing_read.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
class SiftLayout(C.Structure):
    """sfm_sift_layout: where every pyramid level and DoG plane lives inside the temp memory (floats)."""
    _fields_ = [("num_octaves", C.c_int32), ("width", C.c_int32 * 8), ("height", C.c_int32 * 8), ("pitch", C.c_int32 * 8),
                ("image_offset", C.c_int64 * 8), ("dog_offset", C.c_int64 * 8), ("up_offset", C.c_int64),
                ("total_floats", C.c_int64)]


_lib.sfm_sift_temp_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SiftLayout)]
_lib.sfm_extract_sift.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float,
                                  C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
_lib.sfm_extract_sift_begin.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float,
                                        C.c_int, _vp]
_lib.sfm_extract_sift_end.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
_lib.sfm_copy_points_to_vbo.argtypes = [_vp, _vp, _vp, C.c_float]
_lib.sfm_find_homography.argtypes = [_vp, _vp, C.c_int, _vp, C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_uint32, _vp, _vp, _vp]
_lib.sfm_match.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int]
_lib.sfm_match_soa.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp]
_lib.sfm_pair_create.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(_vp)]
_lib.sfm_pair_destroy.argtypes = [_vp]
_lib.sfm_fill_xu.argtypes = [_vp, _vp]
_lib.sfm_set_points.argtypes = [_vp, _vp, _vp]
_lib.sfm_ransac_default_params.argtypes = [C.POINTER(RansacParams), C.c_int]
_lib.sfm_ransac_default_params.restype = None
_lib.sfm_ransac_permutation_indices.argtypes = [_vp, C.c_int, C.c_uint32, _vp]
_lib.sfm_estimate_E.argtypes = [_vp, C.POINTER(RansacParams)]
_lib.sfm_ransac_score.argtypes = [_vp, C.POINTER(RansacParams)]
_lib.sfm_ransac_score_into.argtypes = [_vp, C.POINTER(RansacParams), _vp]
_lib.sfm_ransac_score_candidates.argtypes = [_vp, C.POINTER(RansacParams), _vp]
_lib.sfm_ransac_finalize.argtypes = [_vp, C.POINTER(RansacParams), C.c_uint32]
_lib.sfm_ransac_finalize_key.argtypes = [_vp, C.POINTER(RansacParams), _vp]
_lib.sfm_ransac_finalize_key_on.argtypes = [_vp, C.POINTER(RansacParams), _vp, _vp]
_lib.sfm_estimate_E_pipelined.argtypes = [_vp, C.POINTER(RansacParams)]
_lib.sfm_pair_flush.argtypes = [_vp]
_lib.sfm_ransac_export_key.argtypes = [_vp, _vp]
_lib.sfm_pose_candidates.argtypes = [_vp, C.c_int]
_lib.sfm_choose_pose.argtypes = [_vp, C.c_int]
_lib.sfm_triangulate.argtypes = [_vp, C.c_int]
_lib.sfm_pose_chain.argtypes = [_vp, C.c_int]
_lib.sfm_pair_device_ptr.argtypes = [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]
_lib.sfm_pair_ld.argtypes = [_vp]
_lib.sfm_pair_num_points.argtypes = [_vp]
_lib.sfm_get_XU.argtypes = [_vp, C.c_int, _vp]
_lib.sfm_get_E.argtypes = [_vp, _vp]
_lib.sfm_get_best.argtypes = [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
_lib.sfm_get_key.argtypes = [_vp, C.POINTER(C.c_uint64)]
_lib.sfm_get_inlier_counts.argtypes = [_vp, _vp, C.c_size_t]
_lib.sfm_get_inlier_mask.argtypes = [_vp, _vp]
_lib.sfm_get_E_candidates.argtypes = [_vp, _vp, C.c_size_t]
_lib.sfm_get_pose_candidates.argtypes = [_vp, _vp]
_lib.sfm_get_pose_inverses.argtypes = [_vp, _vp]
_lib.sfm_get_pose_index.argtypes = [_vp, C.POINTER(C.c_int)]
_lib.sfm_get_points.argtypes = [_vp, _vp]
_lib.sfm_pair_reset.argtypes = [_vp, C.c_int]
_lib.sfm_get_result.argtypes = [_vp, _vp]
if AB:
    _lib.sfm_ransac_last_phases.argtypes = [_vp, C.POINTER(C.c_uint64)]
    _lib.sfm_ransac_last_trace.argtypes = [_vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
_lib.sfm_ctx_last_pairs_batched.argtypes = [_vp, C.POINTER(C.c_int)]
_lib.sfm_ransac_last_launch.argtypes = [_vp] + [C.POINTER(C.c_int)] * 4
_lib.sfm_ransac_last_clock.argtypes = [_vp, C.POINTER(C.c_double)]
_lib.sfm_ransac_last_prefilter_rule.argtypes = [_vp, C.POINTER(C.c_int)]


class SfmError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = _lib.sfm_last_error()
        super().__init__(f"{where} failed with code {code}: {msg.decode() if msg else ''}")


def _check(rc, where):
    if rc != OK:
        raise SfmError(rc, where)


def lib():
    """The loaded C-ABI library (ctypes.CDLL)."""
    return _lib


def _ptr(x):
    """Device pointer of a torch tensor / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def default_params(num_points, **kw):
    p = RansacParams()
    _lib.sfm_ransac_default_params(C.byref(p), int(num_points))
    for k, v in kw.items():
        if k == "d_indices":
            v = _ptr(v)
        setattr(p, k, v)
    return p


def sift_temp_layout(width, height, num_octaves=5, scale_up=False):
    L = SiftLayout()
    _check(_lib.sfm_sift_temp_layout(int(width), int(height), int(num_octaves), int(bool(scale_up)), C.byref(L)), "sfm_sift_temp_layout")
    return L


def pack_key(count, hyp):
    """(count << 32) | (0xFFFFFFFF - hyp): max() picks the highest count, lowest id on ties
    (thrust::max_element first-maximum rule, reference sfm.cu:135-137)."""
    return (int(count) << 32) | (0xFFFFFFFF - int(hyp))


def unpack_key(key):
    key = int(key)
    return key >> 32, 0xFFFFFFFF - (key & 0xFFFFFFFF)


def shard_range(num_hypotheses, rank, world):
    """Contiguous shard of hypothesis ids owned by `rank`: [begin, begin+count)."""
    base, rem = divmod(int(num_hypotheses), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


class Context:
    """sfm_ctx: device, stream and matcher scratch."""

    def __init__(self, device=0, stream=None):
        h = _vp()
        _check(_lib.sfm_ctx_create(int(device), C.byref(h)), "sfm_ctx_create")
        self._h = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        """stream: raw hipStream_t value (e.g. torch.cuda.current_stream().cuda_stream) or None."""
        _check(_lib.sfm_ctx_set_stream(self._h, stream), "sfm_ctx_set_stream")

    def set_quirks(self, flags):
        """SFM_QUIRK_* behaviours of the reference for A/B runs (QUIRK_MATCH_TAIL: skip the last num_pts2 % 32 points;
        QUIRK_MATCH_AMBIGUITY: FindMaxCorr10's own `ambiguity`, whose merge ignores seven of the eight second-best scores)."""
        _check(_lib.sfm_ctx_set_quirks(self._h, C.c_uint(int(flags))), "sfm_ctx_set_quirks")

    def set_match_kernel(self, kernel):
        """MATCH_AUTO / MATCH_EXACT / MATCH_PREFILTER / MATCH_FUSED (bit-identical results; A/B runs and tests)."""
        _check(_lib.sfm_ctx_set_match_kernel(self._h, int(kernel)), "sfm_ctx_set_match_kernel")

    def last_match_kernel(self):
        k = C.c_int(0)
        _check(_lib.sfm_ctx_last_match_kernel(self._h, C.byref(k)), "sfm_ctx_last_match_kernel")
        return k.value

    def last_pairs_batched(self):
        """Whether the last process_pairs call on this context took the batched path."""
        b = C.c_int(0)
        _check(_lib.sfm_ctx_last_pairs_batched(self._h, C.byref(b)), "sfm_ctx_last_pairs_batched")
        return bool(b.value)

    def prefilter_probe(self, E, threshold, bound, point, survive_all=False):
        """Operands and matrix-core results of one (hypothesis, point) pair of the pre-filter kernel (test probe; lab-bench flavour only)."""
        e = np.ascontiguousarray(E, np.float32).reshape(9); pt = np.ascontiguousarray(point, np.float32).reshape(4)
        out = np.zeros(100, np.float32)
        _check(_lib.sfm_prefilter_probe(self._h, e.ctypes.data_as(_vp), C.c_float(float(threshold)), C.c_float(float(bound)), pt.ctypes.data_as(_vp),
                                        int(bool(survive_all)), out.ctypes.data_as(_vp)), "sfm_prefilter_probe")
        return {"ns": out[0:32], "ts": out[32:48], "bn": out[48:80], "bt": out[80:96], "nt": out[96], "G": out[97], "rejected": bool(out[98]), "zero_divisor_state": int(out[99])}

    def prefilter_band_probe(self, E, threshold, bound, box, b_safe, point, survive_all=False, pack=False):
        """The band rule's operands and matrix-core result of one (hypothesis, point) pair (test probe; lab-bench flavour only).
        pack: the packed scan of round 6 (sigma = 1.873 / W, `rejected` from the six-bit conversion).  pack_slots_ok / pack_bit: the
        conversion, the bit picking and the survivor table on the RAW value point[0] in each of the 32 (accumulator, step) slots."""
        e = np.ascontiguousarray(E, np.float32).reshape(9); pt = np.ascontiguousarray(point, np.float32).reshape(4)
        bx = np.ascontiguousarray(box, np.float32).reshape(8)
        out = np.zeros(104, np.float32)
        _check(_lib.sfm_prefilter_band_probe(self._h, e.ctypes.data_as(_vp), C.c_float(float(threshold)), C.c_float(float(bound)), bx.ctypes.data_as(_vp),
                                             int(bool(b_safe)) | (2 if pack else 0), pt.ctypes.data_as(_vp), int(bool(survive_all)), out.ctypes.data_as(_vp)), "sfm_prefilter_band_probe")
        return {"ns": out[0:32], "bn": out[48:80], "nt": out[96], "sigma": out[97], "rejected": bool(out[98]),
                "zero_divisor_state": int(out[99]), "second_divisor_state": int(out[100]), "pack_slots_ok": int(out[101]), "pack_bit": bool(out[102])}

    def own_stream(self):
        """Give the context a non-blocking stream of its own (for a second context next to a torch-owned one)."""
        _check(_lib.sfm_ctx_own_stream(self._h), "sfm_ctx_own_stream")

    def synchronize(self):
        _check(_lib.sfm_ctx_synchronize(self._h), "sfm_ctx_synchronize")

    def timer_start(self):
        _check(_lib.sfm_ctx_timer_start(self._h), "sfm_ctx_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        _check(_lib.sfm_ctx_timer_stop(self._h, C.byref(ms)), "sfm_ctx_timer_stop")
        return ms.value

    def kernel_timing(self, enable=True):
        _check(_lib.sfm_ctx_kernel_timing(self._h, 1 if enable else 0), "sfm_ctx_kernel_timing")

    def kernel_timing_read(self):
        """(solve_ms, score_ms, calls) summed over the RANSAC launches since the last read."""
        a = C.c_float(); b = C.c_float(); n = C.c_int()
        _check(_lib.sfm_ctx_kernel_timing_read(self._h, C.byref(a), C.byref(b), C.byref(n)), "sfm_ctx_kernel_timing_read")
        return a.value, b.value, n.value

    def match(self, d_sift1, n1, d_sift2, n2):
        """MatchSiftData core on device SiftPoint arrays (in-place field update of sift1)."""
        _check(_lib.sfm_match(self._h, _ptr(d_sift1), int(n1), _ptr(d_sift2), int(n2)), "sfm_match")

    def match_soa(self, d1, n1, ld1, d2, n2, ld2, best, second, index):
        _check(_lib.sfm_match_soa(self._h, _ptr(d1), int(n1), int(ld1), _ptr(d2), int(n2), int(ld2),
                                  _ptr(best), _ptr(second), _ptr(index)), "sfm_match_soa")

    def extract_sift(self, d_sift, max_pts, d_image, width, height, pitch, num_octaves=5, init_blur=1.0, thresh=3.0,
                     lowest_scale=0.0, scale_up=False, d_temp=None):
        """ExtractSift (cudaSiftH.cu:72-147) on device buffers -> (numPts, stored)."""
        n, st = C.c_int(), C.c_int()
        _check(_lib.sfm_extract_sift(self._h, _ptr(d_sift), int(max_pts), _ptr(d_image), int(width), int(height), int(pitch),
                                     int(num_octaves), float(init_blur), float(thresh), float(lowest_scale), int(bool(scale_up)),
                                     _ptr(d_temp), C.byref(n), C.byref(st)), "sfm_extract_sift")
        return n.value, st.value

    def extract_sift_begin(self, d_sift, max_pts, d_image, width, height, pitch, num_octaves=5, init_blur=1.0, thresh=3.0,
                           lowest_scale=0.0, scale_up=False, d_temp=None):
        """First half of extract_sift: everything enqueued on the context's stream, nothing waited for."""
        _check(_lib.sfm_extract_sift_begin(self._h, _ptr(d_sift), int(max_pts), _ptr(d_image), int(width), int(height), int(pitch),
                                           int(num_octaves), float(init_blur), float(thresh), float(lowest_scale), int(bool(scale_up)),
                                           _ptr(d_temp)), "sfm_extract_sift_begin")

    def extract_sift_end(self):
        """Second half: waits for the extraction in flight -> (numPts, stored)."""
        n, st = C.c_int(), C.c_int()
        _check(_lib.sfm_extract_sift_end(self._h, C.byref(n), C.byref(st)), "sfm_extract_sift_end")
        return n.value, st.value

    def find_homography(self, d_sift, num_pts, num_loops=1000, min_score=0.85, max_ambiguity=0.95, thresh=5.0,
                        seed=0, pts=None, want_all=False):
        """FindHomography (matching.cu:1000-1087).  Returns (H 3x3, num_matches[, counts, homo 8 x L])."""
        L = (int(num_loops) + 15) // 16 * 16
        H = np.zeros(9, np.float32); nm = C.c_int()
        counts = np.zeros(L, np.int32) if want_all else None
        homo = np.zeros((8, L), np.float32) if want_all else None
        p = None
        if pts is not None:
            p = np.ascontiguousarray(pts, np.int32)
            assert p.shape == (4, L)
        _check(_lib.sfm_find_homography(self._h, _ptr(d_sift), int(num_pts), H.ctypes.data_as(_vp), C.byref(nm), int(num_loops),
                                        float(min_score), float(max_ambiguity), float(thresh), int(seed),
                                        p.ctypes.data_as(_vp) if p is not None else None,
                                        counts.ctypes.data_as(_vp) if want_all else None,
                                        homo.ctypes.data_as(_vp) if want_all else None), "sfm_find_homography")
        return (H.reshape(3, 3), nm.value, counts, homo) if want_all else (H.reshape(3, 3), nm.value)

    def permutation_indices(self, num_points, seed, d_indices):
        _check(_lib.sfm_ransac_permutation_indices(self._h, int(num_points), int(seed), _ptr(d_indices)),
               "sfm_ransac_permutation_indices")

    def close(self):
        self._views_block = None                          # process_views' slot block and feature buffer (170 MB at 36 views of 8192 records)
        self._views_feats = None
        if getattr(self, "_h", None):
            _lib.sfm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImagePair:
    """Mirror of SfM::Image_pair (reference SfM/sfm.h:20-60) on top of the C ABI."""

    def __init__(self, ctx, K, Kinv, image_count, num_points):
        self.ctx = ctx
        k = np.ascontiguousarray(K, np.float32).reshape(9)
        ki = np.ascontiguousarray(Kinv, np.float32).reshape(9)
        h = _vp()
        _check(_lib.sfm_pair_create(ctx._h, k.ctypes.data_as(C.POINTER(C.c_float)),
                                    ki.ctypes.data_as(C.POINTER(C.c_float)), int(image_count),
                                    int(num_points), C.byref(h)), "sfm_pair_create")
        self._h = h
        self.num_points = int(num_points)
        self.ld = _lib.sfm_pair_ld(h)

    # -- reference call surface -----------------------------------------------------------------
    def reset(self, num_points):
        """Re-use the pair for another correspondence set of at most its creation-time size (no allocation)."""
        _check(_lib.sfm_pair_reset(self._h, int(num_points)), "sfm_pair_reset")
        self.num_points = int(num_points)
        self.ld = _lib.sfm_pair_ld(self._h)

    def get_result(self):
        """[E(9) | chosen pose (16) | pose index, inlier count, best hypothesis] with one synchronisation."""
        rec = np.empty(RESULT_FLOATS, np.float32)
        _check(_lib.sfm_get_result(self._h, rec.ctypes.data_as(_vp)), "sfm_get_result")
        return rec

    def fillXU(self, d_sift):
        _check(_lib.sfm_fill_xu(self._h, _ptr(d_sift)), "sfm_fill_xu")

    def set_points(self, d_X0, d_X1):
        _check(_lib.sfm_set_points(self._h, _ptr(d_X0), _ptr(d_X1)), "sfm_set_points")

    def estimateE(self, params=None):
        p = params if params is not None else default_params(self.num_points)
        _check(_lib.sfm_estimate_E(self._h, C.byref(p)), "sfm_estimate_E")

    def ransac_score(self, params, key_out=None):
        """Score the shard; with key_out (1-element int64 device tensor) its key is also left there for the all-reduce."""
        if key_out is None:
            _check(_lib.sfm_ransac_score(self._h, C.byref(params)), "sfm_ransac_score")
        else:
            _check(_lib.sfm_ransac_score_into(self._h, C.byref(params), _ptr(key_out)), "sfm_ransac_score_into")

    def last_phases(self):
        """sfm_ransac_last_phases: 8 tick values of block 0 of the last pre-filter scoring launch."""
        t = (C.c_uint64 * 8)()
        _check(_lib.sfm_ransac_last_phases(self._h, t), "sfm_ransac_last_phases")
        return list(t)

    def last_trace(self):
        """sfm_ransac_last_trace: [blocks, 20] uint64 -- start, end of wave 0, ids, (tile, column), 16 wave ends (100 MHz ticks)."""
        buf = (C.c_uint64 * (1024 * 20))()
        n = C.c_size_t()
        _check(_lib.sfm_ransac_last_trace(self._h, buf, 1024 * 20, C.byref(n)), "sfm_ransac_last_trace")
        return np.frombuffer(buf, dtype=np.uint64, count=n.value).reshape(-1, 20).copy()

    def ransac_score_candidates(self, params, d_E):
        """calculateInliers on its own: score caller-supplied candidates (float32 device tensor, 9 x hyp_count)."""
        _check(_lib.sfm_ransac_score_candidates(self._h, C.byref(params), _ptr(d_E)), "sfm_ransac_score_candidates")

    def ransac_finalize(self, params, hyp):
        _check(_lib.sfm_ransac_finalize(self._h, C.byref(params), int(hyp)), "sfm_ransac_finalize")

    def export_key(self, d_key_out):
        _check(_lib.sfm_ransac_export_key(self._h, _ptr(d_key_out)), "sfm_ransac_export_key")

    def ransac_finalize_key(self, params, d_key):
        _check(_lib.sfm_ransac_finalize_key(self._h, C.byref(params), _ptr(d_key)), "sfm_ransac_finalize_key")

    def estimateE_pipelined(self, params):
        """estimateE for a stream of calls: consecutive calls overlap on the device (two slots); flush() before the pose stages."""
        _check(_lib.sfm_estimate_E_pipelined(self._h, C.byref(params)), "sfm_estimate_E_pipelined")

    def flush(self):
        _check(_lib.sfm_pair_flush(self._h), "sfm_pair_flush")

    def ransac_finalize_key_on(self, params, d_key, hip_stream=None):
        """finalize on another stream of the device (None = the context's); E is re-derived from the hypothesis id."""
        _check(_lib.sfm_ransac_finalize_key_on(self._h, C.byref(params), _ptr(d_key), _vp(hip_stream or 0)), "sfm_ransac_finalize_key_on")

    def computePosecandidates(self, mode=POSE_REFERENCE):
        _check(_lib.sfm_pose_candidates(self._h, int(mode)), "sfm_pose_candidates")

    def choosePose(self, mode=POSE_REFERENCE):
        _check(_lib.sfm_choose_pose(self._h, int(mode)), "sfm_choose_pose")

    def linear_triangulation(self, mode=POSE_REFERENCE):
        _check(_lib.sfm_triangulate(self._h, int(mode)), "sfm_triangulate")

    def pose_chain(self, mode=POSE_REFERENCE):
        """computePosecandidates + choosePose + linear_triangulation (src/main.cpp:302-306) in one launch (REFERENCE mode)."""
        _check(_lib.sfm_pose_chain(self._h, int(mode)), "sfm_pose_chain")

    computePoseCandidates = computePosecandidates     # BASELINE.json spelling
    linearTriangulate = linear_triangulation

    # -- accessors --------------------------------------------------------------------------------
    def device_ptr(self, which):
        p = _vp(); b = C.c_size_t()
        _check(_lib.sfm_pair_device_ptr(self._h, int(which), C.byref(p), C.byref(b)), "sfm_pair_device_ptr")
        return p.value, b.value

    def _get(self, fn, name, shape, dtype):
        out = np.empty(shape, dtype)
        _check(fn(self._h, out.ctypes.data_as(_vp)), name)
        return out

    def get_XU(self, which):
        out = np.empty((3, self.num_points), np.float32)
        _check(_lib.sfm_get_XU(self._h, int(which), out.ctypes.data_as(_vp)), "sfm_get_XU")
        return out

    def get_E(self):
        return self._get(_lib.sfm_get_E, "sfm_get_E", (3, 3), np.float32)

    def get_best(self):
        h = C.c_uint32(); c = C.c_uint32()
        _check(_lib.sfm_get_best(self._h, C.byref(h), C.byref(c)), "sfm_get_best")
        return h.value, c.value

    def get_key(self):
        k = C.c_uint64()
        _check(_lib.sfm_get_key(self._h, C.byref(k)), "sfm_get_key")
        return k.value

    def get_inlier_counts(self, count):
        out = np.empty(int(count), np.int32)
        _check(_lib.sfm_get_inlier_counts(self._h, out.ctypes.data_as(_vp), out.size), "sfm_get_inlier_counts")
        return out

    def get_E_candidates(self, count):
        out = np.empty((int(count), 9), np.float32)
        _check(_lib.sfm_get_E_candidates(self._h, out.ctypes.data_as(_vp), int(count)), "sfm_get_E_candidates")
        return out

    def get_inlier_mask(self):
        return self._get(_lib.sfm_get_inlier_mask, "sfm_get_inlier_mask", (self.num_points,), np.uint8)

    def get_pose_candidates(self):
        return self._get(_lib.sfm_get_pose_candidates, "sfm_get_pose_candidates", (4, 4, 4), np.float32)

    def get_pose_inverses(self):
        return self._get(_lib.sfm_get_pose_inverses, "sfm_get_pose_inverses", (4, 4, 4), np.float32)

    def get_pose_index(self):
        i = C.c_int()
        _check(_lib.sfm_get_pose_index(self._h, C.byref(i)), "sfm_get_pose_index")
        return i.value

    def get_points(self):
        return self._get(_lib.sfm_get_points, "sfm_get_points", (4, self.num_points), np.float32)

    def copy_points_to_vbo(self, d_positions, d_velocities, scale=1.0):
        """Image_pair::copyBoidsToVBO (sfm.cu:374-383) into device buffers of 4 * num_points floats."""
        _check(_lib.sfm_copy_points_to_vbo(self._h, _ptr(d_positions), _ptr(d_velocities), float(scale)), "sfm_copy_points_to_vbo")

    def last_launch(self):
        v = [C.c_int() for _ in range(4)]
        _check(_lib.sfm_ransac_last_launch(self._h, *[C.byref(x) for x in v]), "sfm_ransac_last_launch")
        r = C.c_int()
        _check(_lib.sfm_ransac_last_prefilter_rule(self._h, C.byref(r)), "sfm_ransac_last_prefilter_rule")
        return {"kernel": v[0].value, "grid": v[1].value, "block": v[2].value, "lds_bytes": v[3].value, "prefilter_rule": r.value}

    def last_clock_mhz(self):
        v = C.c_double()
        _check(_lib.sfm_ransac_last_clock(self._h, C.byref(v)), "sfm_ransac_last_clock")
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            _lib.sfm_pair_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def estimate_E_distributed(pair, params, rank, world, key_tensor, all_reduce_max):
    """Multi-GPU estimateE: every rank scores its contiguous shard of hypothesis ids,
    ONE all-reduce(max) of the packed 8-byte key picks the winner, every rank finalizes it locally
    (bit-identical E and mask on every rank, no second collective, no host round trip).

    key_tensor: 1-element int64 device tensor; all_reduce_max(key_tensor) reduces it in place
    across ranks (torch.distributed.all_reduce(MAX) -> RCCL over xGMI; gloo in the CPU tests).
    """
    begin, count = shard_range(params.num_hypotheses, rank, world)
    params.hyp_begin, params.hyp_count = begin, count     # count == 0 only when begin == H
    pair.ransac_score(params, key_out=key_tensor)         # empty shard -> key 0; the key lands in key_tensor too (no export step)
    all_reduce_max(key_tensor)
    pair.ransac_finalize_key(params, key_tensor)


# ---- RCCL exchange step in C (include/sfm_amd_comm.h, libsfm_amd_rccl.so) ----------------------------
COMM_EXPORTS = ["sfm_comm_unique_id", "sfm_comm_init", "sfm_comm_destroy", "sfm_comm_rank", "sfm_comm_nccl_ranks", "sfm_estimate_E_sharded",
                "sfm_estimate_E_sharded_pipelined", "sfm_comm_flush", "sfm_process_views_sharded", "sfm_process_views_sharded_u8", "sfm_comm_last_exchange",
                "sfm_comm_exchange_only"]
# (SFM_AMD_COMM_LIB: tests only -- tests/fake_ccl/libsfm_amd_fakeccl.so is the same comm.cpp linked against a shared-memory stand-in for
#  RCCL, so that two ranks can run on the ONE GPU of a test box: tests/test_gpu_fakeccl.py)
COMM_LIB_PATH = os.environ.get("SFM_AMD_COMM_LIB") or os.path.join(os.path.dirname(LIB_PATH), "libsfm_amd_rccl.so")
COMM_ID_BYTES = 128
_comm_lib = None


def comm_lib():
    """libsfm_amd_rccl.so, loaded on first use (it pulls in librccl; the core library does not)."""
    global _comm_lib
    if AB:
        raise ImportError("libsfm_amd_rccl.so is linked against the PRODUCT library: the lab-bench flavour has no communicator")
    if _comm_lib is None:
        if not os.path.exists(COMM_LIB_PATH):
            raise ImportError(f"{COMM_LIB_PATH} is missing: run `make` (or __graft_entry__.build())")
        L = C.CDLL(COMM_LIB_PATH)
        L.sfm_comm_unique_id.argtypes = [_vp]
        L.sfm_comm_init.argtypes = [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]
        L.sfm_comm_destroy.argtypes = [_vp]
        L.sfm_comm_rank.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sfm_estimate_E_sharded.argtypes = [_vp, C.POINTER(RansacParams), _vp]
        L.sfm_estimate_E_sharded_pipelined.argtypes = [_vp, C.POINTER(RansacParams), _vp]
        L.sfm_comm_flush.argtypes = [_vp]
        L.sfm_comm_nccl_ranks.argtypes = [_vp, C.POINTER(C.c_int)]
        L.sfm_comm_last_exchange.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.sfm_process_views_sharded.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.POINTER(C.c_float)), C.c_int, C.c_int, C.c_int,
                                                C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.c_uint32,
                                                C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.sfm_process_views_sharded_u8.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.POINTER(C.c_ubyte)), C.c_int, C.c_int, C.c_int,
                                                   C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.c_uint32,
                                                   C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.sfm_comm_exchange_only.argtypes = [_vp, C.POINTER(RansacParams), _vp]
        _comm_lib = L
    return _comm_lib


class Comm:
    """sfm_comm: one RCCL communicator over the ranks of the job, bound to a Context (its device and stream).
    `unique_id` = the 128 bytes of Comm.unique_id() made on rank 0 and handed to every rank out of band."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(comm_lib().sfm_comm_unique_id(buf), "sfm_comm_unique_id")
        return buf.raw

    def __init__(self, ctx, unique_id, rank, world):
        assert len(unique_id) == COMM_ID_BYTES
        self._ctx = ctx
        h = _vp()
        _check(comm_lib().sfm_comm_init(ctx._h, C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES), int(rank), int(world), C.byref(h)),
               "sfm_comm_init")
        self._h = h
        self.rank, self.world = int(rank), int(world)

    def estimate_E(self, pair, params):
        """estimateE over all ranks (params.num_hypotheses = global count): shard, ONE all-reduce(max), local finalize."""
        _check(comm_lib().sfm_estimate_E_sharded(pair._h, C.byref(params), self._h), "sfm_estimate_E_sharded")

    def estimate_E_pipelined(self, pair, params):
        """The same step with its exchange + finalize on the communicator's own stream, so that the next call's scoring
        overlaps this call's all-reduce; flush() before reading results."""
        _check(comm_lib().sfm_estimate_E_sharded_pipelined(pair._h, C.byref(params), self._h), "sfm_estimate_E_sharded_pipelined")

    def exchange_only(self, pair, params):
        """The exchange step alone (all-reduce of the pair's current key + finalize): diagnostics, bench.py's exchange_us."""
        _check(comm_lib().sfm_comm_exchange_only(pair._h, C.byref(params), self._h), "sfm_comm_exchange_only")

    def flush(self):
        _check(comm_lib().sfm_comm_flush(self._h), "sfm_comm_flush")

    def process_views(self, images, K, Kinv, pairs=None, max_pts=8192, sift=None, num_hypotheses=None, pose_mode=POSE_REFERENCE):
        """sfm_process_views_sharded: BASELINE configs[4] over all ranks inside the C libraries (counts all-gather, count-sized
        grouped broadcasts of the features, records all-gather).
        Returns ({pair_id: record}, counts) like process_views; every rank gets every record."""
        sift = dict(sift or {})
        V = len(images)
        pairs = ring_pairs(V) if pairs is None else list(pairs)
        u8 = all(getattr(im, "dtype", None) == np.uint8 for im in images)           # 8-bit grey images go through sfm_process_views_sharded_u8
        imgs = [np.ascontiguousarray(im, np.uint8 if u8 else np.float32) for im in images]
        h, w = imgs[0].shape
        ctype = C.c_ubyte if u8 else C.c_float
        ptrs = (C.POINTER(ctype) * V)(*[im.ctypes.data_as(C.POINTER(ctype)) for im in imgs])
        entry = comm_lib().sfm_process_views_sharded_u8 if u8 else comm_lib().sfm_process_views_sharded
        pij = np.ascontiguousarray(np.array(pairs, np.int32).reshape(-1))
        rec = np.empty((max(len(pairs), 1), 28), np.float32)
        counts = (C.c_int * V)()
        k = np.ascontiguousarray(K, np.float32).reshape(9); ki = np.ascontiguousarray(Kinv, np.float32).reshape(9)
        _check(entry(self._h, k.ctypes.data_as(C.POINTER(C.c_float)), ki.ctypes.data_as(C.POINTER(C.c_float)), ptrs, V, w, h,
                     pij.ctypes.data_as(C.POINTER(C.c_int)), len(pairs), int(max_pts), int(sift.get("num_octaves", 5)),
                     C.c_double(float(sift.get("init_blur", 1.0))), C.c_float(float(sift.get("thresh", 3.0))), C.c_float(float(sift.get("lowest_scale", 0.0))),
                     int(bool(sift.get("scale_up", False))), int(num_hypotheses or 0), int(pose_mode),
                     rec.ctypes.data_as(C.POINTER(C.c_float)), counts), "sfm_process_views_sharded")
        return {pid: rec[pid].copy() for pid in range(len(pairs)) if rec[pid][26] >= 0}, list(counts)

    def last_exchange(self):
        """(bytes the feature exchange of the last process_views moved into this rank, bytes max_pts-sized slots would have been)."""
        a, b = C.c_uint64(), C.c_uint64()
        _check(comm_lib().sfm_comm_last_exchange(self._h, C.byref(a), C.byref(b)), "sfm_comm_last_exchange")
        return a.value, b.value

    def nccl_ranks(self):
        n = C.c_int()
        _check(comm_lib().sfm_comm_nccl_ranks(self._h, C.byref(n)), "sfm_comm_nccl_ranks")
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            comm_lib().sfm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- many view pairs (BASELINE configs[4]: 36-view ring, view pairs streamed across the GPUs) -------
RESULT_FLOATS = 9 + 16 + 3          # E, chosen pose (4x4), [pose index, inlier count, best hypothesis]


def pair_schedule(num_pairs, rank, world):
    """Static round-robin assignment of pair ids to ranks: rank r owns r, r+world, ..."""
    return list(range(int(rank), int(num_pairs), int(world)))


class PairDesc(C.Structure):
    """sfm_pair_desc (include/sfm_amd.h)."""
    _fields_ = [("d_sift1", C.c_void_p), ("n1", C.c_int), ("d_sift2", C.c_void_p), ("n2", C.c_int)]


_lib.sfm_extract_views.argtypes = [_vp, C.POINTER(C.POINTER(C.c_float)), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_int,
                                   C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int)]
_lib.sfm_extract_views_u8.argtypes = [_vp, C.POINTER(C.POINTER(C.c_ubyte)), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_int,
                                      C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int)]
_lib.sfm_process_pairs.argtypes = [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(PairDesc), C.c_int, C.c_int, C.c_int,
                                   C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]


PAIR_DESC_DTYPE = np.dtype({"names": ["d_sift1", "n1", "d_sift2", "n2"], "formats": [np.uint64, np.int32, np.uint64, np.int32],
                            "offsets": [PairDesc.d_sift1.offset, PairDesc.n1.offset, PairDesc.d_sift2.offset, PairDesc.n2.offset],
                            "itemsize": C.sizeof(PairDesc)})


def process_pairs_local(ctx, descs, K, Kinv, rank=0, world=1, num_hypotheses=None, pose_mode=POSE_REFERENCE):
    """sfm_process_pairs: the pairs rank, rank + world, ... of `descs` through the C library -- per pair MatchSiftData
    (when a second view is given), fillXU, estimateE, pose candidates, choosePose, linear triangulation, enqueued back to
    back, ONE read-back.  descs: sequence of (d_sift1, n1) [already matched] or (d_sift1, n1, d_sift2, n2), or a numpy array
    of PAIR_DESC_DTYPE (device addresses as integers: what process_views builds for hundreds of pairs without a Python loop).
    Returns (records [owned, 28] float32, status [owned] int32) in list order of the owned pairs."""
    if isinstance(descs, np.ndarray):
        assert descs.dtype == PAIR_DESC_DTYPE and descs.flags.c_contiguous
        arr = descs.ctypes.data_as(C.POINTER(PairDesc))
    else:
        arr = (PairDesc * max(1, len(descs)))()
        for i, d in enumerate(descs):
            arr[i].d_sift1 = _ptr(d[0]); arr[i].n1 = int(d[1])
            arr[i].d_sift2 = _ptr(d[2]) if len(d) > 2 and d[2] is not None else None
            arr[i].n2 = int(d[3]) if len(d) > 3 else 0
    owned = len(range(int(rank), len(descs), int(world)))
    rec = np.full((max(owned, 1), 28), -1.0, np.float32)
    status = np.zeros(max(owned, 1), np.int32)
    k = np.ascontiguousarray(K, np.float32).reshape(9); ki = np.ascontiguousarray(Kinv, np.float32).reshape(9)
    _check(_lib.sfm_process_pairs(ctx._h, k.ctypes.data_as(C.POINTER(C.c_float)), ki.ctypes.data_as(C.POINTER(C.c_float)), arr, len(descs),
                                  int(rank), int(world), int(num_hypotheses or 0), int(pose_mode),
                                  rec.ctypes.data_as(C.POINTER(C.c_float)), status.ctypes.data_as(C.POINTER(C.c_int))), "sfm_process_pairs")
    return rec[:owned], status[:owned]


def process_pairs(ctx, pairs, K, Kinv, rank=0, world=1, num_hypotheses=None, pose_mode=POSE_REFERENCE,
                  all_gather=None, device=None):
    """Task-parallel two-view estimation over many view pairs: each rank runs the whole per-pair
    pipeline (fillXU -> estimateE -> pose candidates -> choosePose -> linear triangulation) for the
    pairs it owns through sfm_process_pairs (C, no per-pair host work); NO per-pair collective.  Results are fixed-size
    records [E(9) | P(16) | pose_index, inliers, best_hypothesis] gathered ONCE at the end.

    pairs: sequence of (d_sift, n) device SiftPoint arrays (already matched: match_xpos/ypos filled) or
    (d_sift1, n1, d_sift2, n2) to have MatchSiftData run first;
    all_gather(local_tensor) -> gathered tensor [world * max_local, RESULT_FLOATS] (torch.distributed
    all_gather_into_tensor over RCCL; identity for world == 1).  Returns {pair_id: record ndarray}.
    """
    import torch
    mine = pair_schedule(len(pairs), rank, world)
    max_local = (len(pairs) + world - 1) // world
    rec = np.full((max(max_local, 1), RESULT_FLOATS + 1), -1.0, np.float32)     # last column: pair id (-1 = empty slot)
    if mine:
        r, status = process_pairs_local(ctx, pairs, K, Kinv, rank, world, num_hypotheses, pose_mode)
        for slot, pid in enumerate(mine):
            if status[slot] == 0 or status[slot] == E_SINGULAR:
                rec[slot, :RESULT_FLOATS] = r[slot]
                rec[slot, RESULT_FLOATS] = pid
    local = torch.from_numpy(rec)
    if device is not None:
        local = local.to(device)
    gathered = all_gather(local) if (all_gather is not None and world > 1) else local
    g = gathered.cpu().numpy().reshape(-1, RESULT_FLOATS + 1)
    return {int(r[RESULT_FLOATS]): r[:RESULT_FLOATS].copy() for r in g if r[RESULT_FLOATS] >= 0}


def ring_pairs(num_views):
    """Neighbouring views of a closed ring: (0,1), (1,2), ..., (V-1,0) -- BASELINE configs[4]."""
    return [(i, (i + 1) % num_views) for i in range(num_views)] if num_views > 2 else [(0, 1)][:max(0, num_views - 1)]


def view_slot(v, world, slots):
    """Row of view v in the gathered feature tensor: rank v % world holds it in its local slot v // world."""
    return (int(v) % int(world)) * int(slots) + int(v) // int(world)


def exchange_view_features(block, num_views, rank, world, max_pts, dist=None, cache=None):
    """The feature exchange of the many-views front end, sized by what exists.  block: this rank's uint8 tensor
    [slots, max_pts * 576 + 64] (slot s = view rank + s * world: its SiftPoint records, then its int32 feature count in the tail).
    dist: torch.distributed (RCCL on the GPUs, gloo in the CPU tests) or None for a single rank.
      1. all_gather_into_tensor of the `slots` counts of every rank (4 bytes per view);
      2. every view's count x 576 bytes broadcast from its owner (rank v % world) into ONE compact buffer, views back to back in
         view order (offsets are multiples of 576: descriptors stay 16-byte aligned); a view without features ships nothing.
    Returns (feats, counts, offsets, stats): feats = flat uint8 tensor, view v's records start at byte offsets[v];
    stats = {"feature_bytes": bytes moved into this rank, "slot_bytes": what an all-gather of the max_pts-sized slots moved}.
    A single rank exchanges nothing: feats is the block itself, offsets are its slot starts."""
    import torch
    rec_bytes = int(max_pts) * 576
    slots = block.shape[0]
    num_views, rank, world = int(num_views), int(rank), int(world)
    mine = block[:, rec_bytes:rec_bytes + 4].contiguous().view(torch.int32).reshape(-1)
    if world == 1 or dist is None:
        c = mine.cpu().numpy()
        counts = [int(c[v]) for v in range(num_views)]
        return block.reshape(-1), counts, [v * (rec_bytes + 64) for v in range(num_views)], {"feature_bytes": 0, "slot_bytes": 0}
    allc = torch.empty(world * slots, dtype=torch.int32, device=block.device)
    dist.all_gather_into_tensor(allc, mine)
    c = allc.cpu().numpy()
    counts = [int(c[(v % world) * slots + v // world]) for v in range(num_views)]
    assert all(0 <= n <= int(max_pts) for n in counts), counts
    offsets, total = [], 0
    for n in counts:
        offsets.append(total)
        total += n * 576
    # the compact buffer is kept on `cache` (an object with a _views_feats attribute: the Context) and grown on demand, as comm.cpp
    # does with its d_feats: a fresh 37 MB allocation per call of a 5 ms job shows up as occasional slow steps
    buf = getattr(cache, "_views_feats", None) if cache is not None else None
    if buf is None or buf.numel() < max(total, 1) or buf.device != block.device:
        buf = torch.empty(max(total, 1) + ((max(total, 1) >> 3) if cache is not None else 0), dtype=torch.uint8, device=block.device)
        if cache is not None:
            cache._views_feats = buf
    feats = buf[:max(total, 1)]
    work = []
    for v in range(num_views):
        nb = counts[v] * 576
        if nb == 0:
            continue
        dst = feats[offsets[v]:offsets[v] + nb]
        if v % world == rank:
            dst.copy_(block[v // world, :nb])
        work.append(dist.broadcast(dst, src=v % world, async_op=True))
    for w in work:
        w.wait()
    return feats, counts, offsets, {"feature_bytes": total + 4 * world * slots, "slot_bytes": world * slots * (rec_bytes + 64)}


def process_views(ctx, images, K, Kinv, pairs=None, rank=0, world=1, max_pts=8192, sift=None, num_hypotheses=None,
                  pose_mode=POSE_REFERENCE, dist=None, gather_results=None, device=None, stats=None):
    """Many-view front end of process_pairs: images -> ExtractSift per view (views round-robin over the ranks) ->
    the count-sized feature exchange (exchange_view_features) -> per pair MatchSiftData + the two-view pipeline on the rank that
    owns the pair -> ONE gather of fixed-size result records.

    images: list of equally sized 2-D float32 arrays (grey values 0..255).  sift: dict of ExtractSift arguments
    (num_octaves, init_blur, thresh, lowest_scale, scale_up).  dist: torch.distributed for world > 1 (None: single rank);
    gather_results(float tensor) is an all_gather_into_tensor wrapper (identity when world == 1).  stats (optional dict)
    receives the exchange's byte counts.
    Returns ({pair_id: record}, counts) with record = [E(9) | P(16) | pose index, inliers, best hypothesis] as in
    process_pairs and counts = number of features of every view."""
    import torch
    dev = device if device is not None else torch.device("cuda", ctx.device)
    sift = dict(sift or {})
    V = len(images)
    pairs = ring_pairs(V) if pairs is None else list(pairs)
    h, w = images[0].shape
    p = (w + 127) // 128 * 128
    slots = (V + world - 1) // world
    rec_bytes = max_pts * 576
    # records | int32 count in the tail; every slot of a single-rank run is written by sfm_extract_views (records up to the count,
    # the count itself), so only a multi-rank run -- whose last slots may stay unused -- pays for clearing 4.7 MB per view
    # (kept on the context between calls: 36 slots of 8192 records are 170 MB, and a fresh allocation per call shows up as
    # occasional 10-40 ms steps of a 5 ms job)
    # Re-used when it is at least as large as this call needs (alternating view counts / max_pts do not re-allocate); Context.close() drops it.
    need = slots * (rec_bytes + 64)
    cache = getattr(ctx, "_views_block", None)
    if cache is None or cache.numel() < need or cache.device != dev:
        cache = torch.zeros(need, dtype=torch.uint8, device=dev)
        ctx._views_block = cache
    block = cache[:need].view(slots, rec_bytes + 64)
    if world > 1:
        block[:, rec_bytes:].zero_()                      # the counts of slots this rank does not fill must read 0
    # ExtractSift for this rank's views inside the C library (sfm_extract_views: pinned staging, two streams)
    mine_views = list(range(rank, V, world))
    if mine_views:
        # 8-bit images (all of them uint8 arrays) go through sfm_extract_views_u8: a quarter of the PCIe traffic, same features
        u8 = all(getattr(images[v], "dtype", None) == np.uint8 for v in range(V))
        imgs = [np.ascontiguousarray(images[v], np.uint8 if u8 else np.float32) for v in range(V)]
        assert all(im.shape == (h, w) for im in imgs), "process_views needs equally sized images"
        ctype = C.c_ubyte if u8 else C.c_float
        ptrs = (C.POINTER(ctype) * V)(*[im.ctypes.data_as(C.POINTER(ctype)) for im in imgs])
        cnts = (C.c_int * len(mine_views))()
        fn = _lib.sfm_extract_views_u8 if u8 else _lib.sfm_extract_views
        _check(fn(ctx._h, ptrs, V, w, h, int(rank), int(world), _ptr(block), rec_bytes + 64, int(max_pts),
                  int(sift.get("num_octaves", 5)), float(sift.get("init_blur", 1.0)), float(sift.get("thresh", 3.0)),
                  float(sift.get("lowest_scale", 0.0)), int(bool(sift.get("scale_up", False))), cnts), "sfm_extract_views")
    # the counts of all views (one small all-gather + read-back), then every view's own bytes from its owner
    feats, counts, offsets, xstats = exchange_view_features(block, V, rank, world, max_pts, dist, cache=ctx)
    if stats is not None:
        stats.update(xstats)

    mine = pair_schedule(len(pairs), rank, world)
    max_local = (len(pairs) + world - 1) // world
    rec = np.full((max(max_local, 1), RESULT_FLOATS + 1), -1.0, np.float32)
    # per pair MatchSiftData + the two-view pipeline, all inside sfm_process_pairs (C): MatchSiftData writes its result
    # fields (score .. match_ypos) straight into the first view's records -- nothing else of a record changes, so no copy.
    # The descriptor array is built with numpy from the views' device addresses (630 pairs: a Python loop over torch slices
    # used to keep the GPU idle for milliseconds between the extraction and the first matcher launch).
    vptr = np.array([feats.data_ptr() + offsets[v] for v in range(V)], np.uint64)
    vcnt = np.array(counts, np.int32)
    pij = np.asarray(pairs, np.int64).reshape(-1, 2)
    descs = np.zeros(len(pairs), PAIR_DESC_DTYPE)
    if len(pairs):
        descs["d_sift1"] = vptr[pij[:, 0]]; descs["n1"] = vcnt[pij[:, 0]]
        descs["d_sift2"] = vptr[pij[:, 1]]; descs["n2"] = vcnt[pij[:, 1]]
    if mine:
        r, status = process_pairs_local(ctx, descs, K, Kinv, rank, world, num_hypotheses, pose_mode)
        ok = (status == 0) | (status == E_SINGULAR)
        rec[:len(mine), :RESULT_FLOATS] = np.where(ok[:, None], r, -1.0)
        rec[:len(mine), RESULT_FLOATS] = np.where(ok, np.asarray(mine, np.float32), -1.0)
    if gather_results is not None and world > 1:
        g = gather_results(torch.from_numpy(rec).to(dev)).cpu().numpy().reshape(-1, RESULT_FLOATS + 1)
    else:
        g = rec
    ids = g[:, RESULT_FLOATS]
    return {int(i): g[k, :RESULT_FLOATS] for k, i in enumerate(ids) if i >= 0}, counts
