"""GPU: ExtractSift (reference CudaSift/cudaSiftH.cu:72-232 + cudaSiftD.cu).
 - the oracle's image kernels against the REFERENCE'S OWN ScaleDown / ScaleUp / LowPassBlock /
   LaplaceMultiMem compiled for gfx950 from cudaSiftD.cu in place (oracle/_ref) -- bit-exact;
 - the product (sfm_extract_sift through the C ABI) against the oracle: every pyramid level, every DoG
   plane and every field of every record, bit for bit, in the same (deterministic) order."""
import ctypes as C

import numpy as np
import pytest

import cuda_sfm_amd as S
import oracle as O
from cuda_sfm_amd_synth import synth
from helpers import same_bits

pytestmark = pytest.mark.gpu
FIELDS = ("xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")


def align(a, b=128):
    return (a + b - 1) // b * b


def padded(img, pitch):
    out = np.zeros((img.shape[0], pitch), np.float32)
    out[:, :img.shape[1]] = img
    return out


@pytest.fixture(scope="module")
def R():
    if not O.ref_available("libref_kernels.so"):
        pytest.skip("oracle/_ref/libref_kernels.so not built")
    lib = O.ref_lib("libref_kernels.so")
    f, i = O.f32p, C.c_int
    lib.refk_sift_lowpass.argtypes = [f, f, i, i, i, f]
    lib.refk_sift_scaledown.argtypes = [f, i, i, i, f, i, f]
    lib.refk_sift_scaleup.argtypes = [f, i, i, i, f, i]
    lib.refk_sift_laplace.argtypes = [f, i, i, i, f, i, f]
    return lib


def fp(a):
    return a.ctypes.data_as(O.f32p)


@pytest.mark.parametrize("w,h", [(640, 480), (301, 203), (130, 67)])
def test_oracle_matches_reference_image_kernels(gpu, R, w, h):
    img = synth.image(w, h, seed=5 + w, blobs=60)
    p = align(w)
    src = padded(img, p)
    # LowPassBlock
    for blur in (1.0, 1.5, 0.001):
        k9 = O.sift_lowpass_taps(blur)
        got = np.zeros_like(src)
        assert R.refk_sift_lowpass(fp(src), fp(got), w, p, h, fp(k9)) == 0
        assert same_bits(got[:, :w], O.sift_lowpass(img, k9))
    # ScaleDown
    kt, k5 = O.sift_tables(5)
    p2 = align(w // 2)
    got = np.zeros((h // 2, p2), np.float32)
    assert R.refk_sift_scaledown(fp(src), w, p, h, fp(got), p2, fp(k5)) == 0
    assert same_bits(got[:, :w // 2], O.sift_scaledown(img, k5))
    # ScaleUp
    pu = align(2 * w)
    got = np.zeros((2 * h, pu), np.float32)
    assert R.refk_sift_scaleup(fp(src), w, p, h, fp(got), pu) == 0
    assert same_bits(got[:, :2 * w], O.sift_scaleup(img))
    # LaplaceMultiMem, every octave slot of the table
    low = O.sift_lowpass(img, O.sift_lowpass_taps(1.0))
    lsrc = padded(low, p)
    for octave in (5, 3, 1):
        got = np.zeros((7, h, p), np.float32)
        assert R.refk_sift_laplace(fp(lsrc), w, p, h, fp(got), octave, fp(kt)) == 0
        assert same_bits(got[:, :, :w], O.sift_laplace(low, kt.reshape(8, 192)[octave][:128]))


def run_product(gpu, img, max_pts=32768, **kw):
    torch, dev, ctx = gpu
    h, w = img.shape
    p = align(w)
    d_img = torch.from_numpy(padded(img, p)).to(dev)
    d_sift = torch.zeros((max_pts, 576), dtype=torch.uint8, device=dev)
    L = S.sift_temp_layout(w, h, kw.get("num_octaves", 5), kw.get("scale_up", False))
    d_temp = torch.zeros(L.total_floats, dtype=torch.float32, device=dev)
    n, stored = ctx.extract_sift(d_sift, max_pts, d_img, w, h, p, d_temp=d_temp, **kw)
    rec = d_sift.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)
    return rec, n, stored, L, d_temp.cpu().numpy()


@pytest.mark.parametrize("w,h,octaves", [(640, 480, 5), (301, 203, 4), (130, 67, 2)])
def test_pyramid_and_dog_match_oracle(gpu, w, h, octaves):
    img = synth.image(w, h, seed=9 + w, blobs=80)
    _, _, _, L, temp = run_product(gpu, img, num_octaves=octaves, init_blur=1.0, thresh=3.0)
    kt, k5 = O.sift_tables(octaves)
    level = O.sift_lowpass(img, O.sift_lowpass_taps(1.0))
    for l in range(octaves):
        wl, hl, pl = L.width[l], L.height[l], L.pitch[l]
        assert (wl, hl) == (w >> l, h >> l) and pl == align(wl)
        got = temp[L.image_offset[l]: L.image_offset[l] + pl * hl].reshape(hl, pl)[:, :wl]
        assert same_bits(got, level), f"pyramid level {l}"
        dog = temp[L.dog_offset[l]: L.dog_offset[l] + 7 * pl * hl].reshape(7, hl, pl)[:, :, :wl]
        assert same_bits(dog, O.sift_laplace(level, kt.reshape(8, 192)[octaves - l][:128])), f"DoG level {l}"
        if l + 1 < octaves:
            level = O.sift_scaledown(level, k5)


@pytest.mark.parametrize("w,h,kw", [
    (640, 480, dict(num_octaves=5, init_blur=1.0, thresh=3.0)),
    (301, 203, dict(num_octaves=4, init_blur=1.5, thresh=1.0)),          # main.cpp:270-276 settings
    (320, 240, dict(num_octaves=3, init_blur=1.0, thresh=2.0, scale_up=True)),
    (400, 300, dict(num_octaves=5, init_blur=1.0, thresh=3.0, lowest_scale=2.5)),
    (130, 67, dict(num_octaves=1, init_blur=0.0, thresh=1.5)),
    (1920, 1080, dict(num_octaves=5, init_blur=1.0, thresh=3.0)),       # the size of the reference's README table
])
def test_extract_matches_oracle(gpu, w, h, kw):
    img = synth.image(w, h, seed=21 + w, blobs=min(2500, max(40, w * h // 800)))
    rec, n, stored, _, _ = run_product(gpu, img, **kw)
    opts, on, ostored = O.extract_sift(img, kw.get("num_octaves", 5), kw.get("init_blur", 1.0), kw.get("thresh", 3.0),
                                       kw.get("lowest_scale", 0.0), kw.get("scale_up", False))
    assert (n, stored) == (on, ostored)
    assert n > 20, "scene too poor to test anything"
    for f in FIELDS:
        assert same_bits(rec[f][:stored], opts[f][:stored]), f
    assert not rec["data"][stored:].any()                              # nothing written past the stored records
    assert not np.isnan(rec["data"][:stored]).any()


def test_max_pts_clips_like_the_reference(gpu):
    img = synth.image(640, 480, seed=3, blobs=500)
    full, n_full, stored_full, _, _ = run_product(gpu, img, num_octaves=5, thresh=2.0)
    assert n_full > 1500
    cap = 1000
    rec, n, stored, _, _ = run_product(gpu, img, max_pts=cap, num_octaves=5, thresh=2.0)
    assert n == cap and stored == cap                                   # cudaSiftH.cu:124
    # coarse octaves come first and fit: identical to the uncapped run up to the octave that overflows
    sub = full["subsampling"][:cap]
    last_full_octave = sub[np.flatnonzero(np.diff(sub))[-1]] if np.any(np.diff(sub)) else None
    keep = np.flatnonzero(full["subsampling"][:cap] >= last_full_octave)
    for f in ("xpos", "ypos", "scale", "orientation"):
        assert same_bits(rec[f][keep], full[f][keep]), f


def test_deterministic_and_stream_reuse(gpu):
    img = synth.image(512, 384, seed=8)
    a, na, sa, _, _ = run_product(gpu, img, thresh=2.0)
    b, nb, sb, _, _ = run_product(gpu, img, thresh=2.0)
    assert (na, sa) == (nb, sb) and a.tobytes() == b.tobytes()
    torch, dev, ctx = gpu
    p = align(512)
    d_img = torch.from_numpy(padded(img, p)).to(dev)
    d_sift = torch.zeros((32768, 576), dtype=torch.uint8, device=dev)
    n, stored = ctx.extract_sift(d_sift, 32768, d_img, 512, 384, p, thresh=2.0)     # context-owned temp memory
    c = d_sift.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)
    assert (n, stored) == (na, sa) and c[:stored].tobytes() == a[:stored].tobytes()


def test_argument_checks(gpu):
    torch, dev, ctx = gpu
    d_img = torch.zeros((64, 128), dtype=torch.float32, device=dev)
    d_sift = torch.zeros((64, 576), dtype=torch.uint8, device=dev)
    with pytest.raises(S.SfmError):
        ctx.extract_sift(d_sift, 64, d_img, 100, 64, 128, num_octaves=8)
    with pytest.raises(S.SfmError):
        ctx.extract_sift(d_sift, 64, d_img, 100, 64, 64)                # pitch < width
    n, stored = ctx.extract_sift(d_sift, 64, d_img, 100, 64, 128)       # flat image: nothing found
    assert (n, stored) == (0, 0)


@pytest.mark.parametrize("w,h,octaves", [(40, 30, 5), (17, 9, 3), (64, 4, 2), (129, 65, 7), (1, 1, 1)])
def test_tiny_and_degenerate_sizes(gpu, w, h, octaves):
    """levels shrink to a pixel or vanish (w >> l == 0): pyramid, DoG planes and records still equal the oracle's"""
    rng = np.random.default_rng(w * 131 + h)
    img = np.rint(rng.uniform(0, 255, (h, w))).astype(np.float32)
    rec, n, stored, L, temp = run_product(gpu, img, num_octaves=octaves, init_blur=1.0, thresh=0.5)
    opts, on, ostored = O.extract_sift(img, octaves, 1.0, 0.5)
    assert (n, stored) == (on, ostored)
    for f in FIELDS:
        assert same_bits(rec[f][:stored], opts[f][:stored]), f
    kt, k5 = O.sift_tables(octaves)
    level = O.sift_lowpass(img, O.sift_lowpass_taps(1.0))
    for l in range(octaves):
        wl, hl, pl = L.width[l], L.height[l], L.pitch[l]
        if wl == 0 or hl == 0:
            break
        got = temp[L.image_offset[l]: L.image_offset[l] + pl * hl].reshape(hl, pl)[:, :wl]
        assert same_bits(got, level), f"pyramid level {l}"
        dog = temp[L.dog_offset[l]: L.dog_offset[l] + 7 * pl * hl].reshape(7, hl, pl)[:, :, :wl]
        assert same_bits(dog, O.sift_laplace(level, kt.reshape(8, 192)[octaves - l][:128])), f"DoG level {l}"
        if l + 1 < octaves and wl // 2 > 0 and hl // 2 > 0:
            level = O.sift_scaledown(level, k5)


def test_noise_image_many_points_and_capacity(gpu):
    """white noise at a tiny threshold: thousands of extrema per octave.  With room for them the result equals
    the oracle; with max_pts far below the number of raw extrema (the internal stash overflows and the exact
    two-pass path runs) the first max_pts records of the full result survive -- deterministic, unlike the
    reference, which keeps an arbitrary subset (cudaSiftD.cu:1421)."""
    rng = np.random.default_rng(5)
    img = np.rint(rng.uniform(0, 255, (240, 320))).astype(np.float32)
    rec, n, stored, _, _ = run_product(gpu, img, num_octaves=3, init_blur=1.0, thresh=0.05)
    opts, on, ostored = O.extract_sift(img, 3, 1.0, 0.05)
    assert (n, stored) == (on, ostored) and n > 3000
    for f in FIELDS:
        assert same_bits(rec[f][:stored], opts[f][:stored]), f
    first_fine = int(np.argmax(rec["subsampling"][:stored] == 1.0))
    for cap in (2000, 700, 100, 8):
        crec, cn, cstored, _, _ = run_product(gpu, img, max_pts=cap, num_octaves=3, init_blur=1.0, thresh=0.05)
        ocrec, ocn, ocstored = O.extract_sift(img, 3, 1.0, 0.05, max_pts=cap)
        assert (cn, cstored) == (cap, cap) == (ocn, ocstored)
        for f in FIELDS:
            assert same_bits(crec[f][:cap], ocrec[f][:cap]), (cap, f)
        # octaves that fit are complete; the octave that overflows keeps its first points in (y, x, scale) order and no
        # secondary orientations (they would start after ALL of the octave's points, cudaSiftD.cu:1041)
        k = min(cap, first_fine)
        assert same_bits(crec["data"][:k], rec["data"][:k])


def test_randomised_configurations(gpu):
    """a sweep over image sizes, octave counts, blurs, thresholds, lowest scales and the upsampling switch, on
    images of three kinds (blobs + hard shapes, white noise, smooth ramps with a few blobs): every record equals
    the oracle's"""
    rng = np.random.default_rng(2026)
    seen = 0
    for trial in range(14):
        w, h = int(rng.integers(33, 420)), int(rng.integers(33, 320))
        kind = trial % 3
        if kind == 0:
            img = synth.image(w, h, seed=100 + trial, blobs=max(10, w * h // 1500))
        elif kind == 1:
            img = np.rint(rng.uniform(0, 255, (h, w))).astype(np.float32)
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.rint(60 + 0.2 * xx + 0.3 * yy + 40 * np.sin(xx / 7.0) * np.cos(yy / 5.0)).astype(np.float32)
        kw = dict(num_octaves=int(rng.integers(1, 6)), init_blur=float(rng.choice([0.0, 1.0, 1.5, 2.0])),
                  thresh=float(rng.choice([0.3, 1.0, 2.0, 3.5])), lowest_scale=float(rng.choice([0.0, 0.0, 1.3, 3.0])),
                  scale_up=bool(rng.integers(0, 4) == 0))
        if kw["scale_up"] and w * h > 60000:
            kw["scale_up"] = False
        rec, n, stored, _, _ = run_product(gpu, img, **kw)
        opts, on, ostored = O.extract_sift(img, kw["num_octaves"], kw["init_blur"], kw["thresh"], kw["lowest_scale"], kw["scale_up"])
        assert (n, stored) == (on, ostored), (trial, w, h, kw)
        for f in FIELDS:
            assert same_bits(rec[f][:stored], opts[f][:stored]), (trial, w, h, kw, f)
        seen += stored
    assert seen > 5000


@pytest.mark.parametrize("w,h,octave,thresh,lowest", [(640, 480, 5, 2.0, 0.0), (301, 203, 3, 1.0, 0.0), (200, 150, 5, 3.0, 1.3)])
def test_oracle_detector_matches_reference_findpointsmulti(gpu, R, w, h, octave, thresh, lowest):
    """FindPointsMulti (cudaSiftD.cu:1433-1574) is the detector the reference launches when it is built with
    MANAGEDMEM (cudaSiftH.cu:508-510): the same extremum test, edge test and sub-pixel refinement as the default
    FindPointsMultiNew -- which cannot be built for gfx950 -- compacting candidates with a shared atomic
    instead of warp votes.  Run on the MI355X it must find exactly the oracle's points; every field but
    `scale` (device powf / exp2f vs. the oracle's table / polynomial) bit for bit."""
    R.refk_sift_findpoints.argtypes = [O.f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                       C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    img = synth.image(w, h, seed=31 + w, blobs=max(60, w * h // 900))
    low = O.sift_lowpass(img, O.sift_lowpass_taps(1.0))
    kt, _ = O.sift_tables(5)
    dog = O.sift_laplace(low, kt.reshape(8, 192)[octave][:128])
    p = align(w)
    dogp = np.zeros((7, h, p), np.float32); dogp[:, :, :w] = dog
    max_pts = 32768
    out = np.zeros(max_pts, O.SIFT_DTYPE); cnt = C.c_int(0)
    sub = 2.0
    assert R.refk_sift_findpoints(fp(dogp), w, p, h, sub, lowest / sub, thresh, 0.2, 10.0, octave, max_pts, out.ctypes.data_as(C.c_void_p), C.byref(cnt)) == 0
    ref = out[:cnt.value]
    opts, ocnt = O.sift_find_points(dog, sub, lowest / sub, thresh, max_pts)
    assert cnt.value == ocnt and ocnt > 100
    # the reference appends with atomicInc (arbitrary order): compare as sets keyed by the refined position
    ro = np.lexsort((ref["scale"], ref["xpos"], ref["ypos"])); oo = np.lexsort((opts["scale"], opts["xpos"], opts["ypos"]))
    ref, opts = ref[ro], opts[oo]
    for f in ("xpos", "ypos", "sharpness", "edgeness", "subsampling"):
        assert same_bits(ref[f], opts[f]), f
    assert np.abs(ref["scale"] / opts["scale"] - 1.0).max() < 4e-7


def test_begin_end_two_contexts_overlap_same_results(gpu):
    """sfm_extract_sift_begin / _end: two contexts (the second on a stream of its own) extract two images at once; records
    and counts are those of the plain calls; misuse is reported (end without begin, second begin)."""
    torch, dev, ctx = gpu
    w, h = 384, 288
    imgs = [synth.image(w, h, seed=61, blobs=150), synth.image(w, h, seed=62, blobs=90)]
    kw = dict(num_octaves=4, init_blur=1.0, thresh=2.0)
    plain = [run_product(gpu, im, max_pts=4096, **kw) for im in imgs]
    ctx2 = S.Context(0)
    ctx2.own_stream()
    p = align(w)
    d_img = [torch.from_numpy(padded(im, p)).to(dev) for im in imgs]
    d_out = [torch.zeros((4096, 576), dtype=torch.uint8, device=dev) for _ in imgs]
    L = S.sift_temp_layout(w, h, 4, False)
    d_tmp = [torch.zeros(L.total_floats, dtype=torch.float32, device=dev) for _ in imgs]
    torch.cuda.synchronize()
    with pytest.raises(S.SfmError) as e:
        ctx2.extract_sift_end()
    assert e.value.code == S.E_STATE
    for _ in range(3):
        ctx.extract_sift_begin(d_out[0], 4096, d_img[0], w, h, p, d_temp=d_tmp[0], **kw)
        ctx2.extract_sift_begin(d_out[1], 4096, d_img[1], w, h, p, d_temp=d_tmp[1], **kw)
        with pytest.raises(S.SfmError) as e:
            ctx2.extract_sift_begin(d_out[1], 4096, d_img[1], w, h, p, d_temp=d_tmp[1], **kw)
        assert e.value.code == S.E_STATE
        got = [ctx.extract_sift_end(), ctx2.extract_sift_end()]
        for k in (0, 1):
            rec, n, stored = plain[k][0], plain[k][1], plain[k][2]
            assert got[k] == (n, stored) and n > 30
            mine = d_out[k].cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)
            for f in FIELDS:
                assert same_bits(mine[f][:stored], rec[f][:stored]), (k, f)
    ctx2.close()
