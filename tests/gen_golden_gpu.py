#!/usr/bin/env python3
"""Generates tests/golden/ref_gpu_kernels.npz ON THE GPU BOX: inputs + outputs of the REFERENCE'S OWN
CUDA kernels compiled for gfx950 from /root/reference in place (oracle/ref_build_gpu.sh ->
oracle/_ref/libref_kernels.so, which travels with the snapshot) and run on the MI355X:

  CudaSift/cudaSiftD.cu   LowPassBlock, ScaleDown, ScaleUp, LaplaceMultiMem, FindPointsMulti  (-ffp-contract=off)
  CudaSift/matching.cu    ComputeHomographies + InvertMatrix<8>, TestHomographies  (-ffp-contract=off)

    gpurun -- 'python tests/gen_golden_gpu.py gpurun_out/ref_gpu_kernels.npz'
then copy the file to tests/golden/.  Only data is stored (seeded inputs, kernel outputs); the CPU suite
checks the oracle against it bit for bit (tests/test_oracle_golden.py), so the pinning holds in
containers that have neither a GPU nor the reference checkout."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from cuda_sfm_amd_synth import synth

f32p, i32p = O.f32p, O.i32p


def fp(a):
    return a.ctypes.data_as(f32p)


def align(a, b=128):
    return (a + b - 1) // b * b


def padded(img, pitch):
    out = np.zeros((img.shape[0], pitch), np.float32)
    out[:, :img.shape[1]] = img
    return out


def main(path):
    R = O.ref_lib("libref_kernels.so")
    i = C.c_int
    R.refk_sift_lowpass.argtypes = [f32p, f32p, i, i, i, f32p]
    R.refk_sift_scaledown.argtypes = [f32p, i, i, i, f32p, i, f32p]
    R.refk_sift_scaleup.argtypes = [f32p, i, i, i, f32p, i]
    R.refk_sift_laplace.argtypes = [f32p, i, i, i, f32p, i, f32p]
    R.refk_homography.argtypes = [f32p, i, i32p, i, C.c_float, f32p, i32p]
    out = {}

    w, h = 150, 77                                            # ragged: not a multiple of any tile size
    img = synth.image(w, h, seed=77, blobs=40)
    p = align(w)
    src = padded(img, p)
    out["sift_image"] = img
    for tag, blur in (("lp10", 1.0), ("lp15", 1.5)):
        k9 = O.sift_lowpass_taps(blur)
        got = np.zeros_like(src)
        assert R.refk_sift_lowpass(fp(src), fp(got), w, p, h, fp(k9)) == 0
        out["sift_" + tag + "_taps"] = k9; out["sift_" + tag] = got[:, :w].copy()
    kt, k5 = O.sift_tables(5)
    out["sift_laplace_table"] = kt; out["sift_scaledown_taps"] = k5
    p2 = align(w // 2)
    got = np.zeros((h // 2, p2), np.float32)
    assert R.refk_sift_scaledown(fp(src), w, p, h, fp(got), p2, fp(k5)) == 0
    out["sift_scaledown"] = got[:, :w // 2].copy()
    pu = align(2 * w)
    got = np.zeros((2 * h, pu), np.float32)
    assert R.refk_sift_scaleup(fp(src), w, p, h, fp(got), pu) == 0
    out["sift_scaleup"] = got[:, :2 * w].copy()
    low = padded(out["sift_lp10"], p)
    for octave in (5, 2):
        got = np.zeros((7, h, p), np.float32)
        assert R.refk_sift_laplace(fp(low), w, p, h, fp(got), octave, fp(kt)) == 0
        out[f"sift_dog_octave{octave}"] = got[:, :, :w].copy()

    # FindPointsMulti (the MANAGEDMEM-path detector) on the octave-5 DoG stack
    R.refk_sift_findpoints.argtypes = [f32p, i, i, i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i, i, C.c_void_p, C.POINTER(C.c_int)]
    dogp = np.zeros((7, h, p), np.float32); dogp[:, :, :w] = out["sift_dog_octave5"]
    rec = np.zeros(4096, O.SIFT_DTYPE); cnt = C.c_int(0)
    assert R.refk_sift_findpoints(fp(dogp), w, p, h, 1.0, 0.0, 1.5, 0.2, 10.0, 5, 4096, rec.ctypes.data_as(C.c_void_p), C.byref(cnt)) == 0
    rec = rec[:cnt.value]
    rec = rec[np.lexsort((rec["scale"], rec["xpos"], rec["ypos"]))]          # the kernel appends in arbitrary order
    out.update(find_thresh=np.float32(1.5), find_xpos=rec["xpos"].copy(), find_ypos=rec["ypos"].copy(), find_scale=rec["scale"].copy(),
               find_sharpness=rec["sharpness"].copy(), find_edgeness=rec["edgeness"].copy())

    n, L = 256, 64
    s = synth.homography_scene(n, seed=5)["sift"]
    coord = np.ascontiguousarray(np.stack([s["xpos"], s["ypos"], s["match_xpos"], s["match_ypos"]]).astype(np.float32))
    rng = np.random.default_rng(9)
    pts = np.ascontiguousarray(np.array([rng.choice(n, 4, replace=False) for _ in range(L)], np.int32).T)
    homo = np.empty((8, L), np.float32); cnt = np.empty(L, np.int32)
    assert R.refk_homography(fp(coord), n, pts.ctypes.data_as(i32p), L, 4.0, fp(homo), cnt.ctypes.data_as(i32p)) == 0
    out.update(homo_coord=coord, homo_pts=pts, homo_thresh=np.float32(4.0), homo_h=homo, homo_counts=cnt)

    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_gpu_kernels.npz"))
