"""CPU: host/geomFuncs.h ImproveHomography (reference CudaSift/geomFuncs.cpp:6-72) against the numpy
restatement in oracle/ -- plain C++, no GPU (the reference runs this step on the host as well)."""
import os
import subprocess

import numpy as np

import oracle as O
from cuda_sfm_amd_synth import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_sift(path, s):
    with open(path, "wb") as f:
        f.write(np.int32(len(s)).tobytes()); f.write(s.tobytes())


def read_sift(path, dtype):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    return np.frombuffer(raw[4:], dtype, n)


def run(tmp_path, s, H, loops, min_score, max_amb, thresh):
    exe = os.path.join(ROOT, "tests", "cpp", "geom_test")
    assert os.path.exists(exe), "tests/cpp/geom_test not built (make)"
    a, b = str(tmp_path / "in.sift"), str(tmp_path / "out.sift")
    write_sift(a, s)
    args = [exe, a, b] + [float(v).hex() for v in np.asarray(H, np.float32).reshape(9)] + [str(loops), repr(min_score), repr(max_amb), repr(thresh)]
    r = subprocess.run(args, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stderr)
    tok = r.stdout.split()
    return int(tok[0]), np.array([float.fromhex(t) for t in tok[1:]], np.float32).reshape(3, 3), read_sift(b, s.dtype)


def test_improve_homography_matches_oracle(tmp_path):
    sc = synth.homography_scene(2000, seed=4)
    s = sc["sift"]
    H0 = sc["H"].copy(); H0[0, 2] += 1.5; H0[1, 1] *= 1.001            # a perturbed start, as RANSAC would give
    for loops, thresh in ((5, 5.0), (1, 3.0), (0, 5.0)):
        nfit, H, out = run(tmp_path, s, H0, loops, 0.85, 0.95, thresh)
        onfit, oH, oerr = O.improve_homography(s, H0, loops, 0.85, 0.95, thresh)
        assert np.allclose(H, oH, rtol=1e-6, atol=1e-9)
        assert abs(nfit - onfit) <= 1                                  # a borderline point may flip with the solver's last bits
        assert np.allclose(out["match_error"], oerr, rtol=1e-3, atol=1e-3)
        keep = [k for k in s.dtype.names if k != "match_error"]
        assert all(np.array_equal(out[k], s[k]) for k in keep)         # nothing else is touched
    # refinement pulls the perturbed start back onto the plane
    nfit5, H5, _ = run(tmp_path, s, H0, 5, 0.85, 0.95, 5.0)
    nfit0, _, _ = run(tmp_path, s, H0, 0, 0.85, 0.95, 5.0)
    assert nfit5 >= nfit0 and nfit5 > 0.95 * (~sc["outlier"]).sum()
    assert np.abs(H5 - sc["H"]).max() < np.abs(H0 - sc["H"]).max()


def test_improve_homography_degenerate(tmp_path):
    sc = synth.homography_scene(16, seed=6)
    s = sc["sift"].copy(); s["score"] = 0.0                            # gate rejects everything -> M = 0, not SPD
    nfit, H, _ = run(tmp_path, s, sc["H"], 3, 0.85, 0.95, 5.0)
    oH = O.improve_homography(s, sc["H"], 3, 0.85, 0.95, 5.0)[1]
    assert np.array_equal(H, oH) and H[2, 2] == 1.0 and not H.reshape(9)[:8].any()   # cv::solve zeroes the solution
    nfit, H, _ = run(tmp_path, s, sc["H"], 0, 0.85, 0.95, 5.0)          # no refinement loop: H / h33 passes through
    assert np.allclose(H, sc["H"], rtol=1e-6)
