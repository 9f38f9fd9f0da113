"""CPU: host-side IO helpers of the C++ facade (.sift round trip, PLY sink) -- plain C++, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sift_file_and_ply(tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "io_test")
    assert os.path.exists(exe), "tests/cpp/io_test not built (make)"
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    ply = open(tmp_path / "inl.ply").read().splitlines()
    assert ply[0] == "ply" and "element vertex 3" in ply and ply[-1].split() == ["5", "5", "10"]
    assert os.path.getsize(tmp_path / "a.sift") == 4 + 37 * 576


def test_pnm_reader(tmp_path):
    """ReadPNM: P5, P6 (OpenCV's fixed-point grey conversion), comments in the header, bad files rejected."""
    import numpy as np
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (7, 11), dtype=np.uint8)
    c = rng.integers(0, 256, (5, 9, 3), dtype=np.uint8)
    (tmp_path / "in_gray.pnm").write_bytes(b"P5\n11 7\n255\n" + g.tobytes())
    (tmp_path / "in_color.pnm").write_bytes(b"P6 9 5 255\n" + c.tobytes())
    (tmp_path / "in_comment.pnm").write_bytes(b"P5\n# made by a test\n11 # width\n7\n255\n" + g.tobytes())
    (tmp_path / "bad.pnm").write_bytes(b"P5\n11 7\n255\n" + g.tobytes()[:20])
    exe = os.path.join(ROOT, "tests", "cpp", "io_test")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)

    def load(name):
        raw = (tmp_path / (name + ".f32")).read_bytes()
        w, h = np.frombuffer(raw[:8], np.int32)
        return np.frombuffer(raw[8:], np.float32).reshape(h, w)
    assert np.array_equal(load("in_gray.pnm"), g.astype(np.float32))
    assert np.array_equal(load("in_comment.pnm"), g.astype(np.float32))
    ci = c.astype(np.int64)
    grey = (ci[..., 0] * 4899 + ci[..., 1] * 9617 + ci[..., 2] * 1868 + 8192) >> 14       # cv::cvtColor RGB2GRAY, 8-bit
    assert np.array_equal(load("in_color.pnm"), grey.astype(np.float32))
