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
