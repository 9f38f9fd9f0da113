"""CPU: the C-ABI library loads and exports every symbol include/sfm_amd.h declares; the package
refuses to work without it; calls fail loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sfm_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_[a-z0-9_A-Z]+)\s*\(", txt)))


def test_header_symbols_exported():
    import cuda_sfm_amd as S
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(S.lib(), n)]
    assert not missing, missing
    assert sorted(S.EXPORTS) == names
    assert S.lib().sfm_abi_version() == 3
    assert not S.AB and S.LIB_PATH.endswith("libsfm_amd.so")


def _nm_exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_dynamic_symbol_tables_are_exactly_the_headers():
    """`nm -D` of the product library lists what include/sfm_amd.h declares and NOTHING else (no kernel stubs, no launchers,
    no probe hooks: csrc/exports.map + -fvisibility=hidden); the lab-bench flavour adds exactly include/sfm_amd_ab.h; the
    RCCL library exports exactly include/sfm_amd_comm.h."""
    import cuda_sfm_amd as S
    assert _nm_exports(S.LIB_PATH) == declared_symbols()
    ab_txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sfm_amd_ab.h")).read(), flags=re.S)
    ab_names = sorted(set(re.findall(r"\b(sfm_[a-z0-9_A-Z]+)\s*\(", ab_txt)))
    assert ab_names == sorted(S.AB_EXPORTS) and len(ab_names) == 4
    ab_lib = os.path.join(os.path.dirname(S.LIB_PATH), "libsfm_amd_ab.so")
    assert _nm_exports(ab_lib) == sorted(declared_symbols() + ab_names)
    assert _nm_exports(S.COMM_LIB_PATH) == sorted(S.COMM_EXPORTS)
    for n in ab_names:
        assert not hasattr(S.lib(), n), n


def test_integration_md_documents_every_export():
    """Every entry point of the product library is named in INTEGRATION.md (section 2: the binding a maintainer writes)."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in declared_symbols() if n not in txt]
    assert not missing, missing
    for n in ("sfm_prefilter_probe", "sfm_ransac_last_trace", "sfm_ransac_last_phases"):
        assert n not in txt or "sfm_amd_ab.h" in txt, n


def test_lab_bench_flavour_imports_next_to_the_product():
    import cuda_sfm_amd as S
    import cuda_sfm_amd_ab as A
    assert A.AB and A.LIB_PATH.endswith("libsfm_amd_ab.so") and A.KERNEL_MFMA == 3 and not hasattr(S, "KERNEL_MFMA")
    assert not [n for n in A.EXPORTS if not hasattr(A.lib(), n)]
    assert A.lib().sfm_abi_version() == 3
    with pytest.raises(ImportError):
        A.comm_lib()                                  # the communicator library is linked against the product


def test_test_only_comm_build_has_the_products_abi_and_no_rccl():
    """tests/fake_ccl/libsfm_amd_fakeccl.so (comm.cpp linked against the shared-memory stand-in, tests/test_gpu_fakeccl.py) exports
    exactly what libsfm_amd_rccl.so exports, depends on the core library and NOT on librccl -- and the product's comm library does
    depend on librccl (the stand-in never ships)."""
    import subprocess
    import cuda_sfm_amd as S
    fake = os.path.join(ROOT, "tests", "fake_ccl", "libsfm_amd_fakeccl.so")
    assert os.path.exists(fake), "run `make`"
    assert _nm_exports(fake) == sorted(S.COMM_EXPORTS)
    need_fake = subprocess.run(["readelf", "-d", fake], capture_output=True, text=True).stdout
    assert "libsfm_amd.so" in need_fake and "rccl" not in need_fake and "nccl" not in need_fake
    default_comm = os.path.join(os.path.dirname(S.LIB_PATH), "libsfm_amd_rccl.so")
    need_real = subprocess.run(["readelf", "-d", default_comm], capture_output=True, text=True).stdout
    assert "librccl" in need_real
    assert "fakeccl" not in S.COMM_LIB_PATH or os.environ.get("SFM_AMD_COMM_LIB")


def test_comm_header_symbols_exported():
    """include/sfm_amd_comm.h <-> libsfm_amd_rccl.so (the RCCL exchange step lives in its own library)."""
    import cuda_sfm_amd as S
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sfm_amd_comm.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(sfm_[a-z0-9_A-Z]+)\s*\(", txt)))
    assert names == sorted(S.COMM_EXPORTS)
    L = S.comm_lib()
    assert not [n for n in names if not hasattr(L, n)]
    # the core library stays free of RCCL
    import subprocess
    needed = subprocess.run(["readelf", "-d", S.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed and "nccl" not in needed


def test_struct_layouts_match_header():
    import cuda_sfm_amd as S
    assert C.sizeof(S.RansacParams) == 56
    assert S.SIFT_DTYPE.itemsize == 576 and S.SIFT_DTYPE.fields["data"][1] == 64
    assert S.SIFT_DTYPE.fields["score"][1] == 24 and S.SIFT_DTYPE.fields["match"][1] == 32   # cudaSift.h:6-22 offsets
    p = S.default_params(4096)
    assert p.num_hypotheses == 512 and abs(p.threshold - 1e-6) < 1e-12 and p.jacobi_sweeps == 0


def test_ctypes_mirrors_equal_sizeof_and_offsetof_of_the_header(tmp_path):
    """A C program prints sizeof / offsetof of every struct that crosses the boundary (sfm_ransac_params, sfm_sift_point,
    sfm_pair_desc, sfm_sift_layout); the ctypes / numpy mirrors of the harness must agree field by field."""
    import subprocess
    import cuda_sfm_amd as S
    fields = {
        "sfm_ransac_params": ["num_hypotheses", "hyp_begin", "hyp_count", "seed", "d_indices", "threshold", "jacobi_sweeps", "kernel", "reserved"],
        "sfm_sift_point": ["xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "score", "ambiguity", "match", "match_xpos",
                           "match_ypos", "match_error", "subsampling", "empty", "data"],
        "sfm_pair_desc": ["d_sift1", "n1", "d_sift2", "n2"],
        "sfm_sift_layout": ["num_octaves", "width", "height", "pitch", "image_offset", "dog_offset", "up_offset", "total_floats"],
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sfm_amd.h"', 'int main(void) {']
    for st, fs in fields.items():
        lines.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            lines.append(f'  printf("{st}.{f} %zu %zu\\n", offsetof({st}, {f}), sizeof((({st} *)0)->{f}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        k, *v = line.split()
        got[k] = tuple(int(x) for x in v)
    mirrors = {"sfm_ransac_params": S.RansacParams, "sfm_pair_desc": S.PairDesc, "sfm_sift_layout": S.SiftLayout}
    for st, cls in mirrors.items():
        assert got[st] == (C.sizeof(cls),), (st, got[st], C.sizeof(cls))
        assert [f for f, _ in cls._fields_] == fields[st]
        for f in fields[st]:
            d = getattr(cls, f)
            assert got[f"{st}.{f}"] == (d.offset, d.size), (st, f, got[f"{st}.{f}"], d.offset, d.size)
    assert got["sfm_sift_point"] == (S.SIFT_DTYPE.itemsize,) == (576,)
    assert list(S.SIFT_DTYPE.names) == fields["sfm_sift_point"]
    for f in fields["sfm_sift_point"]:
        dt, off = S.SIFT_DTYPE.fields[f][:2]
        assert got[f"sfm_sift_point.{f}"] == (off, dt.itemsize), (f, got[f"sfm_sift_point.{f}"], off, dt.itemsize)


def test_header_constants_match_the_python_mirror():
    """#define values of include/sfm_amd.h against the names the Python harness uses (matcher / kernel selectors, quirks,
    pose modes, error codes)."""
    import cuda_sfm_amd as S
    txt = open(os.path.join(ROOT, "include", "sfm_amd.h")).read()
    defs = {m.group(1): int(m.group(2).strip("()").rstrip("u"), 0)
            for m in re.finditer(r"^#define\s+(SFM_[A-Z0-9_]+)\s+(\(?-?(?:0x)?[0-9a-fA-F]+u?\)?)\s*(?:/\*.*)?$", txt, flags=re.M)}
    assert len(defs) >= 30
    checked = 0
    for name, value in defs.items():
        py = name[len("SFM_"):]
        if hasattr(S, py):
            assert getattr(S, py) == value, (name, value, getattr(S, py))
            checked += 1
    assert checked >= 12, checked
    assert (S.MATCH_AUTO, S.MATCH_EXACT, S.MATCH_PREFILTER, S.MATCH_FUSED) == (0, 1, 2, 3)


def test_no_cpu_fallback():
    import torch
    import cuda_sfm_amd as S
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(S.SfmError) as e:
        S.Context(0)
    assert e.value.code == S.E_HIP


def test_host_logic_keys_and_shards():
    import cuda_sfm_amd as S
    assert S.unpack_key(S.pack_key(123, 456)) == (123, 456)
    assert S.pack_key(5, 9) > S.pack_key(4, 0) and S.pack_key(5, 3) > S.pack_key(5, 9)      # count first, then lowest id
    for H in (1, 7, 8, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            parts = [S.shard_range(H, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == H
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_header_is_plain_c_and_links(tmp_path):
    """include/sfm_amd.h is the drop-in boundary: it must compile as C99 without any HIP / C++ header and a
    plain C program must link against libsfm_amd.so using nothing else (struct sizes as the reference's FFI expects)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "sfm_amd.h"
int main(void)
{
    sfm_ransac_params p;
    sfm_sift_layout L;
    sfm_ransac_default_params(&p, 4096);
    if (sizeof(sfm_sift_point) != 576 || sizeof(p) != 56) return 2;
    if (sfm_sift_temp_layout(1920, 1080, 5, 0, &L) != SFM_OK || L.width[4] != 120 || L.pitch[0] != 1920) return 3;
    if (sfm_sift_temp_layout(0, 10, 5, 0, &L) != SFM_E_INVALID) return 4;
    printf("%d %u %d %lld\n", sfm_abi_version(), p.num_hypotheses, p.jacobi_sweeps, (long long)L.total_floats);
    return 0;
}
''')
    exe = tmp_path / "abi"
    lib = os.path.join(ROOT, "cuda-sfm_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", lib, "-lsfm_amd", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    ver, hyps, sweeps, floats = r.stdout.split()
    assert (int(ver), int(hyps), int(sweeps)) == (3, 512, 0) and int(floats) > 8 * 1920 * 1080


def test_comm_library_argument_and_lifecycle_paths_without_a_gpu():
    """libsfm_amd_rccl.so as far as it can be driven without a GPU: the out-of-band id (ncclGetUniqueId needs no device),
    argument validation of every entry point (a bad call must come back with SFM_E_INVALID, never dereference), and the
    lifecycle corners of the Python wrapper -- so that the first real multi-rank run does not die on plumbing."""
    import cuda_sfm_amd as S
    L = S.comm_lib()
    a, b = C.create_string_buffer(S.COMM_ID_BYTES), C.create_string_buffer(S.COMM_ID_BYTES)
    assert L.sfm_comm_unique_id(a) == S.OK and L.sfm_comm_unique_id(b) == S.OK
    assert any(a.raw) and a.raw != b.raw                                  # a fresh id per call: rank 0 makes ONE and hands it out
    assert S.Comm.unique_id() != S.Comm.unique_id() and len(S.Comm.unique_id()) == S.COMM_ID_BYTES
    assert L.sfm_comm_unique_id(None) == S.E_INVALID
    out = C.c_void_p()
    fake_ctx = C.c_void_p(0x1000)                                         # never dereferenced: the checks come first
    L.sfm_comm_init.restype = C.c_int
    for ctx, ident, rank, nranks, dst in [(None, a, 0, 1, C.byref(out)), (fake_ctx, None, 0, 1, C.byref(out)), (fake_ctx, a, 0, 1, None),
                                          (fake_ctx, a, 0, 0, C.byref(out)), (fake_ctx, a, -1, 2, C.byref(out)), (fake_ctx, a, 2, 2, C.byref(out)),
                                          (fake_ctx, a, 8, 8, C.byref(out))]:
        assert L.sfm_comm_init(ctx, ident, rank, nranks, dst) == S.E_INVALID
        assert not out.value
    n, r = C.c_int(-7), C.c_int(-7)
    assert L.sfm_comm_nccl_ranks(None, C.byref(n)) == S.E_INVALID and n.value == -7
    assert L.sfm_comm_rank(None, C.byref(r), C.byref(n)) == S.E_INVALID
    assert L.sfm_comm_flush(None) == S.E_INVALID                           # flush-before-read on a communicator that never came up
    p = S.default_params(100)
    assert L.sfm_estimate_E_sharded(None, C.byref(p), None) == S.E_INVALID
    assert L.sfm_estimate_E_sharded_pipelined(None, C.byref(p), None) == S.E_INVALID
    assert L.sfm_comm_destroy(None) == S.OK                                # destroy of nothing is not an error (cleanup paths)
    # wrapper: close() is idempotent and safe on an object whose constructor failed half-way
    c = object.__new__(S.Comm)
    c.close(); c.close()
    c._h = None
    c.close()
    with pytest.raises(AssertionError):
        S.Comm(None, b"short", 0, 1)                                      # an id of the wrong size never reaches the library
