#!/bin/sh
# Compiles the parts of the reference that build with the toolchains in this image, FROM THE
# SOURCES WHERE THEY LIE under $1 (default /root/reference).  Nothing from the reference is
# copied into the repository: translation units are assembled in a mktemp directory, only the
# resulting shared objects land in oracle/_ref/ (git-ignored, shipped to the GPU box with the
# snapshot).  Test infrastructure only -- the product never loads these.
#
#   libref_svd.so      SfM/svd.h inline 3x3 algebra (svd, multAB/AtB/ABt, det, neg,
#                      transpose_copy3x3), host build.  The header's own #include lines
#                      (<cuda.h>, common.h -> cuda_runtime.h) cannot be satisfied here, so the
#                      body below them is compiled as HIP host code: __host__/__device__/
#                      __forceinline__ are the HIP toolchain's own, no stand-in header is written.
#   libref_match.so    CudaSift/match.cu:57-71 MatchC1 (CPU matcher), plain g++.
#   libref_match_fma.so  same, built with FMA contraction (what nvcc does to matching.cu:338-351).
#   libref_kernels.so  (ref_build_gpu.sh) device build of the reference's own kernels for gfx950
#                      (SfM/kernels.h:236-458, CudaSift/matching.cu:289-397), run by tests/test_gpu_ref_kernels.py.
#
# What is NOT buildable here and why: sfm.cu host code, kernels.h wrappers (cuBLAS / cuSOLVER /
# Thrust, CUDA runtime), CudaSift extraction (CUDA textures) -- see DESIGN.md.
set -eu
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"

SVD="$REF/SfM/svd.h"
COMMON="$REF/SfM/common.h"
MATCH="$REF/CudaSift/match.cu"
[ -f "$SVD" ] && [ -f "$COMMON" ] && [ -f "$MATCH" ] || { echo "ref_build: reference sources missing under $REF" >&2; exit 1; }

# ---- libref_svd.so -------------------------------------------------------------------------
{
  echo '#include <hip/hip_runtime.h>'
  echo '#include <math.h>'
  grep -E '^#define access[23]\(' "$COMMON"
  grep -v '^#include' "$SVD"          # the header minus its (unsatisfiable) #include lines
  cat "$HERE/ref_driver_svd.inc"
} > "$TMP/ref_svd.hip"
/opt/rocm/bin/hipcc -x hip --cuda-host-only -O2 -ffp-contract=off -fPIC -shared \
    -Wno-unused-value -o "$OUT/libref_svd.so" "$TMP/ref_svd.hip"

# ---- libref_match*.so ----------------------------------------------------------------------
b=$(grep -n '^void MatchC1' "$MATCH" | cut -d: -f1)
e=$(grep -n '^void MatchC2' "$MATCH" | cut -d: -f1)
{
  echo '#include <cstring>'
  echo 'static int ref_npts = 0;'
  echo '#define NPTS ref_npts'
  echo '#define NDIM 128'
  sed -n "${b},$((e - 1))p" "$MATCH"
  cat "$HERE/ref_driver_match.inc"
} > "$TMP/ref_match.cpp"
g++ -O2 -ffp-contract=off -fPIC -shared -o "$OUT/libref_match.so" "$TMP/ref_match.cpp"
g++ -O2 -mfma -ffp-contract=fast -fPIC -shared -o "$OUT/libref_match_fma.so" "$TMP/ref_match.cpp"

if [ -f "$HERE/ref_build_gpu.sh" ]; then sh "$HERE/ref_build_gpu.sh" "$REF" "$OUT" || echo "ref_build: GPU-side reference build failed (non-fatal)" >&2; fi
echo "ref_build: wrote $(ls "$OUT" | tr '\n' ' ')"
