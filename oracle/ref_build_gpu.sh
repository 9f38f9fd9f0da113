#!/bin/sh
# Device build of the REFERENCE'S OWN KERNELS for gfx950 (test infrastructure only).
#
# hipcc compiles CUDA-dialect kernel code natively (__global__, threadIdx, __shared__, float4 ...),
# so the kernel bodies are taken from the reference sources where they lie -- line ranges located by
# grep at build time, assembled in a mktemp directory, never copied into the repository -- and only the
# resulting shared object lands in oracle/_ref/.  No stand-in headers: the ranges are chosen to exclude
# the #include lines and the cuBLAS / cuSOLVER host wrappers that cannot be built here.
#
#   libref_kernels.so   SfM/kernels.h  __global__ kernels (kernels, copy_point, normalizeE, element_wise_*,
#                       vecnorm, threshold_count, candidate_kernels, compute_linear_triangulation_A,
#                       normalize_pt_kernal, row_extraction_kernel) + SfM/svd.h device functions,
#                       built with -ffp-contract=off (the arithmetic contract of the oracle);
#                       CudaSift/matching.cu CleanMatches + FindMaxCorr10 (the live matcher, :289-397),
#                       built with -ffp-contract=fast (nvcc's default fmad, which fuses :347-350);
#                       CudaSift/matching.cu InvertMatrix<8> + ComputeHomographies + TestHomographies
#                       (:821-996, the FindHomography kernels), built with -ffp-contract=off;
#                       CudaSift/cudaSiftD.cu ScaleDown (:84-169), ScaleUp (:171-194), LaplaceMultiMem
#                       (:1753-1790), LowPassBlock (:1986-2038), FindPointsMulti (:1433-1574, the detector the
#                       reference launches when built with MANAGEDMEM, cudaSiftH.cu:508-510: the same tests
#                       and sub-pixel refinement as the default FindPointsMultiNew, candidates compacted with
#                       a shared-memory atomic instead of warp votes) + the shuffle templates of cudautils.h,
#                       built with -ffp-contract=off (with "fast" the compiler fuses the same source
#                       expression differently at different unroll sites) and correctly rounded division
#                       (HIP's __fdividef is a plain '/').  Not buildable for gfx950: FindPointsMultiNew
#                       (__any_sync with a 32-bit mask), ComputeOrientationsCONST /
#                       ExtractSiftDescriptorsCONSTNew (tex2D).
# Launch geometry in ref_driver_gpu.inc follows the reference's call sites (cited there).
set -eu
REF="${1:-/root/reference}"
OUT="${2:-$(cd "$(dirname "$0")" && pwd)/_ref}"
HERE="$(cd "$(dirname "$0")" && pwd)"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
K="$REF/SfM/kernels.h"; SVD="$REF/SfM/svd.h"; COMMON="$REF/SfM/common.h"; M="$REF/CudaSift/matching.cu"

kb=$(grep -n 'void kernels(float\*d1' "$K" | cut -d: -f1)            # first __global__ after the host wrappers
kb=$((kb - 1))                                                        # its __global__ line
ke=$(grep -n 'T\* cuda_alloc_copy' "$K" | cut -d: -f1)                # template that follows the last kernel
ke=$((ke - 2))
{
  echo '#include <hip/hip_runtime.h>'
  echo '#include <math.h>'
  echo '#include <assert.h>'
  echo '#include <string.h>'
  echo "#include \"$REF/CudaSift/cudaSift.h\""                         # SiftPoint: the reference's own header
  grep -E '^#define access[23]\(' "$COMMON"
  grep -E '^#define (x_pos|y_pos|z_pos) ' "$K"
  grep -v '^#include' "$SVD"
  echo 'namespace refk {'
  sed -n "${kb},${ke}p" "$K"
  echo '}'
  cat "$HERE/ref_driver_gpu.inc"
} > "$TMP/ref_kernels.hip"

mb=$(grep -n '^__global__ void CleanMatches' "$M" | cut -d: -f1)
me=$(grep -n '^#define FMC_GH' "$M" | cut -d: -f1)
{
  echo '#include <hip/hip_runtime.h>'
  echo "#include \"$REF/CudaSift/cudaSift.h\""
  sed -n "${mb},$((me - 1))p" "$M"
  cat "$HERE/ref_driver_match_gpu.inc"
} > "$TMP/ref_matchk.hip"

hb=$(grep -n '^template <int size>' "$M" | head -1 | cut -d: -f1)          # InvertMatrix<size>
he=$(grep -n '^//================= Host matching functions' "$M" | cut -d: -f1)
{
  echo '#include <hip/hip_runtime.h>'
  echo '#include <math.h>'
  sed -n "${hb},$((he - 1))p" "$M"                                        # InvertMatrix, ComputeHomographies, TestHomographies
  cat "$HERE/ref_driver_homo_gpu.inc"
} > "$TMP/ref_homo.hip"

D="$REF/CudaSift/cudaSiftD.cu"; U="$REF/CudaSift/cudautils.h"
cb=$(grep -n '^__constant__ int d_MaxNumPoints' "$D" | cut -d: -f1)
ce=$(grep -n '^__constant__ float d_LaplaceKernel' "$D" | cut -d: -f1)
sdb=$(grep -n '^__global__ void ScaleDown(float' "$D" | cut -d: -f1)
sub=$(grep -n '^__global__ void ScaleUp(float' "$D" | cut -d: -f1)
sue=$(grep -n '^__global__ void ExtractSiftDescriptors(cudaTextureObject_t' "$D" | cut -d: -f1)
lmb=$(grep -n '^__global__ void LaplaceMultiMem(float' "$D" | cut -d: -f1)
lme=$(grep -n '^__global__ void LaplaceMultiMemWide(float' "$D" | cut -d: -f1)
lpb=$(grep -n '^__global__ void LowPassBlock(float' "$D" | cut -d: -f1)
fpb=$(grep -n '^__global__ void FindPointsMulti(float' "$D" | cut -d: -f1)
fpe=$(grep -n '^__global__ void FindPointsMultiOld(float' "$D" | cut -d: -f1)
shd=$(grep -n 'T ShiftDown(T var' "$U" | cut -d: -f1)
she=$(grep -n '^#endif' "$U" | tail -1 | cut -d: -f1)
{
  echo '#include <hip/hip_runtime.h>'
  echo "#include \"$REF/CudaSift/cudaSiftD.h\""                           # tile constants: the reference's own header
  echo "#include \"$REF/CudaSift/cudaSift.h\""                            # SiftPoint
  sed -n "$((shd - 1)),$((she - 1))p" "$U"                                 # templates ShiftDown / ShiftUp / Shuffle (pre-CUDA-9 branch: __shfl_*)
  sed -n "${cb},${ce}p" "$D"                                              # __constant__ tables
  sed -n "${sdb},$((sue - 1))p" "$D"                                      # ScaleDown, ScaleUp
  sed -n "${fpb},$((fpe - 1))p" "$D"                                      # FindPointsMulti (the MANAGEDMEM-path detector)
  sed -n "${lmb},$((lme - 1))p" "$D"                                      # LaplaceMultiMem
  sed -n "${lpb},\$p" "$D"                                                # LowPassBlock (to end of file)
  cat "$HERE/ref_driver_sift_gpu.inc"
} > "$TMP/ref_sift.hip"

HIPCC=/opt/rocm/bin/hipcc
$HIPCC --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -w -c "$TMP/ref_kernels.hip" -o "$TMP/a.o"
$HIPCC --offload-arch=gfx950 -O2 -ffp-contract=fast -fPIC -w -c "$TMP/ref_matchk.hip" -o "$TMP/b.o"
$HIPCC --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DOCML_BASIC_ROUNDED_OPERATIONS -fPIC -w -c "$TMP/ref_homo.hip" -o "$TMP/c.o"
$HIPCC --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -w -c "$TMP/ref_sift.hip" -o "$TMP/d.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libref_kernels.so" "$TMP/a.o" "$TMP/b.o" "$TMP/c.o" "$TMP/d.o"
echo "ref_build_gpu: wrote $OUT/libref_kernels.so"
