/*
 * sfm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the two-view geometric-estimation hot path of
 * Black-Phoenix/CUDA-SfM.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the shipped HIP path
 * (cuda-sfm_amd/csrc) never includes, links or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_svd3 / orc_mult* / orc_det_ref / orc_transpose3 : bit-exact against the
 *     reference's own SfM/svd.h compiled in place (oracle/_ref, tests/golden/svd3_*.npz).
 *   - orc_match                                         : index-exact against the reference's
 *     CPU matcher MatchC1 (CudaSift/match.cu:57-71) compiled in place.
 *   - 8x9 / 4x4 null vectors, 4x4 inverse, K^-1 GEMM     : the reference delegates these to
 *     closed-source cuSOLVER/cuBLAS (kernels.h:102-234) -> PARITY UNPINNED at that boundary;
 *     checked through mathematical invariants and an fp64 LAPACK cross-check instead.
 *
 * Arithmetic contract: IEEE-754 binary32, round-to-nearest-even, no contraction
 * (-ffp-contract=off), fused multiply-add ONLY where fmaf() is written, correctly rounded
 * '/' and sqrtf, subnormals kept.  The HIP kernels follow the same contract so that
 * integer outputs (inlier counts, masks, match indices, hypothesis ids) are bit-exact.
 *
 * All matrices are row-major (reference: SfM/common.h:19-20 access2/access3).
 */
#ifndef SFM_ORACLE_H
#define SFM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Data contract of the feature records (reference: CudaSift/cudaSift.h:6-22), 576 bytes. */
typedef struct {
    float xpos, ypos, scale, sharpness, edgeness, orientation;
    float score, ambiguity;
    int   match;
    float match_xpos, match_ypos, match_error, subsampling;
    float empty[3];
    float data[128];
} orc_sift_point;

/* ---- 3x3 algebra (reference SfM/svd.h) ------------------------------------------- */
void  orc_multAB (const float *a, const float *b, float *m);   /* svd.h:58-65  */
void  orc_multAtB(const float *a, const float *b, float *m);   /* svd.h:67-74  */
void  orc_multABt(const float *a, const float *b, float *m);   /* svd.h:76-83  */
void  orc_neg3   (float *a);                                   /* svd.h:85-96  */
float orc_det_ref(const float *a);   /* svd.h:337-341 AS WRITTEN (third term uses a[0]; quirk Q7) */
float orc_det3   (const float *a);   /* mathematically correct determinant                       */
void  orc_transpose_copy3(const float *a, float *b, int a_ld, int b_ld); /* svd.h:343-349 */
void  orc_svd3(const float *a, float *u, float *s, float *v);  /* svd.h:311-335 */

/* ---- fillXU (sfm.cu:80-92, kernels.h:261-279 copy_point, kernels.h:102-109 GEMM) --- */
void orc_fill_xu(const orc_sift_point *pts, int n, const float kinv[9],
                 float *U0, float *U1, float *X0, float *X1);   /* each 3 x n row-major */

/* ---- RANSAC 8-point (sfm.cu:94-153, kernels.h:236-259, 281-295, 343-355) ----------- */
uint32_t orc_hash32(uint32_t x);
void orc_sample8(uint32_t seed, uint32_t hyp, int n, int idx[8]);
void orc_build_A(const float *X0, const float *X1, int n, const int idx[8], float A[72]);
void orc_AtA9(const float A[72], float S[81]);
void orc_jacobi9(float S[81], float V[81], int sweeps);
void orc_nullvec9(const float A[72], int sweeps, float e[9]);   /* sweeps > 0: normal equations + Jacobi; 0: Householder */
void orc_nullvec9_qr(const float A[72], float e[9]);
void orc_normalizeE(float E[9]);
float orc_residual(const float E[9], float x1x, float x1y, float x1z,
                   float x2x, float x2y, float x2z);
int  orc_count_inliers(const float E[9], const float *X0, const float *X1, int n,
                       float thr, uint8_t *mask /* may be NULL */);
void orc_hypothesis_E(const float *X0, const float *X1, int n, const int idx[8],
                      int sweeps, float E[9]);
/* Scores hypotheses [h0, h0+count).  indices: explicit int32[8*H_total] (global ids) or NULL
 * -> keyed sampler orc_sample8(seed, h, n).  counts / Ecand may be NULL.  Returns packed key
 * of the best hypothesis in the range: (count << 32) | (0xFFFFFFFF - hyp).  OpenMP over hyps. */
uint64_t orc_ransac_range(const float *X0, const float *X1, int n,
                          uint32_t h0, uint32_t count, const int *indices, uint32_t seed,
                          float thr, int sweeps, int *counts, float *Ecand, int nthreads);
/* sfm_oracle_fast.c: the same results through the vectorised division-free filter + exact fallback (cpu_baseline) */
int  orc_count_inliers_fast(const float E[9], const float *X0, const float *X1, int n, float thr);
uint64_t orc_ransac_range_fast(const float *X0, const float *X1, int n,
                               uint32_t h0, uint32_t count, const int *indices, uint32_t seed,
                               float thr, int sweeps, int *counts, float *Ecand, int nthreads);
uint64_t orc_pack_key(uint32_t count, uint32_t hyp);
void     orc_unpack_key(uint64_t key, uint32_t *count, uint32_t *hyp);

/* ---- pose candidates / choosePose / triangulation (sfm.cu:238-344, kernels.h:357-450) */
#define ORC_POSE_REFERENCE 0   /* as written: t = -/+ U[:,2], buggy det, inverse in place (Q7,Q8,Q11) */
#define ORC_POSE_CORRECT   1   /* t = -/+ V[:,2], det3, non-inverted P, cheirality on (P X)_z         */
void orc_pose_candidates(const float E[9], int mode, float P[64]);
void orc_tri_A(float x1, float y1, float x2, float y2, const float m1[16], const float m2[16], float A[16]);
void orc_nullvec4(const float A[16], int sweeps, float v[4]);
void orc_normalize_pt(const float v[4], float out[4]);         /* kernels.h:433-450 */
int  orc_inv4(const float m[16], float out[16]);
/* P: 4 candidates in; Pinv out (4x16).  Returns P_ind. d1/d2 (16 floats each, 4 x 4 row-major
 * [component][candidate]) optional outputs. */
int  orc_choose_pose(const float *X0, const float *X1, int n, const float P[64], int mode,
                     int sweeps, float Pinv[64], float *d1_out, float *d2_out);
void orc_triangulate(const float *X0, const float *X1, int n, const float Pm[16], int sweeps,
                     float *out4xn);

/* ---- descriptor match (matching.cu:301-397, 1090-1206; oracle semantics = match.cu:57-71) */
void orc_match_desc(const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                    float *best, float *second, int *index, int nthreads);
void orc_match_sift(orc_sift_point *s1, int n1, const orc_sift_point *s2, int n2, int nthreads);
/* FindMaxCorr10's own (approximate) second-best score and what its merge picks as best / index (matching.cu:361-390) */
void orc_match_second_ref(const float *d1, int n1, int ld1, const float *d2, int n2, int ld2, float *best_out, float *second_ref, int *index_out, int nthreads);

/* ---- homography RANSAC pre-filter (FindHomography, matching.cu:1000-1087) ---- */
/* coord: 4 x ld row-major (x1; y1; x2; y2), pts[4]: sample.  h[8] (h33 = 1 implied).
 * ComputeHomographies matching.cu:907-948 + InvertMatrix<8> :821-905. */
void orc_homography4(const float *coord, int ld, const int pts[4], float h[8]);
/* TestHomographies matching.cu:953-996 (round-toward-zero products), points 0..n-1. */
int  orc_homography_count(const float h[8], const float *coord, int ld, int n, float thresh2);

void orc_homography_sample(uint32_t seed, uint32_t loop, uint32_t nvalid, uint32_t pick[4]);
int  orc_find_homography(const orc_sift_point *s, int n, int num_loops, float min_score, float max_ambiguity,
                         float thresh, uint32_t seed, float H[9], int *counts, float *homo);

/* ---- ExtractSift (CudaSift/cudaSiftH.cu:72-232 + live kernels of cudaSiftD.cu).
 * Implemented in sift_oracle.c; see its header for pinning status and the documented differences. ---- */
float orc_sift_exp2f(float t);
float orc_sift_expf(float x);
float orc_sift_atan2f(float y, float x);
float orc_sift_fast_atan2f(float y, float x);                       /* cudaSiftD.cu:296-306 */
void  orc_sift_sincosf(float th, float *sn, float *cs);
float orc_sift_tex(const float *img, int pitch, int w, int h, float x, float y);
void  orc_sift_lowpass_kernel(float scale, float k[9]);             /* cudaSiftH.cu:422-431 */
void  orc_sift_scaledown_kernel(float variance, float k[5]);        /* cudaSiftH.cu:316-323 */
void  orc_sift_laplace_kernels(int numOctaves, float initBlur, float *kernel /* 8*12*16 */);
void  orc_sift_lowpass(const float *src, int w, int h, int ps, float *dst, int pd, const float k[9]);
void  orc_sift_scaledown(const float *src, int w, int h, int ps, float *dst, int pd, const float k[5]);
void  orc_sift_scaleup(const float *src, int w, int h, int ps, float *dst, int pd);
void  orc_sift_laplace(const float *img, int w, int h, int pi, float *dog, int pd, const float *kern);
void  orc_sift_find_points(const float *dog, int w, int h, int pd, float subsampling, float lowestScale,
                           float thresh, float factor, float edgeLimit, orc_sift_point *pts, int *count, int maxPts);
int   orc_sift_orientation(const float *img, int w, int h, int pitch, float xpos, float ypos, float scale, float ori[2]);
void  orc_sift_descriptor(const float *img, int w, int h, int pitch, float xpos, float ypos, float scale,
                          float orientation, float desc[128]);
int   orc_extract_sift(const float *image, int width, int height, int pitch, int numOctaves, double initBlur,
                       float thresh, float lowestScale, int scaleUp, orc_sift_point *pts, int maxPts, int *total_stored);

int orc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
