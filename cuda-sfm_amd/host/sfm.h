// sfm.h -- host-side mirror of the reference's SfM::Image_pair (SfM/sfm.h:20-60, SfM/sfm.cu:28-359):
// same class name, constructor signature, method names and call order as the reference's caller uses
// them (src/main.cpp:298-307), implemented on the C ABI of include/sfm_amd.h.
//
//     SfM::Image_pair sfm(K, inv_K, 2, siftData1.numPts);
//     sfm.fillXU(siftData1.d_data);
//     sfm.estimateE();
//     sfm.computePosecandidates();
//     sfm.choosePose();
//     sfm.linear_triangulation();
//
// Additions (the reference keeps its results private and only leaks them through a GL VBO copy,
// sfm.cu:374-383): get* accessors, setRansacParams, setPoseMode and both spellings of the two
// BASELINE names (computePoseCandidates / linearTriangulate).
#ifndef SFM_AMD_SFM_H
#define SFM_AMD_SFM_H

#include <cstdint>
#include <vector>

#include "cudaSift.h"

namespace SfM {

class Image_pair {
    sfm_pair *pair_ = nullptr;
    int num_points_ = 0;
    int pose_mode_ = SFM_POSE_REFERENCE;       // drop-in default: the reference's own behaviour
    sfm_ransac_params params_;
public:
    Image_pair(float k[9], float k_inv[9], int image_count, int num_points) : num_points_(num_points)
    {
        SFM_FACADE_CALL(sfm_pair_create(sfm_facade::context(), k, k_inv, image_count, num_points, &pair_));
        sfm_ransac_default_params(&params_, num_points);      // H = N/8, thr 1e-6 (sfm.cu:95,220)
    }
    Image_pair(const Image_pair &) = delete;
    Image_pair &operator=(const Image_pair &) = delete;
    ~Image_pair() { if (pair_) sfm_pair_destroy(pair_); }

    // ---- the reference's call surface --------------------------------------------------------
    void fillXU(SiftPoint *data) { SFM_FACADE_CALL(sfm_fill_xu(pair_, data)); }               // sfm.cu:80-92
    void estimateE() { SFM_FACADE_CALL(sfm_estimate_E(pair_, &params_)); }                    // sfm.cu:94-153
    void computePosecandidates() { SFM_FACADE_CALL(sfm_pose_candidates(pair_, pose_mode_)); } // sfm.cu:238-252
    void choosePose() { SFM_FACADE_CALL(sfm_choose_pose(pair_, pose_mode_)); }                // sfm.cu:254-307
    void linear_triangulation() { SFM_FACADE_CALL(sfm_triangulate(pair_, pose_mode_)); }      // sfm.cu:309-344
#ifdef SFM_AMD_COMM_H
    // multi-GPU estimateE (include sfm_amd_comm.h first): num_hypotheses is the global count, every rank calls it
    void estimateE(sfm_comm *comm) { SFM_FACADE_CALL(sfm_estimate_E_sharded(pair_, &params_, comm)); }
#endif
    void computePoseCandidates() { computePosecandidates(); }
    void linearTriangulate() { linear_triangulation(); }

    // ---- additions ----------------------------------------------------------------------------
    sfm_ransac_params &ransacParams() { return params_; }
    void setRansacParams(uint32_t num_hypotheses, float threshold, uint32_t seed, const int32_t *d_indices = nullptr)
    {
        params_.num_hypotheses = num_hypotheses; params_.threshold = threshold; params_.seed = seed; params_.d_indices = d_indices;
        params_.hyp_begin = 0; params_.hyp_count = 0;
    }
    void setPoseMode(int mode) { pose_mode_ = mode; }
    // computePosecandidates + choosePose + linear_triangulation (src/main.cpp:302-306) as one launch; same results
    void poseChain() { SFM_FACADE_CALL(sfm_pose_chain(pair_, pose_mode_)); }
    // another correspondence set of at most the constructor's num_points, without re-allocating anything
    void reset(int num_points)
    {
        SFM_FACADE_CALL(sfm_pair_reset(pair_, num_points));
        num_points_ = num_points;
        sfm_ransac_default_params(&params_, num_points);
    }
    void getResult(float record[28]) { SFM_FACADE_CALL(sfm_get_result(pair_, record)); }
    int numPoints() const { return num_points_; }
    sfm_pair *handle() { return pair_; }

    void getE(float E[9]) { SFM_FACADE_CALL(sfm_get_E(pair_, E)); }
    void getBestHypothesis(uint32_t *hyp, uint32_t *count) { SFM_FACADE_CALL(sfm_get_best(pair_, hyp, count)); }
    std::vector<int32_t> getInlierCounts()
    {
        std::vector<int32_t> c(params_.hyp_count ? params_.hyp_count : params_.num_hypotheses - params_.hyp_begin);
        SFM_FACADE_CALL(sfm_get_inlier_counts(pair_, c.data(), c.size()));
        return c;
    }
    std::vector<uint8_t> getInlierMask()
    {
        std::vector<uint8_t> m((size_t)num_points_);
        SFM_FACADE_CALL(sfm_get_inlier_mask(pair_, m.data()));
        return m;
    }
    void getPoseCandidates(float P[64]) { SFM_FACADE_CALL(sfm_get_pose_candidates(pair_, P)); }
    void getPoseInverses(float P[64]) { SFM_FACADE_CALL(sfm_get_pose_inverses(pair_, P)); }
    int getPoseIndex() { int i = 0; SFM_FACADE_CALL(sfm_get_pose_index(pair_, &i)); return i; }
    // sfm.cu:374-383: (x, y, z, 1) vertices and the constant colour buffer, into DEVICE buffers of 4 * N floats
    void copyBoidsToVBO(float *vbodptr_positions, float *vbodptr_velocities)
    {
        SFM_FACADE_CALL(sfm_copy_points_to_vbo(pair_, vbodptr_positions, vbodptr_velocities, 1.0f));
        SFM_FACADE_CALL(sfm_ctx_synchronize(sfm_facade::context()));            // cudaDeviceSynchronize of sfm.cu:382
    }
    std::vector<float> getPoints()          // 4 x N row-major [x; y; z; 1] (d_final_points, sfm.cu:77,335)
    {
        std::vector<float> p((size_t)4 * num_points_);
        SFM_FACADE_CALL(sfm_get_points(pair_, p.data()));
        return p;
    }
    std::vector<float> getX(int image)      // 3 x N normalised coordinates of image 0 / 1
    {
        std::vector<float> x((size_t)3 * num_points_);
        SFM_FACADE_CALL(sfm_get_XU(pair_, image == 0 ? SFM_BUF_X0 : SFM_BUF_X1, x.data()));
        return x;
    }
};

} // namespace SfM

#endif
