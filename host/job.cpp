// job.cpp -- see job.hpp.
#include "job.hpp"
#include <algorithm>
#include <cstring>
#include <numeric>

namespace mis {

static float median_focal(const std::vector<CameraParams>& cams, const std::vector<int>& idx) {
    // image_stitching.cpp:884-895: median of the kept cameras' focals (mean of the middle two for an even count), as float
    std::vector<double> f;
    for (int i : idx) f.push_back(cams[i].focal);
    std::sort(f.begin(), f.end());
    return f.size() % 2 == 1 ? static_cast<float>(f[f.size() / 2]) : static_cast<float>(f[f.size() / 2 - 1] + f[f.size() / 2]) * 0.5f;
}

void StitchJob::check(MisContext* c, int rc, const char* what) const {
    if (rc != MIS_OK) throw std::runtime_error(std::string(what) + " failed: " + mis_last_error(c));
}

StitchJob::StitchJob(int device, int width, int height, const std::vector<CameraParams>& cameras, const StitchConfig& cfg)
    : w_(width), h_(height), n_((int)cameras.size()), cams_(cameras), cfg_(cfg) {
    if (cfg_.features_type != "orb" || cfg_.ba_cost_func != "no" || cfg_.expos_comp_type != "no" || cfg_.seam_find_type != "no")
        throw std::runtime_error("mis::StitchJob runs the hot path (ORB, supplied cameras, no seam-scale step); use mis::Stitcher for the other options");
    if (mis_context_create(device, nullptr, &ctx_) != MIS_OK) throw std::runtime_error("mis_context_create failed: no HIP device (there is no CPU fallback)");
    check(ctx_, mis_stream_create(device, 0, &cstream_), "mis_stream_create");
    if (mis_context_create(device, cstream_, &cctx_) != MIS_OK) throw std::runtime_error("mis_context_create (compose stream) failed");
    MisOrbParams op;
    mis_orb_default_params(&op);
    check(ctx_, mis_orb_create(ctx_, &op, w_, h_, &orb_), "mis_orb_create");
    Ks_.resize((size_t)n_ * 9); Rs_.resize((size_t)n_ * 9);
    for (int i = 0; i < n_; i++) {
        const Mat3<float> K = cams_[i].K().cast<float>(), R = cams_[i].R.cast<float>();
        std::copy(K.m.begin(), K.m.end(), Ks_.begin() + 9 * i);
        std::copy(R.m.begin(), R.m.end(), Rs_.begin() + 9 * i);
    }
}

StitchJob::~StitchJob() {
    if (cctx_) mis_context_synchronize(cctx_);
    if (ctx_) mis_context_synchronize(ctx_);
    if (!pairwise_.empty()) mis_matches_free(pairwise_.data(), (int)pairwise_.size());
    if (cctx_) { mis_image_free(cctx_, &pano_); mis_image_free(cctx_, &mask_); }
    if (blender_) mis_blender_destroy(blender_);
    if (orb_) mis_orb_destroy(orb_);
    if (cctx_) mis_context_destroy(cctx_);
    if (cstream_) mis_stream_destroy(cstream_);
    if (ctx_) mis_context_destroy(ctx_);
}

void StitchJob::synchronize() {
    check(ctx_, mis_context_synchronize(ctx_), "mis_context_synchronize");
    check(cctx_, mis_context_synchronize(cctx_), "mis_context_synchronize (compose)");
}

// warpRoi of the frames `idx` at the scale of that set, panorama roi, blender sizing + prepare -- on the compose stream
StitchJob::Compose StitchJob::prepare(const std::vector<int>& idx) {
    const int m = (int)idx.size();
    const float scale = median_focal(cams_, idx);
    std::vector<float> Ks((size_t)m * 9), Rs((size_t)m * 9);
    for (int k = 0; k < m; k++) {
        std::copy(Ks_.begin() + 9 * idx[k], Ks_.begin() + 9 * idx[k] + 9, Ks.begin() + 9 * k);
        std::copy(Rs_.begin() + 9 * idx[k], Rs_.begin() + 9 * idx[k] + 9, Rs.begin() + 9 * k);
    }
    rois_.assign(m, MisRect{});
    check(cctx_, mis_warp_roi_batch(cctx_, scale, w_, h_, m, Ks.data(), Rs.data(), rois_.data()), "mis_warp_roi_batch");
    std::vector<MisPoint> corners(m);
    std::vector<MisSize> sizes(m);
    for (int k = 0; k < m; k++) { corners[k] = {rois_[k].x, rois_[k].y}; sizes[k] = {rois_[k].width, rois_[k].height}; }
    Compose c;
    check(cctx_, mis_result_roi(corners.data(), sizes.data(), m, &c.pano), "mis_result_roi");
    check(cctx_, mis_blend_config(cfg_.blend_type, cfg_.blend_strength, c.pano.width, c.pano.height, &c.type, &c.bands, &c.sharp), "mis_blend_config");
    if (!blender_ || c.type != key_.type || c.bands != key_.bands || c.sharp != key_.sharp) {      // band count / sharpness are creation parameters
        if (blender_) { mis_blender_destroy(blender_); blender_ = nullptr; }
        check(cctx_, mis_blender_create(cctx_, c.type, c.bands, c.sharp, &blender_), "mis_blender_create");
    }
    key_ = c;
    check(cctx_, mis_blender_prepare(blender_, corners.data(), sizes.data(), m), "mis_blender_prepare");
    return c;
}

// the compositing loop's body for the frames `idx` (rois_ from prepare(idx)): batched fused warp + feed
void StitchJob::compose(const std::vector<MisImage>& frames, const std::vector<int>& idx) {
    const int m = (int)idx.size();
    const float scale = median_focal(cams_, idx);
    std::vector<MisImage> fr(m);
    std::vector<float> Ks((size_t)m * 9), Rs((size_t)m * 9);
    for (int k = 0; k < m; k++) {
        fr[k] = frames[idx[k]];
        std::copy(Ks_.begin() + 9 * idx[k], Ks_.begin() + 9 * idx[k] + 9, Ks.begin() + 9 * k);
        std::copy(Rs_.begin() + 9 * idx[k], Rs_.begin() + 9 * idx[k] + 9, Rs.begin() + 9 * k);
    }
    check(cctx_, mis_compose_frames(blender_, fr.data(), m, scale, Ks.data(), Rs.data(), rois_.data()), "mis_compose_frames");
}

void StitchJob::finalize() {
    // the result images are reused while the panorama keeps its size
    if (pano_.data && (pano_.width != key_.pano.width || pano_.height != key_.pano.height)) { mis_image_free(cctx_, &pano_); mis_image_free(cctx_, &mask_); pano_ = MisImage{}; mask_ = MisImage{}; }
    if (!pano_.data) { pano_ = MisImage{}; mask_ = MisImage{}; pano_.mem = mask_.mem = MIS_MEM_DEVICE; }
    check(cctx_, mis_blender_blend(blender_, &pano_, &mask_), "mis_blender_blend");
}

// runs inside mis_match_all_pairs, on the calling thread, once the matcher's device work is enqueued
void StitchJob::hook(void* self_) {
    StitchJob* self = static_cast<StitchJob*>(self_);
    self->hook_ran_ = true;
    try {
        // queue the compose stream behind the 2-NN pass of this matcher call (the one phase that fills the device)
        const int rc = mis_match_knn_fence(self->ctx_, self->cstream_, mis_match_sequence(self->ctx_), 0);
        if (rc < 0) self->check(self->ctx_, rc, "mis_match_knn_fence");
        std::vector<int> everyone(self->n_);
        std::iota(everyone.begin(), everyone.end(), 0);
        self->compose(*self->hook_frames_, everyone);
        self->finalize();
    } catch (const std::exception& e) {
        self->hook_error_ = e.what();
    }
}

void StitchJob::prep_hook(void* self_) {
    StitchJob* self = static_cast<StitchJob*>(self_);
    self->prep_ran_ = true;
    try {
        std::vector<int> everyone(self->n_);
        std::iota(everyone.begin(), everyone.end(), 0);
        self->prepare(everyone);
    } catch (const std::exception& e) { self->prep_error_ = e.what(); }
}

JobOutput StitchJob::run(const std::vector<MisImage>& frames) {
    if ((int)frames.size() != n_) throw std::runtime_error("StitchJob::run: one frame per camera");
    JobOutput out;
    std::vector<int> everyone(n_);
    std::iota(everyone.begin(), everyone.end(), 0);
    // the compose stream is non-blocking: order it behind whatever produced the frames on the main context's stream
    check(cctx_, mis_context_wait(cctx_, ctx_), "mis_context_wait");
    // sizing + zeroing of the panorama pyramids depends on the cameras only: it runs from the finder's hook, once the feature batch is
    // enqueued (warpRoi ends in a synchronisation of the compose stream: in front of the features it kept the main stream idle)
    prep_ran_ = false; prep_error_.clear();
    check(ctx_, mis_orb_on_enqueued(orb_, &StitchJob::prep_hook, this), "mis_orb_on_enqueued");
    // ---- features (:567-622) ----
    std::vector<MisFeatures> feats(n_);
    std::memset(feats.data(), 0, sizeof(MisFeatures) * n_);
    const int rc_f = mis_orb_detect_batch(orb_, frames.data(), n_, feats.data());
    mis_orb_on_enqueued(orb_, nullptr, nullptr);
    check(ctx_, rc_f, "mis_orb_detect_batch");
    if (!prep_ran_) prep_hook(this);     // a batch that returned before its hook
    if (!prep_error_.empty()) throw std::runtime_error(prep_error_);
    for (int i = 0; i < n_; i++) { feats[i].img_idx = i; out.num_features.push_back(feats[i].n); }
    // ---- matching (:647-653) with the speculative composition enqueued from its hook ----
    if (!pairwise_.empty()) { mis_matches_free(pairwise_.data(), (int)pairwise_.size()); pairwise_.clear(); }
    pairwise_.assign((size_t)n_ * n_, MisMatchesInfo{});
    MisMatchParams mp;
    mis_match_default_params(&mp);
    mp.match_conf = cfg_.match_conf;
    hook_frames_ = &frames; hook_ran_ = false; hook_error_.clear();
    check(ctx_, mis_match_on_enqueued(ctx_, &StitchJob::hook, this), "mis_match_on_enqueued");
    const int rc = mis_match_all_pairs(ctx_, feats.data(), n_, &mp, pairwise_.data());
    mis_match_on_enqueued(ctx_, nullptr, nullptr);
    for (auto& f : feats) mis_features_free(ctx_, &f);
    check(ctx_, rc, "mis_match_all_pairs");
    if (!hook_ran_) hook(this);          // a matcher call without pairs returns before its hook
    if (!hook_error_.empty()) throw std::runtime_error(hook_error_);
    // ---- pruning (:215-278) ----
    out.confidence.resize((size_t)n_ * n_);
    for (int k = 0; k < n_ * n_; k++) out.confidence[k] = pairwise_[k].confidence;
    std::vector<int> idx(n_);
    int kept = 0;
    check(ctx_, mis_leave_biggest_component(pairwise_.data(), n_, cfg_.conf_thresh, idx.data(), &kept), "mis_leave_biggest_component");
    idx.resize(kept);
    if (kept < 2) throw std::runtime_error("Need more images");
    out.indices = idx;
    out.speculation_kept = kept == n_;
    if (!out.speculation_kept) {
        // a frame was dropped: the panorama of the kept set (its own scale, roi and band count) replaces the speculated one
        prepare(idx);
        compose(frames, idx);
        finalize();
    }
    check(cctx_, mis_context_synchronize(cctx_), "mis_context_synchronize (compose)");
    out.pano = pano_; out.mask = mask_;
    out.num_bands = key_.bands; out.pano_width = key_.pano.width; out.pano_height = key_.pano.height;
    out.matches = pairwise_;
    return out;
}

}  // namespace mis
