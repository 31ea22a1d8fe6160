// stitcher.cpp -- see stitcher.hpp.  Stage order and log lines follow the reference's main().
#include "stitcher.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

namespace mis {

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::vector<double> parseMatrixStr(std::string_view sv, int* side) {
    sv = sv.substr(1, sv.size() - 2);
    std::vector<std::string> items;
    for (auto pos = sv.find(','); pos != sv.npos; pos = sv.find(',')) { items.emplace_back(sv.substr(0, pos)); sv = sv.substr(pos + 1); }
    items.emplace_back(sv);
    int len = (int)std::sqrt((double)items.size());
    std::vector<double> m((size_t)len * len);
    for (int i = 0; i < len * len; i++) m[i] = std::strtod(items[i].c_str(), nullptr);
    if (side) *side = len;
    return m;
}

CameraParams cameraFromImageDescription(const std::string& desc, bool* isPortrait) {
    std::vector<std::string> parts;
    std::string_view sv(desc);
    for (int i = 0; i < 5; i++) {
        auto pos = sv.find(';');
        if (pos == sv.npos) throw std::runtime_error("ImageDescription needs 6 ';'-separated fields");
        parts.emplace_back(sv.substr(0, pos));
        sv = sv.substr(pos + 1);
    }
    while (!sv.empty() && (sv.back() == '\n' || sv.back() == '\r' || sv.back() == ' ')) sv.remove_suffix(1);
    parts.emplace_back(sv);
    const bool portrait = std::strtol(parts[0].c_str(), nullptr, 10) != 0;
    int n4 = 0, n3 = 0;
    std::vector<double> cam = parseMatrixStr(parts[4], &n4), K = parseMatrixStr(parts[5], &n3);
    if (n4 != 4 || n3 != 3) throw std::runtime_error("ImageDescription: camera transform must be 4x4 and K 3x3");
    CameraParams p;
    p.aspect = 1.0;
    p.focal = K[1 * 3 + 1];
    if (portrait) { p.ppx = K[1 * 3 + 2]; p.ppy = K[0 * 3 + 2]; } else { p.ppx = K[0 * 3 + 2]; p.ppy = K[1 * 3 + 2]; }
    Mat3<double> R;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = cam[r * 4 + c];
    p.t = {cam[3], cam[7], cam[11]};
    p.R = rehandCameraRotation(R, portrait);
    if (isPortrait) *isPortrait = portrait;
    return p;
}

Stitcher::Stitcher(int device, const StitchConfig& cfg) : cfg_(cfg) {
    int rc = mis_context_create(device, nullptr, &ctx_);
    if (rc != MIS_OK) throw std::runtime_error("mis_context_create failed: no HIP device (there is no CPU fallback)");
}
Stitcher::~Stitcher() { mis_context_destroy(ctx_); }

void Stitcher::check(int rc, const char* what) const {
    if (rc != MIS_OK) throw std::runtime_error(std::string(what) + ": " + mis_last_error(ctx_));
}

static MisImage view(const HostImage& im) {
    MisImage v{};
    v.data = (void*)im.data.data(); v.width = im.width; v.height = im.height; v.channels = im.channels;
    v.stride = (size_t)im.width * im.channels; v.dtype = MIS_U8; v.mem = MIS_MEM_HOST;
    return v;
}

StitchResult Stitcher::stitch(const std::vector<HostImage>& frames, const std::vector<CameraParams>& cams_in) {
    const int n = (int)frames.size();
    if (n < 2 || (int)cams_in.size() != n) throw std::runtime_error("Need more images");
    StitchResult out;
    std::vector<CameraParams> cameras = cams_in;
    const int W = frames[0].width, H = frames[0].height;

    // ---- features (image_stitching.cpp:545, :567-622; work_megapix = -1: full resolution) ----
    double t = now();
    if (cfg_.features_type != "orb" && cfg_.features_type != "sift") throw std::runtime_error("Unknown 2D features type: '" + cfg_.features_type + "'.");
    if (cfg_.ba_cost_func != "no" && cfg_.ba_cost_func != "reproj")
        throw std::runtime_error("bundle adjustment cost function '" + cfg_.ba_cost_func + "' is not implemented (only 'no' and 'reproj')");
    if (cfg_.expos_comp_type != "no" && cfg_.expos_comp_type != "gain_blocks")
        throw std::runtime_error("exposure compensation '" + cfg_.expos_comp_type + "' is not implemented (only 'no' and 'gain_blocks')");
    if (cfg_.seam_find_type != "no" && cfg_.seam_find_type != "voronoi" && cfg_.seam_find_type != "dp_color")
        throw std::runtime_error("seam finder '" + cfg_.seam_find_type + "' is not implemented ('no', 'voronoi' and 'dp_color' are)");
    MisOrb* orb = nullptr;
    MisSift* sift = nullptr;
    std::vector<MisFeatures> features(n);
    std::vector<MisImage> views(n);
    for (int i = 0; i < n; i++) views[i] = view(frames[i]);
    if (cfg_.features_type == "sift") {
        check(mis_sift_create(ctx_, nullptr, W, H, &sift), "mis_sift_create");
        check(mis_sift_detect_batch(sift, views.data(), n, features.data()), "mis_sift_detect_batch");
    } else {
        MisOrbParams op;
        mis_orb_default_params(&op);
        check(mis_orb_create(ctx_, &op, W, H, &orb), "mis_orb_create");
        check(mis_orb_detect_batch(orb, views.data(), n, features.data()), "mis_orb_detect_batch");
    }
    for (int i = 0; i < n; i++) {
        features[i].img_idx = i;
        std::cout << "Features in image #" << i + 1 << ": " << features[i].n << std::endl;
        out.num_features.push_back(features[i].n);
    }
    out.t_features = now() - t;

    // ---- pairwise matching (:647-655) and pruning (:661) ----
    t = now();
    MisMatchParams mp;
    mis_match_default_params(&mp);
    mp.match_conf = cfg_.match_conf;
    std::vector<MisMatchesInfo> pairwise((size_t)n * n);
    check(mis_match_all_pairs(ctx_, features.data(), n, &mp, pairwise.data()), "mis_match_all_pairs");
    for (auto& m : pairwise) out.confidence.push_back(m.confidence);
    out.indices.resize(n);
    int kept = 0;
    mis_leave_biggest_component(pairwise.data(), n, cfg_.conf_thresh, out.indices.data(), &kept);
    out.indices.resize(kept);
    // ---- camera refinement on the kept subset (:671-726): bundle adjustment (reprojection cost), wave correction ----
    if (kept >= 2 && cfg_.ba_cost_func == "reproj") {
        std::vector<MisFeatures> fsub(kept);
        std::vector<MisMatchesInfo> psub((size_t)kept * kept);
        std::vector<MisCameraParams> cp(kept);
        for (int a = 0; a < kept; a++) {
            fsub[a] = features[out.indices[a]];
            fsub[a].img_idx = a;
            for (int b = 0; b < kept; b++) {
                psub[(size_t)a * kept + b] = pairwise[(size_t)out.indices[a] * n + out.indices[b]];   // borrowed arrays
                psub[(size_t)a * kept + b].src_img_idx = a; psub[(size_t)a * kept + b].dst_img_idx = b;
            }
            const CameraParams& c = cameras[out.indices[a]];
            cp[a].focal = c.focal; cp[a].aspect = c.aspect; cp[a].ppx = c.ppx; cp[a].ppy = c.ppy;
            std::copy(c.R.m.begin(), c.R.m.end(), cp[a].R);
            std::copy(c.t.begin(), c.t.end(), cp[a].t);
        }
        if (mis_bundle_adjust_reproj(ctx_, fsub.data(), psub.data(), kept, cfg_.conf_thresh, cfg_.ba_refine_mask.c_str(), cp.data()) != MIS_OK)
            throw std::runtime_error(std::string("Camera parameters adjusting failed: ") + mis_last_error(ctx_));
        if (cfg_.wave_correct != "no") {
            std::vector<double> rm((size_t)kept * 9);
            for (int a = 0; a < kept; a++) std::copy(cp[a].R, cp[a].R + 9, rm.begin() + 9 * a);
            check(mis_wave_correct(rm.data(), kept, cfg_.wave_correct == "vert" ? 1 : 0), "mis_wave_correct");
            for (int a = 0; a < kept; a++) std::copy(rm.begin() + 9 * a, rm.begin() + 9 * a + 9, cp[a].R);
        }
        for (int a = 0; a < kept; a++) {
            CameraParams& c = cameras[out.indices[a]];
            c.focal = cp[a].focal; c.aspect = cp[a].aspect; c.ppx = cp[a].ppx; c.ppy = cp[a].ppy;
            std::copy(cp[a].R, cp[a].R + 9, c.R.m.begin());
            std::copy(cp[a].t, cp[a].t + 3, c.t.begin());
        }
    }
    mis_matches_free(pairwise.data(), n * n);
    for (auto& f : features) mis_features_free(ctx_, &f);
    if (orb) mis_orb_destroy(orb);
    if (sift) mis_sift_destroy(sift);
    out.t_matching = now() - t;
    if (kept < 2) throw std::runtime_error("Need more images");
    out.cameras.clear();
    for (int i : out.indices) out.cameras.push_back(cameras[i]);

    // ---- warped image scale = median focal (:884-895) ----
    std::vector<double> focals;
    for (int i : out.indices) focals.push_back(cameras[i].focal);
    std::sort(focals.begin(), focals.end());
    float warped_image_scale = focals.size() % 2 == 1 ? static_cast<float>(focals[focals.size() / 2])
                                                      : static_cast<float>(focals[focals.size() / 2 - 1] + focals[focals.size() / 2]) * 0.5f;

    // ---- compositing (:1086-1228) at compose scale (:1105-1146): the warper's scale and the intrinsics times compose_work_aspect,
    //      frames (and their sizes, cvRound) resized only when |compose_scale - 1| > 0.1, as the reference tests it ----
    std::cout << "Compositing..." << std::endl;
    t = now();
    double compose_scale = 1.0;
    if (cfg_.compose_megapix > 0) compose_scale = std::min(1.0, std::sqrt(cfg_.compose_megapix * 1e6 / ((double)W * H)));
    const double compose_work_aspect = compose_scale / 1.0;     // work_scale = 1: features at full resolution
    const bool compose_resized = std::abs(compose_scale - 1) > 1e-1;
    const int cW = compose_resized ? (int)std::nearbyint(W * compose_scale) : W, cH = compose_resized ? (int)std::nearbyint(H * compose_scale) : H;
    const float compose_warp_scale = warped_image_scale * static_cast<float>(compose_work_aspect);
    std::vector<MisPoint> corners(kept);
    std::vector<MisSize> sizes(kept);
    std::vector<std::array<float, 9>> Ks(kept), Rs(kept), cKs(kept);     // Ks: work-scale intrinsics (the seam step), cKs: the compositing loop's
    for (int k = 0; k < kept; k++) {
        const CameraParams& c = cameras[out.indices[k]];
        CameraParams cc = c;
        cc.focal *= compose_work_aspect; cc.ppx *= compose_work_aspect; cc.ppy *= compose_work_aspect;
        Mat3<float> K = c.K().cast<float>(), R = c.R.cast<float>(), cK = cc.K().cast<float>();
        std::copy(K.m.begin(), K.m.end(), Ks[k].begin());
        std::copy(R.m.begin(), R.m.end(), Rs[k].begin());
        std::copy(cK.m.begin(), cK.m.end(), cKs[k].begin());
        MisRect roi;
        mis_warp_roi(compose_warp_scale, cW, cH, cKs[k].data(), Rs[k].data(), &roi);
        corners[k] = {roi.x, roi.y};
        sizes[k] = {roi.width, roi.height};
    }
    MisRect pano;
    mis_result_roi(corners.data(), sizes.data(), kept, &pano);
    int btype = 0, bands = 0;
    float sharp = 0;
    mis_blend_config(cfg_.blend_type, cfg_.blend_strength, pano.width, pano.height, &btype, &bands, &sharp);
    if (btype == MIS_BLEND_MULTI_BAND) std::cout << "Multi-band blender, number of bands: " << bands << std::endl;
    else if (btype == MIS_BLEND_FEATHER) std::cout << "Feather blender, sharpness: " << sharp << std::endl;
    MisBlender* blender = nullptr;
    check(mis_blender_create(ctx_, btype, bands, sharp, &blender), "mis_blender_create");
    check(mis_blender_prepare(blender, corners.data(), sizes.data(), kept), "mis_blender_prepare");
    // ---- seam-scale pass (:604-622 resize, :973-990 warp, :1002-1023 exposure compensator, :1029-1065 seam finder) ----
    const bool seam_step = cfg_.expos_comp_type != "no" || cfg_.seam_find_type != "no";
    MisCompensator* compensator = nullptr;
    std::vector<MisImage> masks_warped(kept);
    if (seam_step) {
        const double seam_scale = std::min(1.0, std::sqrt(cfg_.seam_megapix * 1e6 / ((double)W * H)));
        const float swa = (float)seam_scale;   // seam_work_aspect with work_scale = 1
        const float seam_warp_scale = warped_image_scale * swa;
        std::vector<MisImage> images_warped(kept);
        std::vector<MisPoint> seam_corners(kept);
        for (int k = 0; k < kept; k++) {
            MisImage full = view(frames[out.indices[k]]), img{};
            if (seam_scale < 1.0) check(mis_resize_linear_exact(ctx_, &full, 0, 0, seam_scale, seam_scale, &img), "mis_resize_linear_exact");
            else img = full;
            std::array<float, 9> Ks_ = Ks[k];
            Ks_[0] *= swa; Ks_[2] *= swa; Ks_[4] *= swa; Ks_[5] *= swa;
            check(mis_warp_spherical(ctx_, &img, seam_warp_scale, Ks_.data(), Rs[k].data(), MIS_INTER_LINEAR, MIS_BORDER_REFLECT, &images_warped[k], &seam_corners[k]),
                  "mis_warp_spherical (seam scale)");
            std::vector<uint8_t> ones((size_t)img.width * img.height, 255);
            MisImage m{ones.data(), img.width, img.height, 1, (size_t)img.width, MIS_U8, MIS_MEM_HOST};
            MisPoint tl;
            check(mis_warp_spherical(ctx_, &m, seam_warp_scale, Ks_.data(), Rs[k].data(), MIS_INTER_NEAREST, MIS_BORDER_CONSTANT, &masks_warped[k], &tl),
                  "mis_warp_spherical (seam-scale mask)");
            if (seam_scale < 1.0) mis_image_free(ctx_, &img);
        }
        if (cfg_.expos_comp_type == "gain_blocks") {
            check(mis_compensator_create(ctx_, 64, 64, 2, &compensator), "mis_compensator_create");
            check(mis_compensator_feed(compensator, seam_corners.data(), images_warped.data(), masks_warped.data(), kept), "mis_compensator_feed");
        }
        if (cfg_.seam_find_type == "voronoi") check(mis_seam_voronoi(ctx_, seam_corners.data(), masks_warped.data(), kept), "mis_seam_voronoi");
        else if (cfg_.seam_find_type == "dp_color")      // the reference's default (image_stitching.cpp:77, :1056-1065)
            check(mis_seam_dp(ctx_, seam_corners.data(), images_warped.data(), masks_warped.data(), kept, MIS_SEAM_DP_COLOR), "mis_seam_dp");
        for (auto& im : images_warped) mis_image_free(ctx_, &im);
    }
    for (int k = 0; k < kept; k++) {
        std::cout << "Compositing image #" << out.indices[k] + 1 << std::endl;
        MisImage full = view(frames[out.indices[k]]), src{}, img_warped_s{}, mask_warped{};
        if (compose_resized) check(mis_resize_linear_exact(ctx_, &full, 0, 0, compose_scale, compose_scale, &src), "mis_resize_linear_exact (compose scale)");
        else src = full;
        MisPoint tl;
        check(mis_warp_spherical_fused(ctx_, &src, compose_warp_scale, cKs[k].data(), Rs[k].data(), &img_warped_s, &mask_warped, &tl), "mis_warp_spherical_fused");
        if (compose_resized) mis_image_free(ctx_, &src);
        if (compensator) check(mis_compensator_apply(compensator, k, &img_warped_s), "mis_compensator_apply");   // :1162
        if (seam_step) {
            check(mis_seam_mask_apply(ctx_, &masks_warped[k], &mask_warped), "mis_seam_mask_apply");                // :1169-1171
            mis_image_free(ctx_, &masks_warped[k]);
        }
        check(mis_blender_feed(blender, &img_warped_s, &mask_warped, tl), "mis_blender_feed");
        mis_image_free(ctx_, &img_warped_s);
        mis_image_free(ctx_, &mask_warped);
    }
    if (compensator) mis_compensator_destroy(compensator);
    std::vector<int16_t> res16((size_t)pano.width * pano.height * 3);
    out.mask.width = pano.width; out.mask.height = pano.height; out.mask.channels = 1;
    out.mask.data.resize((size_t)pano.width * pano.height);
    MisImage result{res16.data(), pano.width, pano.height, 3, (size_t)pano.width * 6, MIS_S16, MIS_MEM_HOST};
    MisImage result_mask{out.mask.data.data(), pano.width, pano.height, 1, (size_t)pano.width, MIS_U8, MIS_MEM_HOST};
    check(mis_blender_blend(blender, &result, &result_mask), "mis_blender_blend");
    mis_blender_destroy(blender);
    out.t_compositing = now() - t;
    std::cout << "Compositing, time: " << out.t_compositing << " sec" << std::endl;
    // imwrite converts the 16S result with saturate_cast<uchar>
    out.pano.width = pano.width; out.pano.height = pano.height; out.pano.channels = 3;
    out.pano.data.resize(res16.size());
    for (size_t i = 0; i < res16.size(); i++) out.pano.data[i] = (uint8_t)std::clamp<int>(res16[i], 0, 255);
    return out;
}

}  // namespace mis
