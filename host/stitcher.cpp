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
    MisOrbParams op;
    mis_orb_default_params(&op);
    MisOrb* orb = nullptr;
    check(mis_orb_create(ctx_, &op, W, H, &orb), "mis_orb_create");
    std::vector<MisFeatures> features(n);
    for (int i = 0; i < n; i++) {
        MisImage v = view(frames[i]);
        check(mis_orb_detect(orb, &v, &features[i]), "mis_orb_detect");
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
    mis_matches_free(pairwise.data(), n * n);
    for (auto& f : features) mis_features_free(ctx_, &f);
    mis_orb_destroy(orb);
    out.t_matching = now() - t;
    if (kept < 2) throw std::runtime_error("Need more images");
    // (bundle adjustment / wave correction are outside the hot path: the supplied cameras are used as is)

    // ---- warped image scale = median focal (:884-895) ----
    std::vector<double> focals;
    for (int i : out.indices) focals.push_back(cameras[i].focal);
    std::sort(focals.begin(), focals.end());
    float warped_image_scale = focals.size() % 2 == 1 ? static_cast<float>(focals[focals.size() / 2])
                                                      : static_cast<float>(focals[focals.size() / 2 - 1] + focals[focals.size() / 2]) * 0.5f;

    // ---- compositing (:1086-1228), compose_scale = 1 ----
    std::cout << "Compositing..." << std::endl;
    t = now();
    std::vector<MisPoint> corners(kept);
    std::vector<MisSize> sizes(kept);
    std::vector<std::array<float, 9>> Ks(kept), Rs(kept);
    for (int k = 0; k < kept; k++) {
        const CameraParams& c = cameras[out.indices[k]];
        Mat3<float> K = c.K().cast<float>(), R = c.R.cast<float>();
        std::copy(K.m.begin(), K.m.end(), Ks[k].begin());
        std::copy(R.m.begin(), R.m.end(), Rs[k].begin());
        MisRect roi;
        mis_warp_roi(warped_image_scale, W, H, Ks[k].data(), Rs[k].data(), &roi);
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
    for (int k = 0; k < kept; k++) {
        std::cout << "Compositing image #" << out.indices[k] + 1 << std::endl;
        MisImage src = view(frames[out.indices[k]]), img_warped_s{}, mask_warped{};
        MisPoint tl;
        check(mis_warp_spherical_fused(ctx_, &src, warped_image_scale, Ks[k].data(), Rs[k].data(), &img_warped_s, &mask_warped, &tl), "mis_warp_spherical_fused");
        check(mis_blender_feed(blender, &img_warped_s, &mask_warped, tl), "mis_blender_feed");
        mis_image_free(ctx_, &img_warped_s);
        mis_image_free(ctx_, &mask_warped);
    }
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
