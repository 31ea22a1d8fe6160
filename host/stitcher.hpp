// stitcher.hpp -- C++17 host side of the MI355X stitching hot path: the sequence of the reference's
// main() (image_stitching/image_stitching.cpp:281-1232) re-authored over the C ABI of libmistitch
// (include/mistitch.h).  No OpenCV: plain PODs, std::vector, and the library's opaque handles.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>
#include "../include/mistitch.h"
#include "rotation.hpp"

namespace mis {

// The reference's globals-as-config (image_stitching.cpp:49-85), same defaults.
struct StitchConfig {
    double work_megapix = -1, seam_megapix = 0.1, compose_megapix = -1;  // compose -1: full-resolution warp + blend
    float conf_thresh = 0.95f;
    float match_conf = 0.32f;
    int blend_type = MIS_BLEND_MULTI_BAND;
    float blend_strength = 5;
    std::string features_type = "orb";     // "orb" | "sift" (:543-563)
    // camera refinement (:681-726).  The reference's default is "reproj"; "no" keeps the supplied (sensor) cameras.
    std::string ba_cost_func = "no";       // "no" | "reproj"
    std::string ba_refine_mask = "_____";  // the reference's default: rotations only
    std::string wave_correct = "horiz";    // "horiz" | "vert" | "no"; applied after the bundle adjustment only
    // seam-scale step (:940-1070, :1162-1171).  The reference's defaults are "gain_blocks" and "dp_color" (both implemented:
    // mis_compensator_*, mis_seam_dp); this struct defaults to the HOT PATH of the north star (no seam-scale step), like
    // image_stitching_amd.StitchConfig.hot_path() -- pass "gain_blocks" / "dp_color" for the reference's configuration.
    std::string expos_comp_type = "no";    // "no" | "gain_blocks" (64 x 64 blocks, 1 feed, 2 filtering passes)
    std::string seam_find_type = "no";     // "no" | "voronoi" | "dp_color" (the reference's default; this driver's default is the hot path)
};

// cv::detail::CameraParams as main() fills it (focal, aspect, ppx, ppy, R, t)
struct CameraParams {
    double focal = 1, aspect = 1, ppx = 0, ppy = 0;
    Mat3<double> R;
    std::array<double, 3> t{};
    Mat3<double> K() const {
        Mat3<double> k;
        k(0, 0) = focal; k(0, 2) = ppx; k(1, 1) = focal * aspect; k(1, 2) = ppy; k(2, 2) = 1;
        return k;
    }
};

struct HostImage {
    int width = 0, height = 0, channels = 0;
    std::vector<uint8_t> data;
};

// "[a,b,...]" -> n x n row-major doubles, n = floor(sqrt(count))  (serializer.cpp:7-36 parseMatrixStr)
std::vector<double> parseMatrixStr(std::string_view sv, int* side);
// "isPortrait;compass;[proj4x4];[view4x4];[camTransform4x4];[K3x3]" -> CameraParams (image_stitching.cpp:413-517)
CameraParams cameraFromImageDescription(const std::string& desc, bool* isPortrait);
HostImage readPPM(const std::string& path);
void writePPM(const std::string& path, const HostImage& img);

struct StitchResult {
    HostImage pano;      // 8UC3 (saturate_cast<uchar> of the 16SC3 result, as imwrite does)
    HostImage mask;      // 8UC1
    std::vector<int> indices;          // images kept by the biggest-component pruning
    std::vector<CameraParams> cameras; // of the kept images, after the optional refinement
    std::vector<int> num_features;
    std::vector<double> confidence;    // n x n
    double t_features = 0, t_matching = 0, t_compositing = 0;
};

class Stitcher {
public:
    explicit Stitcher(int device = 0, const StitchConfig& cfg = StitchConfig());
    ~Stitcher();
    // frames: 8UC3 BGR of one size; cameras: one per frame (sensor / ground-truth K, R)
    StitchResult stitch(const std::vector<HostImage>& frames, const std::vector<CameraParams>& cameras);

private:
    void check(int rc, const char* what) const;
    StitchConfig cfg_;
    MisContext* ctx_ = nullptr;
};

}  // namespace mis
