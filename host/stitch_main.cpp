// stitch_main.cpp -- command-line driver mirroring the reference's main() (image_stitching.cpp:281):
//   stitch_main <dir>
// <dir> holds frames "<k>.ppm" (binary PPM; libjpeg / libexif are not available in this environment)
// and, per frame, "<k>.txt" with the phone app's EXIF ImageDescription string
// "isPortrait;compass;[proj 4x4];[view 4x4];[cameraTransform 4x4];[K 3x3]" (image_stitching.cpp:413-445).
// Frames are sorted by the leading integer of the file name (:327-335).  Writes result.ppm / result_mask.pgm.
#include <algorithm>
#include <cctype>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include "stitcher.hpp"
#include "exif.hpp"
#include "serializer.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cout << "usage: " << argv[0] << " <image directory> [--features orb|sift] [--ba no|reproj] [--ba_refine_mask xxxxx] [--wave_correct horiz|vert|no]\n"
                     "       [--expos_comp no|gain_blocks] [--seam no|voronoi|dp_color] [--blend no|feather|multiband] [--conf_thresh f] [--match_conf f] [--compose_megapix f] [--seam_megapix f]\n"
                     "(the reference sets these as globals, image_stitching.cpp:49-85)\n";
        return -1;
    }
    mis::StitchConfig cfg;
    for (int i = 2; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--features") cfg.features_type = v;
        else if (k == "--ba") cfg.ba_cost_func = v;
        else if (k == "--ba_refine_mask") cfg.ba_refine_mask = v;
        else if (k == "--wave_correct") cfg.wave_correct = v;
        else if (k == "--expos_comp") cfg.expos_comp_type = v;
        else if (k == "--seam") cfg.seam_find_type = v;
        else if (k == "--blend") cfg.blend_type = v == "no" ? MIS_BLEND_NO : (v == "feather" ? MIS_BLEND_FEATHER : MIS_BLEND_MULTI_BAND);
        else if (k == "--compose_megapix") cfg.compose_megapix = std::strtod(v.c_str(), nullptr);
        else if (k == "--seam_megapix") cfg.seam_megapix = std::strtod(v.c_str(), nullptr);
        else if (k == "--conf_thresh") cfg.conf_thresh = std::strtof(v.c_str(), nullptr);
        else if (k == "--match_conf") cfg.match_conf = std::strtof(v.c_str(), nullptr);
        else { std::cout << "unknown option " << k << "\n"; return -1; }
    }
    namespace fs = std::filesystem;
    std::vector<std::string> img_names;
    for (auto& e : fs::directory_iterator(argv[1])) {
        std::string ext = e.path().extension().string();
        std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
        const std::string stem = e.path().stem().string();
        const bool numbered = !stem.empty() && std::all_of(stem.begin(), stem.end(), [](unsigned char ch) { return std::isdigit(ch) != 0; });
        if (ext == ".ppm" && numbered) img_names.push_back(e.path().string());   // 1.ppm, 2.ppm, ...: not an earlier run's result.ppm
    }
    std::sort(img_names.begin(), img_names.end(), [](const std::string& a, const std::string& b) {
        return std::strtol(fs::path(a).filename().string().c_str(), nullptr, 10) < std::strtol(fs::path(b).filename().string().c_str(), nullptr, 10);
    });
    if (img_names.size() < 2) {
        std::cout << "Need more images\n";
        return -1;
    }
    try {
        std::vector<mis::HostImage> frames;
        std::vector<mis::CameraParams> cams;
        for (auto& name : img_names) {
            frames.push_back(mis::readPPM(name));
            // the camera: the EXIF ImageDescription of "<k>.jpg" beside the frame when there is one (the reference's source,
            // image_stitching.cpp:344-347, :411-417; the pixels still come from the .ppm: no JPEG decoder here), else "<k>.txt"
            std::string desc;
            if (!mis::exifImageDescriptionFile(fs::path(name).replace_extension(".jpg").string(), &desc)) {
                std::ifstream f(fs::path(name).replace_extension(".txt"));
                if (!f) { std::cout << "Can't open camera description for " << name << "\n"; return -1; }
                std::stringstream ss;
                ss << f.rdbuf();
                desc = ss.str();
            }
            bool portrait = false;
            cams.push_back(mis::cameraFromImageDescription(desc, &portrait));
        }
        mis::Stitcher st(0, cfg);
        mis::StitchResult r = st.stitch(frames, cams);
        // the reference checkpoints the refined cameras and the kept indices (image_stitching.cpp:707-708)
        mis::serializeCameraParams(r.cameras, (fs::path(argv[1]) / "cams.data").string());
        mis::serializeIndices(r.indices, (fs::path(argv[1]) / "indices.data").string());
        mis::writePPM((fs::path(argv[1]) / "result.ppm").string(), r.pano);
        mis::writePPM((fs::path(argv[1]) / "result_mask.pgm").string(), r.mask);
        std::cout << "result " << r.pano.width << "x" << r.pano.height << ", kept " << r.indices.size() << " of " << frames.size() << " images\n";
    } catch (const std::exception& e) {
        std::cout << e.what() << "\n";
        return 1;
    }
    return 0;
}
