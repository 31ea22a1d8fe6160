// stitch_main.cpp -- command-line driver mirroring the reference's main() (image_stitching.cpp:281):
//   stitch_main <dir>
// <dir> holds frames "<k>.ppm" (binary PPM; libjpeg / libexif are not available in this environment)
// and, per frame, "<k>.txt" with the phone app's EXIF ImageDescription string
// "isPortrait;compass;[proj 4x4];[view 4x4];[cameraTransform 4x4];[K 3x3]" (image_stitching.cpp:413-445).
// Frames are sorted by the leading integer of the file name (:327-335).  Writes result.ppm / result_mask.pgm.
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include "stitcher.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cout << "usage: " << argv[0] << " <image directory>\n";
        return -1;
    }
    namespace fs = std::filesystem;
    std::vector<std::string> img_names;
    for (auto& e : fs::directory_iterator(argv[1])) {
        std::string ext = e.path().extension().string();
        std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
        if (ext == ".ppm") img_names.push_back(e.path().string());
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
            std::ifstream f(fs::path(name).replace_extension(".txt"));
            if (!f) { std::cout << "Can't open camera description for " << name << "\n"; return -1; }
            std::stringstream ss;
            ss << f.rdbuf();
            bool portrait = false;
            cams.push_back(mis::cameraFromImageDescription(ss.str(), &portrait));
        }
        mis::Stitcher st(0);
        mis::StitchResult r = st.stitch(frames, cams);
        mis::writePPM((fs::path(argv[1]) / "result.ppm").string(), r.pano);
        mis::writePPM((fs::path(argv[1]) / "result_mask.pgm").string(), r.mask);
        std::cout << "result " << r.pano.width << "x" << r.pano.height << ", kept " << r.indices.size() << " of " << frames.size() << " images\n";
    } catch (const std::exception& e) {
        std::cout << e.what() << "\n";
        return 1;
    }
    return 0;
}
