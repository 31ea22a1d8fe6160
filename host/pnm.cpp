// pnm.cpp -- binary PGM / PPM input and output of the host tools (declared in stitcher.hpp); no GPU library needed
#include <fstream>
#include <stdexcept>
#include "stitcher.hpp"

namespace mis {

HostImage readPPM(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Can't open image " + path);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    f >> magic >> w >> h >> maxv;
    f.get();
    if ((magic != "P6" && magic != "P5") || maxv != 255 || w <= 0 || h <= 0) throw std::runtime_error("unsupported PNM file " + path);
    HostImage img;
    img.width = w; img.height = h; img.channels = magic == "P6" ? 3 : 1;
    img.data.resize((size_t)w * h * img.channels);
    f.read((char*)img.data.data(), (std::streamsize)img.data.size());
    if (img.channels == 3)  // PPM stores RGB, the pipeline works on BGR like imread
        for (size_t i = 0; i < (size_t)w * h; i++) std::swap(img.data[3 * i], img.data[3 * i + 2]);
    return img;
}

void writePPM(const std::string& path, const HostImage& img) {
    std::ofstream f(path, std::ios::binary);
    f << (img.channels == 3 ? "P6" : "P5") << "\n" << img.width << " " << img.height << "\n255\n";
    if (img.channels == 3) {
        std::vector<uint8_t> rgb(img.data);
        for (size_t i = 0; i < (size_t)img.width * img.height; i++) std::swap(rgb[3 * i], rgb[3 * i + 2]);
        f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
    } else f.write((const char*)img.data.data(), (std::streamsize)img.data.size());
}

}  // namespace mis
