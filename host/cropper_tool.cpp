// cropper_tool -- test aid (tests/test_host_cpp.py): reads a binary PGM/PPM (P5 / P6), then
//   cropper_tool contour IN.pgm            prints the external contours: one line "n x0 y0 x1 y1 ..." each
//   cropper_tool fill IN.pgm OUT.pgm       largest contour filled (drawContours FILLED)
//   cropper_tool crop IN.ppm OUT.ppm       crop(); prints "x y w h"
#include <cstdio>
#include <cstring>
#include <exception>
#include "cropper.hpp"

int main(int argc, char** argv) {
    try {
        if (argc == 3 && !std::strcmp(argv[1], "contour")) {
            mis::HostImage m = mis::readPPM(argv[2]);
            for (const auto& c : mis::findExternalContours(m)) {
                std::printf("%zu", c.size());
                for (const auto& p : c) std::printf(" %d %d", p.x, p.y);
                std::printf("\n");
            }
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "fill")) {
            mis::HostImage m = mis::readPPM(argv[2]);
            auto cs = mis::findExternalContours(m);
            size_t id = 0;
            for (size_t i = 0; i < cs.size(); i++) if (cs[i].size() > cs[id].size()) id = i;
            mis::writePPM(argv[3], mis::fillContour(cs.at(id), m.width, m.height));
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "crop")) {
            mis::HostImage m = mis::readPPM(argv[2]);
            mis::Rect r = mis::crop(m);
            mis::writePPM(argv[3], m);
            std::printf("%d %d %d %d\n", r.x, r.y, r.width, r.height);
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: cropper_tool contour IN | fill IN OUT | crop IN OUT\n");
    return 1;
}
