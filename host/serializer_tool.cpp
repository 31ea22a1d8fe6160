// serializer_tool -- round-trip aid for tests/test_host_cpp.py:
//   serializer_tool copy-cams IN OUT        read cams.data text IN, write it back to OUT
//   serializer_tool copy-indices IN OUT
//   serializer_tool matrix "<text>"          deserializeMatrix + print rows x cols and the float values (%.9g)
//   serializer_tool exif FILE.jpg            the EXIF ImageDescription of a JPEG file (exit code 3 when it has none)
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include "exif.hpp"
#include "serializer.hpp"

int main(int argc, char** argv) {
    try {
        if (argc == 4 && !std::strcmp(argv[1], "copy-cams")) { mis::serializeCameraParams(mis::deserializeCameraParams(argv[2]), argv[3]); return 0; }
        if (argc == 4 && !std::strcmp(argv[1], "copy-indices")) { mis::serializeIndices(mis::deserializeIndices(argv[2]), argv[3]); return 0; }
        if (argc == 3 && !std::strcmp(argv[1], "matrix")) {
            mis::MatF m = mis::deserializeMatrix(argv[2]);
            std::printf("%d %d", m.rows, m.cols);
            for (float v : m.v) std::printf(" %.9g", v);
            std::printf("\n");
            return 0;
        }
        if (argc == 3 && !std::strcmp(argv[1], "exif")) {
            std::string d;
            if (!mis::exifImageDescriptionFile(argv[2], &d)) return 3;
            std::fwrite(d.data(), 1, d.size(), stdout);
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: serializer_tool copy-cams IN OUT | copy-indices IN OUT | matrix TEXT | exif FILE.jpg\n");
    return 1;
}
