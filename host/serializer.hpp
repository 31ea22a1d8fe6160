// serializer.hpp -- the `cams.data` / `indices.data` stage checkpoint of the reference
// (image_stitching/serializer.cpp:38-193, used at image_stitching/image_stitching.cpp:651-720): the on-disk format
// either side of the match stage, so that warp + blend can run from a checkpoint written by the reference and
// vice versa.  Same function names and text format; the file paths are arguments (the reference hard-codes
// "./cams.data" and "./indices.data", which stay the defaults).
//
//   cams.data     one line per camera:  aspect@focal@ppx@ppy@<t>@<R>
//   <matrix>      "[" then every element followed by "," -- or by ";" at a row end -- then "]", elements printed
//                 with operator<< (precision 6, %g style); read back with strtold into a float (CV_32F) matrix
//   indices.data  one decimal index per line
#pragma once
#include <string>
#include <string_view>
#include <vector>
#include "stitcher.hpp"

namespace mis {

// a small row-major matrix as deserializeMatrix returns it (CV_32F)
struct MatF {
    int rows = 0, cols = 0;
    std::vector<float> v;
};

std::vector<std::string> splitMatrixStrItems(std::string_view sv);                        // serializer.cpp:7-20
std::string serializeMatrix(const double* m, int rows, int cols);                          // serializer.cpp:38-66 (CV_64F)
std::string serializeMatrix(const float* m, int rows, int cols);                           //                  (CV_32F)
MatF deserializeMatrix(std::string s);                                                     // serializer.cpp:68-111
void serializeCameraParams(const std::vector<CameraParams>& cams, const std::string& path = "./cams.data");     // :113-127
std::vector<CameraParams> deserializeCameraParams(const std::string& path = "./cams.data");                     // :129-170
void serializeIndices(const std::vector<int>& indices, const std::string& path = "./indices.data");             // :172-180
std::vector<int> deserializeIndices(const std::string& path = "./indices.data");                                // :182-193

}  // namespace mis
