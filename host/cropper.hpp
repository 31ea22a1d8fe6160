// cropper.hpp -- the reference's cropper API (image_stitching/cropper.h:1-10, cropper.cpp:6-209) over plain buffers:
// the largest inscribed axis-aligned rectangle heuristic applied to a stitched panorama.  Same four entry points,
// same semantics; cv::Mat / cv::Rect / cv::Point become HostImage / Rect / Point.
//
// OpenCV pieces restated (PARITY UNPINNED like every OpenCV-backed stage):
//   findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE)  the 8-neighbour border following of contours.cpp
//                                                          (Suzuki-Abe): every border pixel in visiting order
//   drawContours(..., FILLED)                             the region enclosed by the contour (holes filled)
#pragma once
#include <vector>
#include "stitcher.hpp"

namespace mis {

struct Point { int x = 0, y = 0; };
struct Rect { int x = 0, y = 0, width = 0, height = 0; };

// cropper.cpp:6-112 -- true when no exterior (zero) pixel lies on the rectangle's border; otherwise sets one or two of
// the out-codes to 1: the side(s) with the most exterior pixels (the reference's tie rules)
bool checkInteriorExterior(const HostImage& mask /* 8UC1 */, const Rect& croppingMask, int& top, int& bottom, int& left, int& right);
bool compareX(Point a, Point b);   // cropper.cpp:114-117
bool compareY(Point a, Point b);   // cropper.cpp:119-122
// cropper.cpp:124-209 -- source: 8UC3 (or 8UC1); replaced by its crop.  Returns the rectangle used.
Rect crop(HostImage& source);

// the restated OpenCV pieces, exposed for the tests
std::vector<std::vector<Point>> findExternalContours(const HostImage& mask /* 8UC1, non-zero = foreground */);
HostImage fillContour(const std::vector<Point>& contour, int width, int height);   // 8UC1, 255 inside or on the contour

}  // namespace mis
