// cropper.cpp -- see cropper.hpp
#include "cropper.hpp"
#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace mis {

bool checkInteriorExterior(const HostImage& mask, const Rect& r, int& top, int& bottom, int& left, int& right) {
    if (mask.channels != 1 || r.x < 0 || r.y < 0 || r.width <= 0 || r.height <= 0 || r.x + r.width > mask.width || r.y + r.height > mask.height)
        throw std::runtime_error("checkInteriorExterior: rectangle outside the mask");
    auto at = [&](int y, int x) { return mask.data[(size_t)(r.y + y) * mask.width + (r.x + x)]; };
    bool result = true;
    int top_row = 0, bottom_row = 0, left_column = 0, right_column = 0;
    for (int x = 0; x < r.width; ++x) if (at(0, x) == 0) { result = false; ++top_row; }
    for (int x = 0; x < r.width; ++x) if (at(r.height - 1, x) == 0) { result = false; ++bottom_row; }
    for (int y = 0; y < r.height; ++y) if (at(y, 0) == 0) { result = false; ++left_column; }
    for (int y = 0; y < r.height; ++y) if (at(y, r.width - 1) == 0) { result = false; ++right_column; }
    // the side with the most exterior pixels is the one to move (cropper.cpp:64-109, including its tie behaviour)
    if (top_row > bottom_row) {
        if (top_row > left_column && top_row > right_column) top = 1;
    } else if (bottom_row > left_column) {
        if (bottom_row > right_column) bottom = 1;
    }
    if (left_column >= right_column) {
        if (left_column >= bottom_row && left_column >= top_row) left = 1;
    } else if (right_column >= top_row) {
        if (right_column >= bottom_row) right = 1;
    }
    return result;
}

bool compareX(Point a, Point b) { return a.x < b.x; }
bool compareY(Point a, Point b) { return a.y < b.y; }

// contours.cpp: raster scan for outer-border starts (a foreground pixel whose west neighbour is background and that no
// earlier border passed), RETR_EXTERNAL (borders inside an already traced border are skipped), 8-neighbour border
// following with the direction codes 0 = E, 1 = NE, 2 = N, 3 = NW, 4 = W, 5 = SW, 6 = S, 7 = SE (y grows downwards).
std::vector<std::vector<Point>> findExternalContours(const HostImage& mask) {
    if (mask.channels != 1) throw std::runtime_error("findExternalContours: 8UC1 mask expected");
    const int w = mask.width, h = mask.height, step = w + 2;
    std::vector<int> img((size_t)step * (h + 2), 0);   // 1-pixel zero frame; 0 background, 1 foreground, other = border labels
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = mask.data[(size_t)y * w + x] ? 1 : 0;
    const int deltas[16] = {1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1, 1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1};
    const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    std::vector<std::vector<Point>> contours;
    int nbd = 2;
    for (int y = 1; y <= h; y++) {
        // lnbd: value of the last border pixel met on this row (0 = the frame).  Positive: we entered that border through
        // its left side and are inside it; negative: we left it through a pixel whose east neighbour is background.
        int lnbd = 0, prev = 0;
        for (int x = 1; x <= w; x++) {
            int p = img[(size_t)y * step + x];
            if (p == prev) continue;
            if (prev == 0 && p == 1) {
                if (!(lnbd > 0)) {   // RETR_EXTERNAL: only borders that are not inside another border
                    // ---- follow the outer border (icvFetchContour) ----
                    std::vector<Point> c;
                    const size_t i0 = (size_t)y * step + x;
                    int s_end = 4, s = 4;
                    size_t i1;
                    do {
                        s = (s - 1) & 7;
                        i1 = i0 + deltas[s];
                    } while (img[i1] == 0 && s != s_end);
                    Point pt{x - 1, y - 1};
                    if (s == s_end) {
                        img[i0] = -nbd;           // isolated pixel
                        c.push_back(pt);
                    } else {
                        size_t i3 = i0, i4 = 0;
                        for (;;) {
                            s_end = s;
                            while (s < 15) {
                                i4 = i3 + deltas[++s];
                                if (img[i4] != 0) break;
                            }
                            s &= 7;
                            // east neighbour examined and background: the border leaves to the right here
                            if ((unsigned)(s - 1) < (unsigned)s_end) img[i3] = -nbd;
                            else if (img[i3] == 1) img[i3] = nbd;
                            c.push_back(pt);
                            pt.x += dx[s]; pt.y += dy[s];
                            if (i4 == i0 && i3 == i1) break;
                            i3 = i4;
                            s = (s + 4) & 7;
                        }
                    }
                    contours.push_back(std::move(c));
                    nbd++;
                    p = img[(size_t)y * step + x];   // the start pixel carries its label now
                }
            } else if (p == 0 && prev >= 1 && (prev & -2)) {
                lnbd = prev;                      // a hole starts behind a labelled border pixel
            }
            prev = p;
            if (prev & -2) lnbd = prev;
        }
    }
    return contours;
}

HostImage fillContour(const std::vector<Point>& contour, int width, int height) {
    // pixels enclosed by the contour = everything the background cannot reach from the frame through non-contour
    // pixels with 4-connectivity (the complement of an 8-connected closed chain is 4-connected)
    HostImage out;
    out.width = width; out.height = height; out.channels = 1;
    out.data.assign((size_t)width * height, 255);
    const int step = width + 2;
    std::vector<uint8_t> st((size_t)step * (height + 2), 0);   // 0 unknown, 1 contour, 2 outside
    for (const Point& p : contour) st[(size_t)(p.y + 1) * step + p.x + 1] = 1;
    std::vector<size_t> stack;
    auto push = [&](size_t i) { if (st[i] == 0) { st[i] = 2; stack.push_back(i); } };
    for (int x = 0; x < step; x++) { push((size_t)x); push((size_t)(height + 1) * step + x); }
    for (int y = 0; y < height + 2; y++) { push((size_t)y * step); push((size_t)y * step + step - 1); }
    while (!stack.empty()) {
        const size_t i = stack.back();
        stack.pop_back();
        const int y = (int)(i / step), x = (int)(i % step);
        if (x > 0) push(i - 1);
        if (x < step - 1) push(i + 1);
        if (y > 0) push(i - step);
        if (y < height + 1) push(i + step);
    }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            if (st[(size_t)(y + 1) * step + x + 1] == 2) out.data[(size_t)y * width + x] = 0;
    return out;
}

Rect crop(HostImage& source) {
    if (source.channels != 3 && source.channels != 1) throw std::runtime_error("crop: 8UC3 or 8UC1 image expected");
    const int w = source.width, h = source.height, cn = source.channels;
    // cvtColor(RGB2GRAY) > 0: 15-bit weights R 9798, G 19235, B 3735 on channels 0, 1, 2 (the reference passes a BGR image
    // to COLOR_RGB2GRAY; only "> 0" matters afterwards)
    HostImage mask;
    mask.width = w; mask.height = h; mask.channels = 1;
    mask.data.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        int g = cn == 3 ? (source.data[3 * i] * 9798 + source.data[3 * i + 1] * 19235 + source.data[3 * i + 2] * 3735 + (1 << 14)) >> 15 : source.data[i];
        mask.data[i] = g > 0 ? 255 : 0;
    }
    std::vector<std::vector<Point>> contours = findExternalContours(mask);
    if (contours.empty()) throw std::runtime_error("crop: the image is empty");
    size_t id = 0, maxSize = 0;
    for (size_t i = 0; i < contours.size(); ++i)
        if (contours[i].size() > maxSize) { maxSize = contours[i].size(); id = i; }
    const HostImage contourMask = fillContour(contours[id], w, h);
    std::vector<Point> cSortedX = contours[id], cSortedY = contours[id];
    std::sort(cSortedX.begin(), cSortedX.end(), compareX);
    std::sort(cSortedY.begin(), cSortedY.end(), compareY);
    int minXId = 0, maxXId = (int)cSortedX.size() - 1, minYId = 0, maxYId = (int)cSortedY.size() - 1;
    Rect croppingMask;
    while (minXId < maxXId && minYId < maxYId) {
        const Point mn{cSortedX[minXId].x, cSortedY[minYId].y}, mx{cSortedX[maxXId].x, cSortedY[maxYId].y};
        croppingMask = Rect{mn.x, mn.y, mx.x - mn.x, mx.y - mn.y};
        if (croppingMask.width <= 0 || croppingMask.height <= 0) break;   // cv::Mat::operator() would throw on an empty ROI
        int ocTop = 0, ocBottom = 0, ocLeft = 0, ocRight = 0;
        if (checkInteriorExterior(contourMask, croppingMask, ocTop, ocBottom, ocLeft, ocRight)) break;
        if (ocLeft) ++minXId;
        if (ocRight) --maxXId;
        if (ocTop) ++minYId;
        if (ocBottom) --maxYId;
    }
    if (croppingMask.width <= 0 || croppingMask.height <= 0) throw std::runtime_error("crop: no interior rectangle found");
    HostImage out;
    out.width = croppingMask.width; out.height = croppingMask.height; out.channels = cn;
    out.data.resize((size_t)out.width * out.height * cn);
    for (int y = 0; y < out.height; y++)
        std::memcpy(&out.data[(size_t)y * out.width * cn], &source.data[((size_t)(croppingMask.y + y) * w + croppingMask.x) * cn], (size_t)out.width * cn);
    source = std::move(out);
    return croppingMask;
}

}  // namespace mis
