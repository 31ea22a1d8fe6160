// seam.hip -- DpSeamFinder(COLOR), the reference's default seam finder, host logic of the library:
//   seam_finder = makePtr<detail::DpSeamFinder>(DpSeamFinder::COLOR)      image_stitching/image_stitching.cpp:1056-1057
//   seam_finder->find(images_warped_f, corners, masks_warped)             image_stitching/image_stitching.cpp:1065
// The reference runs it on the CPU inside OpenCV (stitching/src/seam_finders.cpp: DpSeamFinder::find / process /
// findComponents / findEdges / resolveConflicts / getSeamTips / computeCosts / estimateSeam / updateLabelsUsingSeam,
// core's cv::partition, imgproc's floodFill) on the seam-scale images (~0.1 MP each); restated here from the published
// algorithm.  PARITY UNPINNED (OpenCV absent offline).  Steps recalled with less than full confidence are marked [uncertain].
// No kernels: the data is two ~420 x 240 images per pair and the algorithm is sequential (labelling, dynamic programming along
// a seam, flood fills); images and masks are copied to the host, the masks go back to the device.
#include "common.h"
#include "dev_math.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <utility>
#include <vector>

namespace {
static std::atomic<long long> g_seam_ns[8];   // MIS_SEAM_TRACE: setup, components, edges, tips, estimate, update labels, rescan, masks
struct SeamTimer { int i; std::chrono::steady_clock::time_point t0; explicit SeamTimer(int i_) : i(i_), t0(std::chrono::steady_clock::now()) {} ~SeamTimer() { g_seam_ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } };

struct Pt { int x, y; };
inline bool operator==(const Pt& a, const Pt& b) { return a.x == b.x && a.y == b.y; }

template <typename T>
struct Grid {
    int w = 0, h = 0;
    std::vector<T> v;
    void create(int w_, int h_, T fill = T()) { w = w_; h = h_; v.assign((size_t)w_ * h_, fill); }
    T& operator()(int y, int x) { return v[(size_t)y * w + x]; }
    const T& operator()(int y, int x) const { return v[(size_t)y * w + x]; }
};

struct HostImage {   // CV_32FC3 view of a seam-scale image (the reference converts images_warped to CV_32F at :992-994)
    int w = 0, h = 0;
    std::vector<float> px;   // 3 floats per pixel
    const float* at(int y, int x) const { return &px[((size_t)y * w + x) * 3]; }
};

// cv::floodFill(image, seed, newVal) on a 32-bit integer image: 4-connected, the region of pixels equal to the seed's value
void flood_fill(Grid<int>& g, int sx, int sy, int new_val) {
    const int old = g(sy, sx);
    if (old == new_val) return;
    // scanline form: a stack of row spans instead of single pixels (the filled set is the 4-connected region either way)
    struct Span { int y, x0, x1; };
    static thread_local std::vector<Span> stack;
    stack.clear();
    auto fill_row = [&](int y, int x) {       // fills the run of `old` through (x, y), returns it
        int* row = &g(y, 0);
        int a = x, b = x;
        while (a > 0 && row[a - 1] == old) a--;
        while (b + 1 < g.w && row[b + 1] == old) b++;
        for (int i = a; i <= b; i++) row[i] = new_val;
        return Span{y, a, b};
    };
    stack.push_back(fill_row(sy, sx));
    while (!stack.empty()) {
        const Span s = stack.back();
        stack.pop_back();
        for (int dy = -1; dy <= 1; dy += 2) {
            const int y = s.y + dy;
            if (y < 0 || y >= g.h) continue;
            const int* row = &g(y, 0);
            for (int x = s.x0; x <= s.x1; x++)
                if (row[x] == old) { const Span t = fill_row(y, x); stack.push_back(t); x = t.x1; }
        }
    }
}

enum { FIRST = 1, SECOND = 2, INTERS = 4, INTERS_FIRST = INTERS | FIRST, INTERS_SECOND = INTERS | SECOND };

// the per-pair state of DpSeamFinder (seam_finders.hpp: "processing images pair data" / "components data")
struct DpPair {
    Pt union_tl{0, 0}, union_br{0, 0};
    int uw = 0, uh = 0;
    Grid<uint8_t> mask1, mask2, contour1, contour2;
    int ncomps = 0;
    Grid<int> labels;
    std::vector<int> states;
    std::vector<Pt> tls, brs;
    std::vector<std::vector<Pt>> contours;
    std::set<std::pair<int, int>> edges;

    bool on_border(int y, int x, int l) const {
        return (x == 0 || labels(y, x - 1) != l) || (x == uw - 1 || labels(y, x + 1) != l) || (y == 0 || labels(y - 1, x) != l) || (y == uh - 1 || labels(y + 1, x) != l);
    }

    // DpSeamFinder::findComponents
    void find_components() {
        ncomps = 0;
        labels.create(uw, uh);
        states.clear(); tls.clear(); brs.clear(); contours.clear();
        for (int y = 0; y < uh; y++) {
            const uint8_t* a = &mask1(y, 0);
            const uint8_t* b = &mask2(y, 0);
            int* lr = &labels(y, 0);
            for (int x = 0; x < uw; x++) lr[x] = a[x] ? (b[x] ? INT_MAX : INT_MAX - 1) : (b[x] ? INT_MAX - 2 : 0);
        }
        for (int y = 0; y < uh; y++) {
            int* lr = &labels(y, 0);
            const int* up = y > 0 ? &labels(y - 1, 0) : nullptr;
            const int* dn = y < uh - 1 ? &labels(y + 1, 0) : nullptr;
            for (int x = 0; x < uw; x++) {
                if (lr[x] >= INT_MAX - 2) {
                    if (lr[x] == INT_MAX) states.push_back(INTERS);
                    else if (lr[x] == INT_MAX - 1) states.push_back(FIRST);
                    else states.push_back(SECOND);
                    flood_fill(labels, x, y, ++ncomps);
                    tls.push_back(Pt{x, y});
                    brs.push_back(Pt{x + 1, y + 1});
                    contours.push_back(std::vector<Pt>());
                }
                const int l = lr[x];
                if (l) {
                    const int ci = l - 1;
                    tls[ci].x = std::min(tls[ci].x, x); tls[ci].y = std::min(tls[ci].y, y);
                    brs[ci].x = std::max(brs[ci].x, x + 1); brs[ci].y = std::max(brs[ci].y, y + 1);
                    // on_border(y, x, l) on the row pointers
                    if ((x == 0 || lr[x - 1] != l) || (x == uw - 1 || lr[x + 1] != l) || (!up || up[x] != l) || (!dn || dn[x] != l)) contours[ci].push_back(Pt{x, y});
                }
            }
        }
    }

    // DpSeamFinder::findEdges: components that touch (4-neighbourhood) are joined by an edge in both directions
    void find_edges() {
        // (a ncomps x ncomps table instead of an ordered map touched once per contour pixel)
        std::vector<uint8_t> touchm((size_t)ncomps * ncomps, 0);
        for (int ci = 0; ci < ncomps; ci++)
            for (const Pt& p : contours[ci]) {
                const int x = p.x, y = p.y, l = ci + 1;
                auto touch = [&](int yy, int xx) {
                    const int o = labels(yy, xx);
                    if (o && o != l) { touchm[(size_t)ci * ncomps + (o - 1)] = 1; touchm[(size_t)(o - 1) * ncomps + ci] = 1; }
                };
                if (x > 0) touch(y, x - 1);
                if (y > 0) touch(y - 1, x);
                if (x < uw - 1) touch(y, x + 1);
                if (y < uh - 1) touch(y + 1, x);
            }
        edges.clear();
        for (int a = 0; a < ncomps; a++)
            for (int b = 0; b < ncomps; b++)
                if (touchm[(size_t)a * ncomps + b]) edges.insert(std::make_pair(a, b));
    }

    bool has_only_one_neighbor(int comp) const {
        auto b = edges.lower_bound(std::make_pair(comp, INT_MIN)), e = edges.upper_bound(std::make_pair(comp, INT_MAX));
        return ++b == e;
    }

    bool close_to_contour(int y, int x, const Grid<uint8_t>& cm) const {
        const int rad = 2;
        for (int dy = -rad; dy <= rad; dy++) {
            if (y + dy < 0 || y + dy >= uh) continue;
            for (int dx = -rad; dx <= rad; dx++)
                if (x + dx >= 0 && x + dx < uw && cm(y + dy, x + dx)) return true;
        }
        return false;
    }

    bool touches(int y, int x, int l) const {
        return (x > 0 && labels(y, x - 1) == l) || (y > 0 && labels(y - 1, x) == l) || (x < uw - 1 && labels(y, x + 1) == l) || (y < uh - 1 && labels(y + 1, x) == l);
    }

    // DpSeamFinder::getSeamTips: the two ends of the seam between comp1 (an intersection) and comp2
    bool get_seam_tips(int comp1, int comp2, Pt* p1, Pt* p2) const {
        std::vector<Pt> special;
        const int l2 = comp2 + 1;
        for (const Pt& p : contours[comp1])
            if (close_to_contour(p.y, p.x, contour1) && close_to_contour(p.y, p.x, contour2) && touches(p.y, p.x, l2)) special.push_back(p);
        if (special.size() < 2) return false;
        // cv::partition(specialPoints, labels, ClosePoints(10)): equivalence classes of "closer than 10 px", numbered in the
        // order of their first member
        const int n = (int)special.size();
        std::vector<int> parent(n);
        for (int i = 0; i < n; i++) parent[i] = i;
        auto root = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                const int dx = special[i].x - special[j].x, dy = special[i].y - special[j].y;
                if (dx * dx + dy * dy < 10 * 10) { const int a = root(i), b = root(j); if (a != b) parent[b] = a; }
            }
        std::vector<int> cls(n, -1), lab(n);
        int nlabels = 0;
        for (int i = 0; i < n; i++) {
            const int r = root(i);
            if (cls[r] < 0) cls[r] = nlabels++;
            lab[i] = cls[r];
        }
        if (nlabels < 2) return false;
        std::vector<Pt> sum(nlabels, Pt{0, 0});
        std::vector<std::vector<Pt>> points(nlabels);
        for (int i = 0; i < n; i++) { sum[lab[i]].x += special[i].x; sum[lab[i]].y += special[i].y; points[lab[i]].push_back(special[i]); }
        // the two most distant clusters (centres rounded like cvRound: half to even)
        int idx[2] = {-1, -1};
        double max_dist = -DBL_MAX;
        for (int i = 0; i < nlabels - 1; i++)
            for (int j = i + 1; j < nlabels; j++) {
                const double s1 = (double)points[i].size(), s2 = (double)points[j].size();
                const double cx1 = (double)mis_round_d(sum[i].x / s1), cy1 = (double)mis_round_d(sum[i].y / s1);
                const double cx2 = (double)mis_round_d(sum[j].x / s2), cy2 = (double)mis_round_d(sum[j].y / s2);
                const double dist = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
                if (dist > max_dist) { max_dist = dist; idx[0] = i; idx[1] = j; }
            }
        // in each of them the point closest to the centre
        Pt p[2];
        for (int i = 0; i < 2; i++) {
            const std::vector<Pt>& pts = points[idx[i]];
            const double size = (double)pts.size();
            const double cx = (double)mis_round_d(sum[idx[i]].x / size), cy = (double)mis_round_d(sum[idx[i]].y / size);
            size_t closest = pts.size();
            double min_dist = DBL_MAX;
            for (size_t j = 0; j < pts.size(); j++) {
                const double dist = (pts[j].x - cx) * (pts[j].x - cx) + (pts[j].y - cy) * (pts[j].y - cy);
                if (dist < min_dist) { min_dist = dist; closest = j; }
            }
            p[i] = pts[closest];
        }
        *p1 = p[0]; *p2 = p[1];
        return true;
    }

    // diffL2Square3<float>
    static float diff(const HostImage& a, int y1, int x1, const HostImage& b, int y2, int x2) {
        const float* r1 = a.at(y1, x1);
        const float* r2 = b.at(y2, x2);
        const float d0 = r1[0] - r2[0], d1 = r1[1] - r2[1], d2 = r1[2] - r2[2];
        return (d0 * d0 + d1 * d1) + d2 * d2;
    }

    // DpSeamFinder::computeCosts (COLOR): cost of cutting between horizontally / vertically adjacent pixels of the component.
    // [uncertain] OpenCV reads labels_(y, x) for x == roi.br().x / y == roi.br().y, one past the component's box (possibly one
    // past the union): here anything outside the union counts as "not this component" (the bad-region cost).
    void compute_costs(const HostImage& im1, const HostImage& im2, Pt tl1, Pt tl2, int comp, Grid<float>& costV, Grid<float>& costH) const {
        const int l = comp + 1;
        const int rx = tls[comp].x, ry = tls[comp].y, rw = brs[comp].x - rx, rh = brs[comp].y - ry;
        const int dx1 = union_tl.x - tl1.x, dy1 = union_tl.y - tl1.y, dx2 = union_tl.x - tl2.x, dy2 = union_tl.y - tl2.y;
        const float bad = 3.f * 255.f * 255.f;   // normL2(Point3f(255, 255, 255), Point3f(0, 0, 0)): the squared norm
        auto lab = [&](int y, int x) { return (x >= 0 && x < uw && y >= 0 && y < uh) ? labels(y, x) : 0; };
        costV.create(rw + 1, rh);
        for (int y = ry; y < ry + rh; y++)
            for (int x = rx; x < rx + rw + 1; x++) {
                if (lab(y, x) == l && x > 0 && lab(y, x - 1) == l)
                    costV(y - ry, x - rx) = (diff(im1, y + dy1, x + dx1 - 1, im2, y + dy2, x + dx2) + diff(im1, y + dy1, x + dx1, im2, y + dy2, x + dx2 - 1)) / 2;
                else
                    costV(y - ry, x - rx) = bad;
            }
        costH.create(rw, rh + 1);
        for (int y = ry; y < ry + rh + 1; y++)
            for (int x = rx; x < rx + rw; x++) {
                if (lab(y, x) == l && y > 0 && lab(y - 1, x) == l)
                    costH(y - ry, x - rx) = (diff(im1, y + dy1 - 1, x + dx1, im2, y + dy2, x + dx2) + diff(im1, y + dy1, x + dx1, im2, y + dy2 - 1, x + dx2)) / 2;
                else
                    costH(y - ry, x - rx) = bad;
            }
    }

    // DpSeamFinder::estimateSeam: dynamic programming from p1 to p2 along the longer axis of (p2 - p1)
    bool estimate_seam(const HostImage& im1, const HostImage& im2, Pt tl1, Pt tl2, int comp, Pt p1, Pt p2, std::vector<Pt>& seam, bool* is_horizontal) const {
        Grid<float> costV, costH;
        compute_costs(im1, im2, tl1, tl2, comp, costV, costH);
        const int rx = tls[comp].x, ry = tls[comp].y, rw = brs[comp].x - rx, rh = brs[comp].y - ry;
        Pt src{p1.x - rx, p1.y - ry}, dst{p2.x - rx, p2.y - ry};
        const int l = comp + 1;
        bool swapped = false;
        const bool horiz = std::abs(dst.x - src.x) > std::abs(dst.y - src.y);
        *is_horizontal = horiz;
        if (horiz) { if (src.x > dst.x) { std::swap(src, dst); swapped = true; } }
        else if (src.y > dst.y) { std::swap(src, dst); swapped = true; }
        Grid<uint8_t> control, reachable;
        Grid<float> cost;
        control.create(rw, rh, 0); reachable.create(rw, rh, 0); cost.create(rw, rh, 0.f);
        reachable(src.y, src.x) = 1;
        cost(src.y, src.x) = 0.f;
        std::pair<float, int> steps[3];
        if (horiz) {
            for (int x = src.x + 1; x <= dst.x; x++)
                for (int y = 0; y < rh; y++) {
                    int nsteps = 0;   // the seam follows the upper side of pixels
                    if (labels(y + ry, x + rx) == l) {
                        if (reachable(y, x - 1)) steps[nsteps++] = std::make_pair(cost(y, x - 1) + costH(y, x - 1), 1);
                        if (y > 0 && reachable(y - 1, x - 1)) steps[nsteps++] = std::make_pair(cost(y - 1, x - 1) + costH(y - 1, x - 1) + costV(y - 1, x), 2);
                        if (y < rh - 1 && reachable(y + 1, x - 1)) steps[nsteps++] = std::make_pair(cost(y + 1, x - 1) + costH(y + 1, x - 1) + costV(y, x), 3);
                    }
                    if (nsteps) {
                        const std::pair<float, int> opt = *std::min_element(steps, steps + nsteps);
                        cost(y, x) = opt.first; control(y, x) = (uint8_t)opt.second; reachable(y, x) = 255;
                    }
                }
        } else {
            for (int y = src.y + 1; y <= dst.y; y++)
                for (int x = 0; x < rw; x++) {
                    int nsteps = 0;   // the seam follows the left side of pixels
                    if (labels(y + ry, x + rx) == l) {
                        if (reachable(y - 1, x)) steps[nsteps++] = std::make_pair(cost(y - 1, x) + costV(y - 1, x), 1);
                        if (x > 0 && reachable(y - 1, x - 1)) steps[nsteps++] = std::make_pair(cost(y - 1, x - 1) + costV(y - 1, x - 1) + costH(y, x - 1), 2);
                        if (x < rw - 1 && reachable(y - 1, x + 1)) steps[nsteps++] = std::make_pair(cost(y - 1, x + 1) + costV(y - 1, x + 1) + costH(y, x), 3);
                    }
                    if (nsteps) {
                        const std::pair<float, int> opt = *std::min_element(steps, steps + nsteps);
                        cost(y, x) = opt.first; control(y, x) = (uint8_t)opt.second; reachable(y, x) = 255;
                    }
                }
        }
        if (!reachable(dst.y, dst.x)) return false;
        Pt p = dst;
        seam.clear();
        seam.push_back(Pt{p.x + rx, p.y + ry});
        if (horiz) {
            while (p.x != src.x) {
                if (control(p.y, p.x) == 2) { p.y--; p.x--; }
                else if (control(p.y, p.x) == 3) { p.y++; p.x--; }
                else p.x--;
                seam.push_back(Pt{p.x + rx, p.y + ry});
            }
        } else {
            while (p.y != src.y) {
                if (control(p.y, p.x) == 2) { p.x--; p.y--; }
                else if (control(p.y, p.x) == 3) { p.x++; p.y--; }
                else p.y--;
                seam.push_back(Pt{p.x + rx, p.y + ry});
            }
        }
        if (!swapped) std::reverse(seam.begin(), seam.end());
        return seam.front() == p1 && seam.back() == p2;   // CV_Assert in the reference
    }

    // DpSeamFinder::updateLabelsUsingSeam: the parts of comp1 that the seam cuts off towards comp2 take comp2's label
    void update_labels_using_seam(int comp1, int comp2, const std::vector<Pt>& seam, bool horizontal) {
        const int ox = tls[comp1].x, oy = tls[comp1].y;
        Grid<int> mask;
        mask.create(brs[comp1].x - ox, brs[comp1].y - oy, 0);
        for (const Pt& p : contours[comp1]) mask(p.y - oy, p.x - ox) = 255;
        for (const Pt& p : seam) mask(p.y - oy, p.x - ox) = 255;
        const int l1 = comp1 + 1, l2 = comp2 + 1;
        int nc = 0;
        for (int y = 0; y < mask.h; y++)
            for (int x = 0; x < mask.w; x++)
                if (!mask(y, x) && labels(y + oy, x + ox) == l1) flood_fill(mask, x, y, ++nc);
        for (const Pt& p : contours[comp1]) {
            const int x = p.x - ox, y = p.y - oy;
            bool ok = false;
            static const int dx[8] = {-1, +1, 0, 0, -1, +1, -1, +1}, dy[8] = {0, 0, -1, +1, -1, -1, +1, +1};
            for (int j = 0; j < 8; j++) {
                const int c = x + dx[j], r = y + dy[j];
                if (c >= 0 && c < mask.w && r >= 0 && r < mask.h && mask(r, c) && mask(r, c) != 255) { ok = true; mask(y, x) = mask(r, c); }
            }
            if (!ok) mask(y, x) = 0;
        }
        for (const Pt& p : seam) {
            const int x = p.x - ox, y = p.y - oy;
            if (horizontal) {
                if (y < mask.h - 1 && mask(y + 1, x) && mask(y + 1, x) != 255) mask(y, x) = mask(y + 1, x);
                else mask(y, x) = 0;
            } else {
                if (x < mask.w - 1 && mask(y, x + 1) && mask(y, x + 1) != 255) mask(y, x) = mask(y, x + 1);
                else mask(y, x) = 0;
            }
        }
        // new components connected with the second component / with components other than the two at hand
        std::map<int, int> connect2, connect_other;
        for (int i = 1; i <= nc; i++) { connect2[i] = 0; connect_other[i] = 0; }
        for (const Pt& p : contours[comp1]) {
            const int x = p.x, y = p.y;
            if (touches(y, x, l2)) connect2[mask(y - oy, x - ox)]++;
            auto other = [&](int yy, int xx) { const int o = labels(yy, xx); return o != l1 && o != l2; };
            if ((x > 0 && other(y, x - 1)) || (y > 0 && other(y - 1, x)) || (x < uw - 1 && other(y, x + 1)) || (y < uh - 1 && other(y + 1, x)))
                connect_other[mask(y - oy, x - ox)]++;
        }
        std::vector<int> is_adj((size_t)std::max(nc, 255) + 1, 0);   // (keys 0 and 255 can enter the maps through mask values)
        const double len = (double)contours[comp1].size();
        for (const auto& e : connect2) {
            int res = 0;
            if (e.second / len > 0.05) {
                auto sub = connect_other.find(e.first);
                if (sub != connect_other.end() && (sub->second / len < 0.1)) res = 1;
            }
            is_adj[e.first] = res;
        }
        for (int y = 0; y < mask.h; y++)
            for (int x = 0; x < mask.w; x++)
                if (mask(y, x) && is_adj[mask(y, x)]) labels(y + oy, x + ox) = l2;
    }

    // DpSeamFinder::resolveConflicts
    void resolve_conflicts(const HostImage& im1, const HostImage& im2, Pt tl1, Pt tl2, Grid<uint8_t>& m1, Grid<uint8_t>& m2) {
        bool has_conflict = true;
        while (has_conflict) {
            int c1 = 0, c2 = 0;
            has_conflict = false;
            for (const auto& e : edges) {
                c1 = e.first; c2 = e.second;
                if ((states[c1] & INTERS) && (states[c1] & (~INTERS)) != states[c2]) { has_conflict = true; break; }
            }
            if (!has_conflict) break;
            const int l1 = c1 + 1, l2 = c2 + 1;
            if (has_only_one_neighbor(c1)) {
                for (int y = tls[c1].y; y < brs[c1].y; y++)
                    for (int x = tls[c1].x; x < brs[c1].x; x++)
                        if (labels(y, x) == l1) labels(y, x) = l2;
                states[c1] = states[c2] == FIRST ? INTERS_SECOND : INTERS_FIRST;
            } else {
                Pt p1, p2;
                bool tips;
                { SeamTimer tm(3); tips = get_seam_tips(c1, c2, &p1, &p2); }
                if (tips) {
                    std::vector<Pt> seam;
                    bool horizontal = false;
                    bool est;
                    { SeamTimer tm(4); est = estimate_seam(im1, im2, tl1, tl2, c1, p1, p2, seam, &horizontal); }
                    if (est) { SeamTimer tm(5); update_labels_using_seam(c1, c2, seam, horizontal); }
                }
                states[c1] = states[c2] == FIRST ? INTERS_SECOND : INTERS_FIRST;
            }
            // box and contour of both components again (scanned inside their previous boxes, as the reference does)
            const int c[2] = {c1, c2}, l[2] = {l1, l2};
            SeamTimer tm_rescan(6);
            for (int i = 0; i < 2; i++) {
                const int x0 = tls[c[i]].x, x1 = brs[c[i]].x, y0 = tls[c[i]].y, y1 = brs[c[i]].y;
                tls[c[i]] = Pt{INT_MAX, INT_MAX};
                brs[c[i]] = Pt{INT_MIN, INT_MIN};
                contours[c[i]].clear();
                for (int y = y0; y < y1; y++) {
                    const int* lr = &labels(y, 0);
                    const int* up = y > 0 ? &labels(y - 1, 0) : nullptr;
                    const int* dn = y < uh - 1 ? &labels(y + 1, 0) : nullptr;
                    const int li = l[i];
                    for (int x = x0; x < x1; x++)
                        if (lr[x] == li) {
                            tls[c[i]].x = std::min(tls[c[i]].x, x); tls[c[i]].y = std::min(tls[c[i]].y, y);
                            brs[c[i]].x = std::max(brs[c[i]].x, x + 1); brs[c[i]].y = std::max(brs[c[i]].y, y + 1);
                            if ((x == 0 || lr[x - 1] != li) || (x == uw - 1 || lr[x + 1] != li) || (!up || up[x] != li) || (!dn || dn[x] != li)) contours[c[i]].push_back(Pt{x, y});
                        }
                }
            }
            // [uncertain] the resolved edge leaves the graph in both directions
            edges.erase(std::make_pair(c1, c2));
            edges.erase(std::make_pair(c2, c1));
        }
        // update the masks: a pixel of a component that went to the first image leaves the second image's mask, and vice versa
        const int dx1 = union_tl.x - tl1.x, dy1 = union_tl.y - tl1.y, dx2 = union_tl.x - tl2.x, dy2 = union_tl.y - tl2.y;
        for (int y = 0; y < m2.h; y++)
            for (int x = 0; x < m2.w; x++) {
                const int l = labels(y - dy2, x - dx2);
                if (l > 0 && (states[l - 1] & FIRST) && m1(y - dy2 + dy1, x - dx2 + dx1)) m2(y, x) = 0;
            }
        for (int y = 0; y < m1.h; y++)
            for (int x = 0; x < m1.w; x++) {
                const int l = labels(y - dy1, x - dx1);
                if (l > 0 && (states[l - 1] & SECOND) && m2(y - dy1 + dy2, x - dx1 + dx2)) m1(y, x) = 0;
            }
    }

    // DpSeamFinder::process
    void process(const HostImage& im1, const HostImage& im2, Pt tl1, Pt tl2, Grid<uint8_t>& m1, Grid<uint8_t>& m2) {
        const Pt itl{std::max(tl1.x, tl2.x), std::max(tl1.y, tl2.y)};
        const Pt ibr{std::min(tl1.x + im1.w, tl2.x + im2.w), std::min(tl1.y + im1.h, tl2.y + im2.h)};
        if (itl.x >= ibr.x || itl.y >= ibr.y) return;   // no overlap: no conflicts
        SeamTimer tm_setup(0);
        union_tl = Pt{std::min(tl1.x, tl2.x), std::min(tl1.y, tl2.y)};
        union_br = Pt{std::max(tl1.x + im1.w, tl2.x + im2.w), std::max(tl1.y + im1.h, tl2.y + im2.h)};
        uw = union_br.x - union_tl.x; uh = union_br.y - union_tl.y;
        mask1.create(uw, uh, 0); mask2.create(uw, uh, 0);
        for (int y = 0; y < m1.h; y++) memcpy(&mask1(y + tl1.y - union_tl.y, tl1.x - union_tl.x), &m1(y, 0), (size_t)m1.w);
        for (int y = 0; y < m2.h; y++) memcpy(&mask2(y + tl2.y - union_tl.y, tl2.x - union_tl.x), &m2(y, 0), (size_t)m2.w);
        contour1.create(uw, uh, 0); contour2.create(uw, uh, 0);
        auto contour_of = [&](const Grid<uint8_t>& m, Grid<uint8_t>& c) {
            for (int y = 0; y < uh; y++) {
                const uint8_t* r = &m(y, 0);
                const uint8_t* up = y > 0 ? &m(y - 1, 0) : nullptr;
                const uint8_t* dn = y < uh - 1 ? &m(y + 1, 0) : nullptr;
                uint8_t* o = &c(y, 0);
                for (int x = 0; x < uw; x++)
                    if (r[x] && ((x == 0 || !r[x - 1]) || (x == uw - 1 || !r[x + 1]) || (!up || !up[x]) || (!dn || !dn[x]))) o[x] = 255;
            }
        };
        contour_of(mask1, contour1);
        contour_of(mask2, contour2);
        { SeamTimer tm(1); find_components(); }
        { SeamTimer tm(2); find_edges(); }
        resolve_conflicts(im1, im2, tl1, tl2, m1, m2);
    }
};

}  // namespace

// DpSeamFinder::find: every pair of images, the most distant centres first.
// [uncertain] the reference orders the pairs with std::sort (unstable) + std::reverse; pairs at equal distance are ordered here as a
// stable sort + reverse leaves them.
extern "C" int mis_seam_dp(MisContext* ctx, const MisPoint* corners, const MisImage* images, MisImage* masks, int n, int cost_func) {
    if (!ctx) return MIS_E_INVALID;
    MIS_CHECK(ctx, corners && images && masks && n >= 0, MIS_E_INVALID, "null argument");
    MIS_CHECK(ctx, cost_func == MIS_SEAM_DP_COLOR, MIS_E_UNSUPPORTED, "DpSeamFinder: only the COLOR cost (the reference's \"dp_color\") is implemented");
    if (n == 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    if (getenv("MIS_SEAM_TRACE")) { const auto t_e = std::chrono::steady_clock::now(); hipStreamSynchronize(ctx->stream); fprintf(stderr, "dp seams: entry sync %.2f ms\n", std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_e).count() / 1e3); }
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<HostImage> im((size_t)n);
    std::vector<Grid<uint8_t>> mk((size_t)n);
    // device images and masks come over in one go: every copy into a pinned staging buffer, ONE synchronisation (a pageable
    // destination made each of the 3 n small 2-D copies a synchronous 0.8 ms affair: 39 of the finder's 52 ms)
    size_t stage_total = 0;
    std::vector<size_t> ioff((size_t)n), moff((size_t)n);
    for (int i = 0; i < n; i++) {
        const MisImage& I = images[i];
        const MisImage& M = masks[i];
        MIS_CHECK(ctx, I.data && M.data && I.dtype == MIS_U8 && I.channels == 3 && M.dtype == MIS_U8 && M.channels == 1 && I.width == M.width && I.height == M.height &&
                           I.width > 0 && I.height > 0, MIS_E_INVALID, "image %d: need an 8UC3 image and an 8U mask of the same size", i);
        ioff[i] = stage_total; stage_total += mis_align_up((size_t)I.width * I.height * 3, 256);
        moff[i] = stage_total; stage_total += mis_align_up((size_t)M.width * M.height, 256);
    }
    uint8_t* stage = nullptr;
    { void* sp = nullptr; int rc = mis_host_stage(ctx, stage_total, &sp); if (rc != MIS_OK) return rc; stage = (uint8_t*)sp; }
    // device inputs are gathered into one device block first (device-to-device copies are cheap launches) and cross the bus in ONE
    // copy: 3 n separate device-to-host copies of ~0.1 MB took 1 ms each on the stream of a second context, 31 of the finder's 52 ms
    bool any_dev = false;
    for (int i = 0; i < n; i++) any_dev |= images[i].mem == MIS_MEM_DEVICE || masks[i].mem == MIS_MEM_DEVICE;
    void* dblk = nullptr; size_t dgot = 0;
    if (any_dev) { int rc = mis_pool_alloc(ctx, stage_total, &dblk, &dgot); if (rc != MIS_OK) return rc; }
    uint8_t* dstage = (uint8_t*)dblk;
    // every HIP error below returns through here: the device block goes back to the pool
#define SEAM_HIP(call)                                                                                                          \
    do {                                                                                                                        \
        hipError_t e_ = (call);                                                                                                 \
        if (e_ != hipSuccess) {                                                                                                 \
            if (dblk) { hipStreamSynchronize(ctx->stream); mis_pool_free(ctx, dblk, dgot); }                                     \
            return mis_set_error(ctx, MIS_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
        }                                                                                                                       \
    } while (0)
    for (int i = 0; i < n; i++) {
        const MisImage& I = images[i];
        const MisImage& M = masks[i];
        if (I.mem == MIS_MEM_DEVICE) SEAM_HIP(hipMemcpy2DAsync(dstage + ioff[i], (size_t)I.width * 3, I.data, I.stride, (size_t)I.width * 3, I.height, hipMemcpyDeviceToDevice, ctx->stream));
        if (M.mem == MIS_MEM_DEVICE) SEAM_HIP(hipMemcpy2DAsync(dstage + moff[i], M.width, M.data, M.stride, M.width, M.height, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (any_dev) SEAM_HIP(hipMemcpyAsync(stage, dstage, stage_total, hipMemcpyDeviceToHost, ctx->stream));
    const auto t_issued = std::chrono::steady_clock::now();
    SEAM_HIP(hipStreamSynchronize(ctx->stream));
    const auto t_synced = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) {   // host inputs go into their slots of the staging buffer directly
        const MisImage& I = images[i];
        const MisImage& M = masks[i];
        if (I.mem != MIS_MEM_DEVICE) for (int y = 0; y < I.height; y++) memcpy(stage + ioff[i] + (size_t)y * I.width * 3, (const uint8_t*)I.data + (size_t)y * I.stride, (size_t)I.width * 3);
        if (M.mem != MIS_MEM_DEVICE) for (int y = 0; y < M.height; y++) memcpy(stage + moff[i] + (size_t)y * M.width, (const uint8_t*)M.data + (size_t)y * M.stride, M.width);
    }
    for (int i = 0; i < n; i++) {
        const MisImage& I = images[i];
        im[i].w = I.width; im[i].h = I.height;
        const size_t np3 = (size_t)I.width * I.height * 3;
        im[i].px.resize(np3);
        const uint8_t* sp = stage + ioff[i];
        for (size_t k = 0; k < np3; k++) im[i].px[k] = (float)sp[k];     // convertTo(CV_32F), image_stitching.cpp:992-994
        mk[i].w = I.width; mk[i].h = I.height;
        mk[i].v.assign(stage + moff[i], stage + moff[i] + (size_t)I.width * I.height);
    }
    const auto t_loaded = std::chrono::steady_clock::now();
    struct PairD { int d, i, j; };
    std::vector<PairD> pairs;
    for (int i = 0; i + 1 < n; i++)
        for (int j = i + 1; j < n; j++) {
            const int c1x = corners[i].x + im[i].w / 2, c1y = corners[i].y + im[i].h / 2, c2x = corners[j].x + im[j].w / 2, c2y = corners[j].y + im[j].h / 2;
            pairs.push_back(PairD{(c1x - c2x) * (c1x - c2x) + (c1y - c2y) * (c1y - c2y), i, j});
        }
    std::stable_sort(pairs.begin(), pairs.end(), [](const PairD& a, const PairD& b) { return a.d < b.d; });
    std::reverse(pairs.begin(), pairs.end());
    // A pair reads its two images and reads / writes its two masks, nothing else: pairs without a common image commute.  The
    // pairs keep the reference's order wherever it matters -- a pair waits for every earlier pair that shares an image with it --
    // and otherwise run side by side on host threads (level = 1 + the highest level among those earlier pairs).  Pairs whose
    // images do not overlap change nothing and are dropped first.
    std::vector<PairD> work;
    for (const PairD& p : pairs) {
        const int x0 = std::max(corners[p.i].x, corners[p.j].x), y0 = std::max(corners[p.i].y, corners[p.j].y);
        const int x1 = std::min(corners[p.i].x + im[p.i].w, corners[p.j].x + im[p.j].w), y1 = std::min(corners[p.i].y + im[p.i].h, corners[p.j].y + im[p.j].h);
        if (x0 < x1 && y0 < y1) work.push_back(p);
    }
    std::vector<int> level(work.size(), 0), last((size_t)n, 0);
    int nlevels = 0;
    for (size_t k = 0; k < work.size(); k++) {
        level[k] = std::max(last[work[k].i], last[work[k].j]) + 1;
        last[work[k].i] = last[work[k].j] = level[k];
        nlevels = std::max(nlevels, level[k]);
    }
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto run_pair = [&](const PairD& p) {
        static thread_local DpPair dp;      // its grids keep their capacity from pair to pair
        dp.process(im[p.i], im[p.j], Pt{corners[p.i].x, corners[p.i].y}, Pt{corners[p.j].x, corners[p.j].y}, mk[p.i], mk[p.j]);
    };
    for (int lv = 1; lv <= nlevels; lv++) {
        std::vector<size_t> ids;
        for (size_t k = 0; k < work.size(); k++) if (level[k] == lv) ids.push_back(k);
        if (ids.size() == 1 || hw == 1) { for (size_t k : ids) run_pair(work[k]); continue; }
        std::atomic<size_t> next{0};
        auto worker = [&]() { for (size_t q = next++; q < ids.size(); q = next++) run_pair(work[ids[q]]); };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < std::min<size_t>(hw, ids.size()); t++) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
    }
    const auto t_solved = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) {
        const MisImage& M = masks[i];
        if (M.mem == MIS_MEM_DEVICE) memcpy(stage + moff[i], mk[i].v.data(), (size_t)M.width * M.height);
        else for (int y = 0; y < M.height; y++) memcpy((uint8_t*)M.data + (size_t)y * M.stride, mk[i].v.data() + (size_t)y * M.width, M.width);
    }
    if (any_dev) {
        SEAM_HIP(hipMemcpyAsync(dstage, stage, stage_total, hipMemcpyHostToDevice, ctx->stream));   // (the image slots travel back unused: one copy)
        for (int i = 0; i < n; i++) {
            const MisImage& M = masks[i];
            if (M.mem == MIS_MEM_DEVICE) SEAM_HIP(hipMemcpy2DAsync(M.data, M.stride, dstage + moff[i], M.width, M.width, M.height, hipMemcpyDeviceToDevice, ctx->stream));
        }
        mis_pool_free(ctx, dblk, dgot);
        dblk = nullptr;
    }
#undef SEAM_HIP
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (getenv("MIS_SEAM_TRACE")) {
        auto ms = [](auto a, auto b) { return std::chrono::duration_cast<std::chrono::microseconds>(b - a).count() / 1e3; };
        fprintf(stderr, "dp seams wall: load %.2f ms (issue %.2f, sync %.2f, convert %.2f), pairs %.2f ms, store %.2f ms\n", ms(t_begin, t_loaded), ms(t_begin, t_issued), ms(t_issued, t_synced), ms(t_synced, t_loaded), ms(t_loaded, t_solved), ms(t_solved, std::chrono::steady_clock::now()));
        fprintf(stderr, "dp seams: %zu overlapping pairs in %d levels; ms (summed over threads): setup+all %.2f components %.2f edges %.2f tips %.2f estimate %.2f labels %.2f rescan %.2f\n", work.size(), nlevels,
                g_seam_ns[0] / 1e6, g_seam_ns[1] / 1e6, g_seam_ns[2] / 1e6, g_seam_ns[3] / 1e6, g_seam_ns[4] / 1e6, g_seam_ns[5] / 1e6, g_seam_ns[6] / 1e6);
        for (auto& v : g_seam_ns) v = 0;
    }
    return MIS_OK;
}
