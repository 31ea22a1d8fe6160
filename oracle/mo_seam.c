/*
 * mo_seam.c -- ORACLE (test infrastructure): DpSeamFinder(COLOR), the reference's default seam finder.
 * Reference call sites: image_stitching/image_stitching.cpp:1056-1057 (makePtr<detail::DpSeamFinder>(DpSeamFinder::COLOR)),
 * :1065 (seam_finder->find(images_warped_f, corners, masks_warped)), :992-994 (the CV_32F conversion of the warped images).
 * OpenCV source restated: stitching/src/seam_finders.cpp (DpSeamFinder::find, process, findComponents, findEdges,
 * resolveConflicts, hasOnlyOneNeighbor, closeToContour, getSeamTips, computeCosts, estimateSeam, updateLabelsUsingSeam),
 * core/include/opencv2/core/operations.hpp (cv::partition), imgproc floodFill (4-connected, exact value).
 * PARITY UNPINNED (recalled; OpenCV is absent offline).  Steps recalled with less than full confidence carry [uncertain].
 * Plain C with flat arrays: the edge set is an ncomps x ncomps matrix scanned row-major (= the order of std::set<pair>), the
 * std::map counters of updateLabelsUsingSeam are arrays indexed by the key.  Never linked into the product.
 */
#include "mo_seam.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { S_FIRST = 1, S_SECOND = 2, S_INTERS = 4 };

typedef struct { int x, y; } P2;
typedef struct { P2* p; int n, cap; } PVec;

static void pv_push(PVec* v, int x, int y) {
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 64; v->p = (P2*)realloc(v->p, sizeof(P2) * (size_t)v->cap); }
    v->p[v->n].x = x; v->p[v->n].y = y; v->n++;
}

typedef struct {
    int tlx, tly, uw, uh;          /* union rectangle */
    uint8_t *mask1, *mask2, *cont1, *cont2;
    int* labels;
    int ncomps, cap;
    int* states;
    P2 *tls, *brs;
    PVec* contours;
    uint8_t* edges;                /* ncomps x ncomps */
} Dp;

#define LAB(d, y, x) ((d)->labels[(size_t)(y) * (d)->uw + (x)])

/* floodFill on an int image of width w, height h: 4-connected region of the seed's value */
static void flood(int* g, int w, int h, int sx, int sy, int nv) {
    const int old = g[(size_t)sy * w + sx];
    if (old == nv) return;
    int cap = 1024, n = 0;
    P2* st = (P2*)malloc(sizeof(P2) * (size_t)cap);
    st[n].x = sx; st[n].y = sy; n++;
    g[(size_t)sy * w + sx] = nv;
    while (n) {
        const P2 p = st[--n];
        static const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
        for (int k = 0; k < 4; k++) {
            const int x = p.x + dx[k], y = p.y + dy[k];
            if (x < 0 || x >= w || y < 0 || y >= h || g[(size_t)y * w + x] != old) continue;
            g[(size_t)y * w + x] = nv;
            if (n == cap) { cap *= 2; st = (P2*)realloc(st, sizeof(P2) * (size_t)cap); }
            st[n].x = x; st[n].y = y; n++;
        }
    }
    free(st);
}

static int on_border(const Dp* d, int y, int x, int l) {
    return (x == 0 || LAB(d, y, x - 1) != l) || (x == d->uw - 1 || LAB(d, y, x + 1) != l) || (y == 0 || LAB(d, y - 1, x) != l) || (y == d->uh - 1 || LAB(d, y + 1, x) != l);
}

static void add_comp(Dp* d, int state, int x, int y) {
    if (d->ncomps == d->cap) {
        d->cap = d->cap ? 2 * d->cap : 16;
        d->states = (int*)realloc(d->states, sizeof(int) * (size_t)d->cap);
        d->tls = (P2*)realloc(d->tls, sizeof(P2) * (size_t)d->cap);
        d->brs = (P2*)realloc(d->brs, sizeof(P2) * (size_t)d->cap);
        d->contours = (PVec*)realloc(d->contours, sizeof(PVec) * (size_t)d->cap);
    }
    const int c = d->ncomps++;
    d->states[c] = state;
    d->tls[c].x = x; d->tls[c].y = y; d->brs[c].x = x + 1; d->brs[c].y = y + 1;
    d->contours[c].p = NULL; d->contours[c].n = d->contours[c].cap = 0;
}

/* DpSeamFinder::findComponents */
static void find_components(Dp* d) {
    const int w = d->uw, h = d->uh;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t o = (size_t)y * w + x;
            d->labels[o] = (d->mask1[o] && d->mask2[o]) ? INT_MAX : d->mask1[o] ? INT_MAX - 1 : d->mask2[o] ? INT_MAX - 2 : 0;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int l = LAB(d, y, x);
            if (l >= INT_MAX - 2) {
                add_comp(d, l == INT_MAX ? S_INTERS : l == INT_MAX - 1 ? S_FIRST : S_SECOND, x, y);
                flood(d->labels, w, h, x, y, d->ncomps);
            }
            l = LAB(d, y, x);
            if (l) {
                const int c = l - 1;
                if (x < d->tls[c].x) d->tls[c].x = x;
                if (y < d->tls[c].y) d->tls[c].y = y;
                if (x + 1 > d->brs[c].x) d->brs[c].x = x + 1;
                if (y + 1 > d->brs[c].y) d->brs[c].y = y + 1;
                if (on_border(d, y, x, l)) pv_push(&d->contours[c], x, y);
            }
        }
}

/* DpSeamFinder::findEdges */
static void find_edges(Dp* d) {
    const int n = d->ncomps;
    d->edges = (uint8_t*)calloc((size_t)n * n + 1, 1);
    for (int c = 0; c < n; c++)
        for (int i = 0; i < d->contours[c].n; i++) {
            const int x = d->contours[c].p[i].x, y = d->contours[c].p[i].y, l = c + 1;
            int o;
            if (x > 0 && (o = LAB(d, y, x - 1)) && o != l) { d->edges[(size_t)c * n + o - 1] = 1; d->edges[(size_t)(o - 1) * n + c] = 1; }
            if (y > 0 && (o = LAB(d, y - 1, x)) && o != l) { d->edges[(size_t)c * n + o - 1] = 1; d->edges[(size_t)(o - 1) * n + c] = 1; }
            if (x < d->uw - 1 && (o = LAB(d, y, x + 1)) && o != l) { d->edges[(size_t)c * n + o - 1] = 1; d->edges[(size_t)(o - 1) * n + c] = 1; }
            if (y < d->uh - 1 && (o = LAB(d, y + 1, x)) && o != l) { d->edges[(size_t)c * n + o - 1] = 1; d->edges[(size_t)(o - 1) * n + c] = 1; }
        }
}

static int close_to(const Dp* d, int y, int x, const uint8_t* cm) {
    for (int dy = -2; dy <= 2; dy++) {
        if (y + dy < 0 || y + dy >= d->uh) continue;
        for (int dx = -2; dx <= 2; dx++)
            if (x + dx >= 0 && x + dx < d->uw && cm[(size_t)(y + dy) * d->uw + x + dx]) return 1;
    }
    return 0;
}

static int touches(const Dp* d, int y, int x, int l) {
    return (x > 0 && LAB(d, y, x - 1) == l) || (y > 0 && LAB(d, y - 1, x) == l) || (x < d->uw - 1 && LAB(d, y, x + 1) == l) || (y < d->uh - 1 && LAB(d, y + 1, x) == l);
}

static double cv_round_d(double v) { return (double)lrint(v); }   /* cvRound: half to even */

/* DpSeamFinder::getSeamTips */
static int seam_tips(const Dp* d, int c1, int c2, P2* p1, P2* p2) {
    const int l2 = c2 + 1;
    PVec sp = {NULL, 0, 0};
    for (int i = 0; i < d->contours[c1].n; i++) {
        const int x = d->contours[c1].p[i].x, y = d->contours[c1].p[i].y;
        if (close_to(d, y, x, d->cont1) && close_to(d, y, x, d->cont2) && touches(d, y, x, l2)) pv_push(&sp, x, y);
    }
    if (sp.n < 2) { free(sp.p); return 0; }
    /* cv::partition with ClosePoints(10): classes of the relation "squared distance < 100", labelled by first member */
    const int n = sp.n;
    int* par = (int*)malloc(sizeof(int) * (size_t)n * 3);
    int *cls = par + n, *lab = par + 2 * n;
    for (int i = 0; i < n; i++) { par[i] = i; cls[i] = -1; }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) {
            const int dx = sp.p[i].x - sp.p[j].x, dy = sp.p[i].y - sp.p[j].y;
            if (dx * dx + dy * dy >= 100) continue;
            int a = i, b = j;
            while (par[a] != a) a = par[a];
            while (par[b] != b) b = par[b];
            if (a != b) par[b] = a;
        }
    int nl = 0;
    for (int i = 0; i < n; i++) {
        int r = i;
        while (par[r] != r) r = par[r];
        if (cls[r] < 0) cls[r] = nl++;
        lab[i] = cls[r];
    }
    int ok = 0;
    if (nl >= 2) {
        long* sx = (long*)calloc((size_t)nl * 3, sizeof(long));
        long *sy = sx + nl, *cnt = sx + 2 * nl;
        for (int i = 0; i < n; i++) { sx[lab[i]] += sp.p[i].x; sy[lab[i]] += sp.p[i].y; cnt[lab[i]]++; }
        int idx[2] = {-1, -1};
        double best = -DBL_MAX;
        for (int i = 0; i < nl - 1; i++)
            for (int j = i + 1; j < nl; j++) {
                const double cx1 = cv_round_d((int)sx[i] / (double)cnt[i]), cy1 = cv_round_d((int)sy[i] / (double)cnt[i]);
                const double cx2 = cv_round_d((int)sx[j] / (double)cnt[j]), cy2 = cv_round_d((int)sy[j] / (double)cnt[j]);
                const double dist = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
                if (dist > best) { best = dist; idx[0] = i; idx[1] = j; }
            }
        P2 out[2];
        for (int k = 0; k < 2; k++) {
            const double cx = cv_round_d((int)sx[idx[k]] / (double)cnt[idx[k]]), cy = cv_round_d((int)sy[idx[k]] / (double)cnt[idx[k]]);
            double md = DBL_MAX;
            out[k].x = out[k].y = 0;
            for (int i = 0; i < n; i++) {        /* members of the class in their original order */
                if (lab[i] != idx[k]) continue;
                const double dist = (sp.p[i].x - cx) * (sp.p[i].x - cx) + (sp.p[i].y - cy) * (sp.p[i].y - cy);
                if (dist < md) { md = dist; out[k] = sp.p[i]; }
            }
        }
        *p1 = out[0]; *p2 = out[1];
        free(sx);
        ok = 1;
    }
    free(par); free(sp.p);
    return ok;
}

typedef struct { const float* px; int w, h; } FImg;

static float diff3(const FImg* a, int y1, int x1, const FImg* b, int y2, int x2) {   /* diffL2Square3<float> */
    const float* r1 = a->px + ((size_t)y1 * a->w + x1) * 3;
    const float* r2 = b->px + ((size_t)y2 * b->w + x2) * 3;
    const float d0 = r1[0] - r2[0], d1 = r1[1] - r2[1], d2 = r1[2] - r2[2];
    return (d0 * d0 + d1 * d1) + d2 * d2;
}

static int lab_or0(const Dp* d, int y, int x) { return (x >= 0 && x < d->uw && y >= 0 && y < d->uh) ? LAB(d, y, x) : 0; }

/* DpSeamFinder::estimateSeam (with computeCosts, COLOR).  [uncertain] labels one past the union count as "not this component". */
static int estimate_seam(const Dp* d, const FImg* im1, const FImg* im2, int tl1x, int tl1y, int tl2x, int tl2y, int comp, P2 p1, P2 p2, PVec* seam, int* horizontal) {
    const int l = comp + 1, rx = d->tls[comp].x, ry = d->tls[comp].y, rw = d->brs[comp].x - rx, rh = d->brs[comp].y - ry;
    const int dx1 = d->tlx - tl1x, dy1 = d->tly - tl1y, dx2 = d->tlx - tl2x, dy2 = d->tly - tl2y;
    const float bad = 3.f * 255.f * 255.f;
    float* cV = (float*)malloc(sizeof(float) * (size_t)(rw + 1) * rh);          /* rh x (rw + 1) */
    float* cH = (float*)malloc(sizeof(float) * (size_t)rw * (rh + 1));          /* (rh + 1) x rw */
    for (int y = ry; y < ry + rh; y++)
        for (int x = rx; x < rx + rw + 1; x++)
            cV[(size_t)(y - ry) * (rw + 1) + (x - rx)] =
                (lab_or0(d, y, x) == l && x > 0 && lab_or0(d, y, x - 1) == l)
                    ? (diff3(im1, y + dy1, x + dx1 - 1, im2, y + dy2, x + dx2) + diff3(im1, y + dy1, x + dx1, im2, y + dy2, x + dx2 - 1)) / 2 : bad;
    for (int y = ry; y < ry + rh + 1; y++)
        for (int x = rx; x < rx + rw; x++)
            cH[(size_t)(y - ry) * rw + (x - rx)] =
                (lab_or0(d, y, x) == l && y > 0 && lab_or0(d, y - 1, x) == l)
                    ? (diff3(im1, y + dy1 - 1, x + dx1, im2, y + dy2, x + dx2) + diff3(im1, y + dy1, x + dx1, im2, y + dy2 - 1, x + dx2)) / 2 : bad;
#define CV_(y, x) cV[(size_t)(y) * (rw + 1) + (x)]
#define CH_(y, x) cH[(size_t)(y) * rw + (x)]
    P2 src = {p1.x - rx, p1.y - ry}, dst = {p2.x - rx, p2.y - ry};
    int swapped = 0;
    const int horiz = abs(dst.x - src.x) > abs(dst.y - src.y);
    *horizontal = horiz;
    if (horiz ? src.x > dst.x : src.y > dst.y) { const P2 t = src; src = dst; dst = t; swapped = 1; }
    uint8_t* ctl = (uint8_t*)calloc((size_t)rw * rh * 2, 1);
    uint8_t* reach = ctl + (size_t)rw * rh;
    float* cost = (float*)calloc((size_t)rw * rh, sizeof(float));
#define AT(a, y, x) a[(size_t)(y) * rw + (x)]
    AT(reach, src.y, src.x) = 1;
    if (horiz) {
        for (int x = src.x + 1; x <= dst.x; x++)
            for (int y = 0; y < rh; y++) {
                if (LAB(d, y + ry, x + rx) != l) continue;
                float bc = 0; int bs = 0;   /* min over (cost, step) pairs, lexicographic */
                if (AT(reach, y, x - 1)) { bc = AT(cost, y, x - 1) + CH_(y, x - 1); bs = 1; }
                if (y > 0 && AT(reach, y - 1, x - 1)) { const float c = AT(cost, y - 1, x - 1) + CH_(y - 1, x - 1) + CV_(y - 1, x); if (!bs || c < bc) { bc = c; bs = 2; } }
                if (y < rh - 1 && AT(reach, y + 1, x - 1)) { const float c = AT(cost, y + 1, x - 1) + CH_(y + 1, x - 1) + CV_(y, x); if (!bs || c < bc) { bc = c; bs = 3; } }
                if (bs) { AT(cost, y, x) = bc; AT(ctl, y, x) = (uint8_t)bs; AT(reach, y, x) = 255; }
            }
    } else {
        for (int y = src.y + 1; y <= dst.y; y++)
            for (int x = 0; x < rw; x++) {
                if (LAB(d, y + ry, x + rx) != l) continue;
                float bc = 0; int bs = 0;
                if (AT(reach, y - 1, x)) { bc = AT(cost, y - 1, x) + CV_(y - 1, x); bs = 1; }
                if (x > 0 && AT(reach, y - 1, x - 1)) { const float c = AT(cost, y - 1, x - 1) + CV_(y - 1, x - 1) + CH_(y, x - 1); if (!bs || c < bc) { bc = c; bs = 2; } }
                if (x < rw - 1 && AT(reach, y - 1, x + 1)) { const float c = AT(cost, y - 1, x + 1) + CV_(y - 1, x + 1) + CH_(y, x); if (!bs || c < bc) { bc = c; bs = 3; } }
                if (bs) { AT(cost, y, x) = bc; AT(ctl, y, x) = (uint8_t)bs; AT(reach, y, x) = 255; }
            }
    }
    int ok = 0;
    if (AT(reach, dst.y, dst.x)) {
        P2 p = dst;
        seam->n = 0;
        pv_push(seam, p.x + rx, p.y + ry);
        while (horiz ? p.x != src.x : p.y != src.y) {
            const int c = AT(ctl, p.y, p.x);
            if (horiz) { if (c == 2) p.y--; else if (c == 3) p.y++; p.x--; }
            else { if (c == 2) p.x--; else if (c == 3) p.x++; p.y--; }
            pv_push(seam, p.x + rx, p.y + ry);
        }
        if (!swapped)
            for (int i = 0, j = seam->n - 1; i < j; i++, j--) { const P2 t = seam->p[i]; seam->p[i] = seam->p[j]; seam->p[j] = t; }
        ok = seam->p[0].x == p1.x && seam->p[0].y == p1.y && seam->p[seam->n - 1].x == p2.x && seam->p[seam->n - 1].y == p2.y;
    }
    free(cV); free(cH); free(ctl); free(cost);
    return ok;
#undef AT
#undef CV_
#undef CH_
}

/* DpSeamFinder::updateLabelsUsingSeam */
static void update_labels(Dp* d, int c1, int c2, const PVec* seam, int horizontal) {
    const int ox = d->tls[c1].x, oy = d->tls[c1].y, mw = d->brs[c1].x - ox, mh = d->brs[c1].y - oy, l1 = c1 + 1, l2 = c2 + 1;
    int* m = (int*)calloc((size_t)mw * mh, sizeof(int));
#define M(y, x) m[(size_t)(y) * mw + (x)]
    const PVec* ct = &d->contours[c1];
    for (int i = 0; i < ct->n; i++) M(ct->p[i].y - oy, ct->p[i].x - ox) = 255;
    for (int i = 0; i < seam->n; i++) M(seam->p[i].y - oy, seam->p[i].x - ox) = 255;
    int nc = 0;
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++)
            if (!M(y, x) && LAB(d, y + oy, x + ox) == l1) flood(m, mw, mh, x, y, ++nc);
    for (int i = 0; i < ct->n; i++) {
        const int x = ct->p[i].x - ox, y = ct->p[i].y - oy;
        static const int dx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, dy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
        int ok = 0;
        for (int j = 0; j < 8; j++) {
            const int c = x + dx[j], r = y + dy[j];
            if (c >= 0 && c < mw && r >= 0 && r < mh && M(r, c) && M(r, c) != 255) { ok = 1; M(y, x) = M(r, c); }
        }
        if (!ok) M(y, x) = 0;
    }
    for (int i = 0; i < seam->n; i++) {
        const int x = seam->p[i].x - ox, y = seam->p[i].y - oy;
        if (horizontal) M(y, x) = (y < mh - 1 && M(y + 1, x) && M(y + 1, x) != 255) ? M(y + 1, x) : 0;
        else M(y, x) = (x < mw - 1 && M(y, x + 1) && M(y, x + 1) != 255) ? M(y, x + 1) : 0;
    }
    const int nk = (nc > 255 ? nc : 255) + 1;
    int* con2 = (int*)calloc((size_t)nk * 3, sizeof(int));
    int *cono = con2 + nk, *adj = con2 + 2 * nk;
    for (int i = 0; i < ct->n; i++) {
        const int x = ct->p[i].x, y = ct->p[i].y, k = M(y - oy, x - ox);
        if (touches(d, y, x, l2)) con2[k]++;
        int o, other = 0;
        if (x > 0 && (o = LAB(d, y, x - 1)) != l1 && o != l2) other = 1;
        if (y > 0 && (o = LAB(d, y - 1, x)) != l1 && o != l2) other = 1;
        if (x < d->uw - 1 && (o = LAB(d, y, x + 1)) != l1 && o != l2) other = 1;
        if (y < d->uh - 1 && (o = LAB(d, y + 1, x)) != l1 && o != l2) other = 1;
        if (other) cono[k]++;
    }
    const double len = (double)ct->n;
    for (int k = 0; k < nk; k++) adj[k] = (con2[k] / len > 0.05 && cono[k] / len < 0.1) ? 1 : 0;
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++)
            if (M(y, x) && adj[M(y, x)]) LAB(d, y + oy, x + ox) = l2;
    free(con2); free(m);
#undef M
}

static void refresh_comp(Dp* d, int c) {
    const int l = c + 1, x0 = d->tls[c].x, x1 = d->brs[c].x, y0 = d->tls[c].y, y1 = d->brs[c].y;
    d->tls[c].x = d->tls[c].y = INT_MAX; d->brs[c].x = d->brs[c].y = INT_MIN;
    d->contours[c].n = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            if (LAB(d, y, x) == l) {
                if (x < d->tls[c].x) d->tls[c].x = x;
                if (y < d->tls[c].y) d->tls[c].y = y;
                if (x + 1 > d->brs[c].x) d->brs[c].x = x + 1;
                if (y + 1 > d->brs[c].y) d->brs[c].y = y + 1;
                if (on_border(d, y, x, l)) pv_push(&d->contours[c], x, y);
            }
}

/* DpSeamFinder::process + resolveConflicts for one pair; masks are edited in place */
static void process_pair(const FImg* im1, const FImg* im2, int tl1x, int tl1y, int tl2x, int tl2y, uint8_t* m1, uint8_t* m2) {
    const int w1 = im1->w, h1 = im1->h, w2 = im2->w, h2 = im2->h;
    const int ix0 = tl1x > tl2x ? tl1x : tl2x, iy0 = tl1y > tl2y ? tl1y : tl2y;
    const int ix1 = tl1x + w1 < tl2x + w2 ? tl1x + w1 : tl2x + w2, iy1 = tl1y + h1 < tl2y + h2 ? tl1y + h1 : tl2y + h2;
    if (ix0 >= ix1 || iy0 >= iy1) return;
    Dp d;
    memset(&d, 0, sizeof(d));
    d.tlx = tl1x < tl2x ? tl1x : tl2x; d.tly = tl1y < tl2y ? tl1y : tl2y;
    const int brx = tl1x + w1 > tl2x + w2 ? tl1x + w1 : tl2x + w2, bry = tl1y + h1 > tl2y + h2 ? tl1y + h1 : tl2y + h2;
    d.uw = brx - d.tlx; d.uh = bry - d.tly;
    const size_t un = (size_t)d.uw * d.uh;
    d.mask1 = (uint8_t*)calloc(un * 4, 1); d.mask2 = d.mask1 + un; d.cont1 = d.mask2 + un; d.cont2 = d.cont1 + un;
    d.labels = (int*)malloc(sizeof(int) * un);
    for (int y = 0; y < h1; y++) memcpy(d.mask1 + (size_t)(y + tl1y - d.tly) * d.uw + (tl1x - d.tlx), m1 + (size_t)y * w1, (size_t)w1);
    for (int y = 0; y < h2; y++) memcpy(d.mask2 + (size_t)(y + tl2y - d.tly) * d.uw + (tl2x - d.tlx), m2 + (size_t)y * w2, (size_t)w2);
    for (int y = 0; y < d.uh; y++)
        for (int x = 0; x < d.uw; x++) {
            const size_t o = (size_t)y * d.uw + x;
            const uint8_t* mk[2] = {d.mask1, d.mask2};
            uint8_t* ck[2] = {d.cont1, d.cont2};
            for (int k = 0; k < 2; k++)
                if (mk[k][o] && ((x == 0 || !mk[k][o - 1]) || (x == d.uw - 1 || !mk[k][o + 1]) || (y == 0 || !mk[k][o - d.uw]) || (y == d.uh - 1 || !mk[k][o + d.uw]))) ck[k][o] = 255;
        }
    find_components(&d);
    find_edges(&d);
    const int n = d.ncomps;
    PVec seam = {NULL, 0, 0};
    for (;;) {
        /* the first edge (c1, c2), in the order of std::set<std::pair<int, int>>, whose intersection component c1 is not yet
         * assigned to the image c2 belongs to */
        int c1 = -1, c2 = -1;
        for (int a = 0; a < n && c1 < 0; a++)
            for (int b = 0; b < n; b++)
                if (d.edges[(size_t)a * n + b] && (d.states[a] & S_INTERS) && (d.states[a] & ~S_INTERS) != d.states[b]) { c1 = a; c2 = b; break; }
        if (c1 < 0) break;
        const int l1 = c1 + 1, l2 = c2 + 1;
        int nb = 0;
        for (int b = 0; b < n; b++) nb += d.edges[(size_t)c1 * n + b];
        if (nb == 1) {                                     /* hasOnlyOneNeighbor: the whole component goes over */
            for (int y = d.tls[c1].y; y < d.brs[c1].y; y++)
                for (int x = d.tls[c1].x; x < d.brs[c1].x; x++)
                    if (LAB(&d, y, x) == l1) LAB(&d, y, x) = l2;
        } else {
            P2 p1, p2;
            int horizontal = 0;
            if (seam_tips(&d, c1, c2, &p1, &p2) && estimate_seam(&d, im1, im2, tl1x, tl1y, tl2x, tl2y, c1, p1, p2, &seam, &horizontal))
                update_labels(&d, c1, c2, &seam, horizontal);
        }
        d.states[c1] = d.states[c2] == S_FIRST ? (S_INTERS | S_SECOND) : (S_INTERS | S_FIRST);
        refresh_comp(&d, c1);
        refresh_comp(&d, c2);
        d.edges[(size_t)c1 * n + c2] = 0;                  /* [uncertain] the resolved edge leaves the graph, both directions */
        d.edges[(size_t)c2 * n + c1] = 0;
    }
    const int dx1 = d.tlx - tl1x, dy1 = d.tly - tl1y, dx2 = d.tlx - tl2x, dy2 = d.tly - tl2y;
    for (int y = 0; y < h2; y++)
        for (int x = 0; x < w2; x++) {
            const int l = LAB(&d, y - dy2, x - dx2);
            if (l > 0 && (d.states[l - 1] & S_FIRST) && m1[(size_t)(y - dy2 + dy1) * w1 + (x - dx2 + dx1)]) m2[(size_t)y * w2 + x] = 0;
        }
    for (int y = 0; y < h1; y++)
        for (int x = 0; x < w1; x++) {
            const int l = LAB(&d, y - dy1, x - dx1);
            if (l > 0 && (d.states[l - 1] & S_SECOND) && m2[(size_t)(y - dy1 + dy2) * w2 + (x - dx1 + dx2)]) m1[(size_t)y * w1 + x] = 0;
        }
    for (int c = 0; c < d.ncomps; c++) free(d.contours[c].p);
    free(seam.p); free(d.contours); free(d.states); free(d.tls); free(d.brs); free(d.edges); free(d.labels); free(d.mask1);
}

/* DpSeamFinder::find: all pairs, the most distant image centres first ([uncertain] ties: stable order, then reversed) */
int mo_seam_dp_color(int n, const int* corners_xy, const int* sizes_wh, const uint8_t* const* images_bgr, uint8_t* const* masks) {
    if (n <= 0) return 0;
    FImg* im = (FImg*)calloc((size_t)n, sizeof(FImg));
    for (int i = 0; i < n; i++) {
        const int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        float* f = (float*)malloc(sizeof(float) * (size_t)w * h * 3);
        for (size_t k = 0; k < (size_t)w * h * 3; k++) f[k] = (float)images_bgr[i][k];     /* convertTo(CV_32F) */
        im[i].px = f; im[i].w = w; im[i].h = h;
    }
    const int np = n * (n - 1) / 2;
    int* pr = (int*)malloc(sizeof(int) * (size_t)(np + 1) * 3);
    int k = 0;
    for (int i = 0; i + 1 < n; i++)
        for (int j = i + 1; j < n; j++) {
            const int ax = corners_xy[2 * i] + im[i].w / 2, ay = corners_xy[2 * i + 1] + im[i].h / 2;
            const int bx = corners_xy[2 * j] + im[j].w / 2, by = corners_xy[2 * j + 1] + im[j].h / 2;
            pr[3 * k] = (ax - bx) * (ax - bx) + (ay - by) * (ay - by); pr[3 * k + 1] = i; pr[3 * k + 2] = j; k++;
        }
    for (int a = 1; a < np; a++) {           /* insertion sort: stable, ascending distance */
        const int d0 = pr[3 * a], i0 = pr[3 * a + 1], j0 = pr[3 * a + 2];
        int b = a - 1;
        while (b >= 0 && pr[3 * b] > d0) { pr[3 * b + 3] = pr[3 * b]; pr[3 * b + 4] = pr[3 * b + 1]; pr[3 * b + 5] = pr[3 * b + 2]; b--; }
        pr[3 * b + 3] = d0; pr[3 * b + 4] = i0; pr[3 * b + 5] = j0;
    }
    for (int a = np - 1; a >= 0; a--) {      /* reversed */
        const int i = pr[3 * a + 1], j = pr[3 * a + 2];
        process_pair(&im[i], &im[j], corners_xy[2 * i], corners_xy[2 * i + 1], corners_xy[2 * j], corners_xy[2 * j + 1], masks[i], masks[j]);
    }
    for (int i = 0; i < n; i++) free((void*)im[i].px);
    free(im); free(pr);
    return 0;
}
