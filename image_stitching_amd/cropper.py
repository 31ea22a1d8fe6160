"""The reference's cropper API (image_stitching/cropper.h, cropper.cpp:6-209) in Python: checkInteriorExterior,
compareX, compareY, crop -- host logic on the blended panorama / its mask.  Mirror of host/cropper.cpp:
the OpenCV pieces (findContours RETR_EXTERNAL + CHAIN_APPROX_NONE, drawContours FILLED) are restated, parity
unpinned like every OpenCV-backed stage."""
import numpy as np


def checkInteriorExterior(mask, rect, codes):
    """mask: 2-D uint8; rect: (x, y, w, h); codes: dict with top / bottom / left / right (set to 1 in place, as the
    reference's int& out-parameters).  Returns True when the rectangle's border holds no exterior (zero) pixel."""
    x, y, w, h = rect
    sub = mask[y:y + h, x:x + w]
    top_row = int((sub[0, :] == 0).sum())
    bottom_row = int((sub[-1, :] == 0).sum())
    left_column = int((sub[:, 0] == 0).sum())
    right_column = int((sub[:, -1] == 0).sum())
    result = (top_row + bottom_row + left_column + right_column) == 0
    if top_row > bottom_row:
        if top_row > left_column and top_row > right_column:
            codes["top"] = 1
    elif bottom_row > left_column:
        if bottom_row > right_column:
            codes["bottom"] = 1
    if left_column >= right_column:
        if left_column >= bottom_row and left_column >= top_row:
            codes["left"] = 1
    elif right_column >= top_row:
        if right_column >= bottom_row:
            codes["right"] = 1
    return result


def compareX(a, b):
    return a[0] < b[0]


def compareY(a, b):
    return a[1] < b[1]


_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def findExternalContours(mask):
    """8-neighbour border following of every outer border that is not inside another one; points as (x, y) in
    visiting order, every border pixel kept (CHAIN_APPROX_NONE)."""
    h, w = mask.shape
    img = np.zeros((h + 2, w + 2), np.int32)
    img[1:-1, 1:-1] = (mask != 0)
    contours = []
    nbd = 2
    for y in range(1, h + 1):
        row = img[y]
        if not row.any():
            continue
        lnbd, prev = 0, 0
        for x in range(1, w + 1):
            p = int(row[x])
            if p == prev:
                continue
            if prev == 0 and p == 1:
                if not lnbd > 0:
                    c = []
                    s_end = s = 4
                    while True:
                        s = (s - 1) & 7
                        if img[y + _DY[s], x + _DX[s]] != 0 or s == s_end:
                            break
                    x1, y1 = x + _DX[s], y + _DY[s]
                    if s == s_end and img[y1, x1] == 0:
                        img[y, x] = -nbd
                        c.append((x - 1, y - 1))
                    else:
                        x3, y3 = x, y
                        while True:
                            s_end = s
                            while s < 15:
                                s += 1
                                x4, y4 = x3 + _DX[s & 7], y3 + _DY[s & 7]
                                if img[y4, x4] != 0:
                                    break
                            s &= 7
                            if 0 <= s - 1 < s_end:
                                img[y3, x3] = -nbd
                            elif img[y3, x3] == 1:
                                img[y3, x3] = nbd
                            c.append((x3 - 1, y3 - 1))
                            if (x4, y4) == (x, y) and (x3, y3) == (x1, y1):
                                break
                            x3, y3 = x4, y4
                            s = (s + 4) & 7
                    contours.append(c)
                    nbd += 1
                    p = int(img[y, x])
            elif p == 0 and prev >= 1 and (prev & -2):
                lnbd = prev
            prev = p
            if prev & -2:
                lnbd = prev
    return contours


def fillContour(contour, width, height):
    """drawContours(..., FILLED): 255 inside or on the contour (the complement of what the frame reaches through
    non-contour pixels with 4-connectivity)."""
    st = np.zeros((height + 2, width + 2), np.uint8)
    pts = np.asarray(contour, np.int64)
    st[pts[:, 1] + 1, pts[:, 0] + 1] = 1
    outside = np.zeros_like(st, bool)
    outside[0, :] = outside[-1, :] = outside[:, 0] = outside[:, -1] = True
    free = st == 0
    while True:   # 4-connected flood by repeated dilation (vectorised)
        grown = outside.copy()
        grown[1:, :] |= outside[:-1, :]
        grown[:-1, :] |= outside[1:, :]
        grown[:, 1:] |= outside[:, :-1]
        grown[:, :-1] |= outside[:, 1:]
        grown &= free
        grown |= outside
        if (grown == outside).all():
            break
        outside = grown
    return np.where(outside[1:-1, 1:-1], 0, 255).astype(np.uint8)


def crop(source):
    """crop(cv::Mat& source): -> (cropped image, (x, y, w, h)).  source: (H, W, 3) or (H, W) uint8."""
    source = np.asarray(source)
    if source.ndim == 3:
        g = (source[..., 0].astype(np.int64) * 9798 + source[..., 1].astype(np.int64) * 19235 + source[..., 2].astype(np.int64) * 3735 + (1 << 14)) >> 15
    else:
        g = source
    mask = np.where(g > 0, 255, 0).astype(np.uint8)
    contours = findExternalContours(mask)
    if not contours:
        raise ValueError("crop: the image is empty")
    best = max(range(len(contours)), key=lambda i: (len(contours[i]), -i))     # first contour of maximal length
    cmask = fillContour(contours[best], mask.shape[1], mask.shape[0])
    xs = sorted(p[0] for p in contours[best])
    ys = sorted(p[1] for p in contours[best])
    minx, maxx, miny, maxy = 0, len(xs) - 1, 0, len(ys) - 1
    rect = (0, 0, 0, 0)
    while minx < maxx and miny < maxy:
        rect = (xs[minx], ys[miny], xs[maxx] - xs[minx], ys[maxy] - ys[miny])
        if rect[2] <= 0 or rect[3] <= 0:
            break
        codes = dict(top=0, bottom=0, left=0, right=0)
        if checkInteriorExterior(cmask, rect, codes):
            break
        minx += codes["left"]
        maxx -= codes["right"]
        miny += codes["top"]
        maxy -= codes["bottom"]
    if rect[2] <= 0 or rect[3] <= 0:
        raise ValueError("crop: no interior rectangle found")
    x, y, w, h = rect
    return source[y:y + h, x:x + w].copy(), rect
