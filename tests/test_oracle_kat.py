"""Known-answer tests that pin the CPU oracle (no GPU).

Only the rotation math is pinned by the reference itself (tests/golden/rotation_kat.json, captured
from image_stitching/euler.h + quaternion.h); everything OpenCV-backed is checked
against closed-form expectations of the published algorithms ("parity unpinned")."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rotation_kat_pinned_by_reference(oracle_mod):
    o = oracle_mod
    kat = json.load(open(os.path.join(HERE, "golden", "rotation_kat.json")))
    e = np.array(kat["euler_in_rad"])
    R = o.euler_to_rot(e, kat["order"])
    back = o.rot_to_euler(R, kat["order"])
    assert np.array_equal(back, np.array(kat["euler_roundtrip"]))
    q = o.quat_from_rot(R)
    assert np.array_equal(q, np.array(kat["quaternion_xyzw"]))
    Rp = o.camera_rehand(R, is_portrait=False)
    assert np.array_equal(Rp, np.array(kat["R_rehanded_landscape"]))
    ef = o.rot_to_euler(Rp.astype(np.float32), "YXZ", np.float32)
    assert np.allclose(ef, np.array(kat["euler_float_of_rehanded_YXZ"], np.float32), rtol=0, atol=1e-7)


@pytest.mark.parametrize("order", ["XYZ", "YXZ", "ZXY", "ZYX", "YZX", "XZY"])
def test_euler_roundtrip_all_orders(oracle_mod, order):
    o = oracle_mod
    rng = np.random.default_rng(7)
    for _ in range(50):
        e = rng.uniform(-1.2, 1.2, 3)
        R = o.euler_to_rot(e, order)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert np.allclose(o.rot_to_euler(R, order), e, atol=1e-12)


def test_euler_gimbal_branch(oracle_mod):
    o = oracle_mod
    R = o.euler_to_rot(np.array([math.pi / 2, 0.3, 0.0]), "YXZ")  # |m23| = 1 -> second branch, z = 0
    e = o.rot_to_euler(R, "YXZ")
    assert e[2] == 0.0 and abs(e[0] - math.pi / 2) < 1e-7


def test_quaternion_all_branches(oracle_mod):
    o = oracle_mod
    # trace > 0, m11 largest, m22 largest, m33 largest
    for e in ([0.1, 0.2, 0.3], [3.0, 0.1, 0.1], [0.1, 3.0, 0.1], [0.1, 0.1, 3.0]):
        R = o.euler_to_rot(np.array(e), "XYZ")
        q = o.quat_from_rot(R)
        assert abs(np.linalg.norm(q) - 1) < 1e-12
        assert np.allclose(o.quat_to_rot(q), R, atol=1e-12)


def test_rng_and_pattern(oracle_mod):
    # cv::RNG(0x34985739) multiply-with-carry: first values from the recurrence itself
    state = 0x34985739
    vals = []
    for _ in range(4):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        vals.append(state & 0xFFFFFFFF)
    orb = oracle_mod.Orb(64, 64)
    pat = orb.pattern()
    exp = [int(v % 41) - 20 for v in vals]
    assert [int(pat[0, 0]), int(pat[0, 1]), int(pat[1, 0]), int(pat[1, 1])] == exp
    assert pat.min() >= -20 and pat.max() <= 20
    # umax of the 40-pixel patch: quarter disc of radius 20, symmetric octagon rule
    um = orb.umax()
    assert um[0] == 20 and um[20] == 4 and um[15] == 13 and all(um[i] >= um[i + 1] for i in range(20))


def test_fast_atan2_matches_atan2(oracle_mod):
    L = oracle_mod.lib()
    for ang in np.linspace(0, 359.5, 721):
        y, x = math.sin(math.radians(ang)), math.cos(math.radians(ang))
        a = L.mo_fast_atan2(np.float32(y * 37.0), np.float32(x * 37.0))
        d = abs(a - ang)
        assert min(d, 360 - d) < 0.3
    assert L.mo_fast_atan2(0.0, 0.0) == 0.0


def test_trig_polynomials(oracle_mod):
    L = oracle_mod.lib()
    xs = np.linspace(-3.2, 6.4, 2001).astype(np.float32)
    for x in xs:
        assert abs(L.mo_sinf(x) - math.sin(float(x))) < 3e-7
        assert abs(L.mo_cosf(x) - math.cos(float(x))) < 3e-7
    for x in np.linspace(-1, 1, 401).astype(np.float32):
        assert abs(L.mo_acosf(x) - math.acos(float(x))) < 1e-6
    for a in np.linspace(-3.1, 3.1, 401):
        assert abs(L.mo_atan2f(np.float32(math.sin(a)), np.float32(math.cos(a))) - a) < 1e-6
    for x in [1e-300, 0.005, 0.3, 0.9999, 1.0, 7.5, 1e10]:
        assert abs(L.mo_log_d(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))


def test_gray_q15_and_gauss_kernel(oracle_mod):
    o = oracle_mod
    img = np.zeros((1, 4, 3), np.uint8)
    img[0, 0] = (255, 255, 255)
    img[0, 1] = (255, 0, 0)
    img[0, 2] = (0, 255, 0)
    img[0, 3] = (0, 0, 255)
    g = o.bgr2gray(img)[0]
    assert list(g) == [255, (255 * 3735 + 16384) >> 15, (255 * 19235 + 16384) >> 15, (255 * 9798 + 16384) >> 15]
    assert o.gauss7_kernel_q8() == [18, 34, 48, 56, 48, 34, 18]


def test_orb_level_budget(oracle_mod):
    orb = oracle_mod.Orb(3840, 2160)
    assert [orb.level_nfeatures(l) for l in range(8)] == [869, 724, 603, 503, 419, 349, 291, 242]
    assert [orb.level_size(l) for l in range(8)] == [(3840, 2160), (3200, 1800), (2667, 1500), (2222, 1250),
                                                    (1852, 1042), (1543, 868), (1286, 723), (1072, 603)]


def test_resize_linear_exact_identity_and_constant(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(o.resize_linear_exact(a, 53, 37), a)
    c = np.full((40, 60), 77, np.uint8)
    assert np.all(o.resize_linear_exact(c, 50, 33) == 77)
    # a horizontal ramp stays monotone
    r = np.tile(np.arange(0, 240, 4, dtype=np.uint8), (10, 1))
    d = o.resize_linear_exact(r, 50, 10)
    assert np.all(np.diff(d.astype(int), axis=1) >= 0)


def test_pyramids_closed_form(oracle_mod):
    o = oracle_mod
    # constant images stay constant through pyrDown / pyrUp
    c = np.full((16, 24, 3), 1000, np.int16)
    assert np.all(o.pyr_down_s16(c) == 1000)
    assert np.all(o.pyr_up_s16(o.pyr_down_s16(c)) == 1000)
    w = np.full((16, 24), 0.5, np.float32)
    assert np.all(o.pyr_down_f32(w) == 0.5)
    # impulse response of pyrDown: centre tap 36/256 of 25600 = 3600
    imp = np.zeros((16, 16), np.int16)
    imp[8, 8] = 25600
    d = o.pyr_down_s16(imp)
    assert d[4, 4] == 3600 and d[4, 3] == 100 * 6 and d[3, 3] == 100
    # pyrUp impulse: even sample gets 36/64, neighbours 24/64 and 16/64
    imp2 = np.zeros((8, 8), np.int16)
    imp2[4, 4] = 6400
    u = o.pyr_up_s16(imp2)
    assert u[8, 8] == 3600 and u[8, 9] == 2400 and u[9, 9] == 1600 and u[8, 7] == 2400 and u[7, 7] == 1600


def test_distance_l1(oracle_mod):
    o = oracle_mod
    m = np.full((9, 11), 255, np.uint8)
    m[4, 5] = 0
    d = o.distance_l1(m)
    yy, xx = np.mgrid[0:9, 0:11]
    assert np.array_equal(d, (np.abs(yy - 4) + np.abs(xx - 5)).astype(np.float32))
    assert np.all(o.distance_l1(np.full((5, 5), 255, np.uint8)) == 8192.0)


def test_knn2_and_ratio_union(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t[17] = q[5]
    t[250] = q[5]  # exact duplicate: tie on distance 0 must resolve to the smaller train index
    idx, dist = o.knn2_hamming(q, t)
    ref = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    order = np.lexsort((np.arange(300)[None, :].repeat(200, 0), ref), axis=1)
    assert np.array_equal(idx, order[:, :2].astype(np.int32))
    assert idx[5, 0] == 17 and idx[5, 1] == 250 and dist[5, 0] == 0 and dist[5, 1] == 0


def test_homography_exact_recovery(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(5)
    H = np.array([[1.02, 0.01, 30.0], [-0.02, 0.98, -12.0], [1e-5, -2e-5, 1.0]])
    src = rng.uniform(-400, 400, (300, 2)).astype(np.float32)
    p = np.c_[src, np.ones(300)] @ H.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst[:60] += rng.uniform(30, 80, (60, 2)).astype(np.float32)  # 20 % outliers
    ok, Hest, mask, iters = o.find_homography_ransac(src, dst)
    assert ok and mask[60:].all() and mask[:60].sum() == 0
    assert np.allclose(Hest, H, rtol=0, atol=2e-3)
    assert 1 <= iters < 2000  # adaptive early exit
    ok4, H4 = o.homography_dlt(src[100:104], dst[100:104])
    assert ok4 and np.allclose(H4, H, atol=5e-2)


def test_jacobi_eigen(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(9)
    A = rng.normal(size=(9, 9))
    A = A @ A.T
    W, V = o.jacobi_eigen(A)
    assert np.all(np.diff(W) <= 0)
    assert np.allclose(V @ A @ V.T, np.diag(W), atol=1e-9)


def test_ransac_num_iters(oracle_mod):
    o = oracle_mod
    assert o.ransac_update_num_iters(0.995, 0.0, 2000) == 0
    assert o.ransac_update_num_iters(0.995, 1.0, 2000) == 2000
    n = o.ransac_update_num_iters(0.995, 0.5, 2000)
    assert n == round(math.log(0.005) / math.log(1 - 0.5 ** 4))


def test_leave_biggest_component(oracle_mod):
    conf = np.zeros((5, 5))
    for a, b in ((0, 1), (1, 2), (3, 4)):
        conf[a, b] = conf[b, a] = 2.0
    conf[2, 3] = conf[3, 2] = 0.5  # below 0.95: does not connect
    assert list(oracle_mod.leave_biggest_component(conf, 0.95)) == [0, 1, 2]


def test_warp_roi_and_identity(oracle_mod):
    o = oracle_mod
    w, h, f = 320, 200, 300.0
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)
    x, y, rw, rh = o.warp_roi(f, w, h, K, R)
    # optical axis maps to u = 0, v = pi/2 * scale
    assert x < 0 < x + rw and y < f * math.pi / 2 < y + rh
    P = o.projector(f, K, R)
    u, v = o.map_forward(P, w / 2, h / 2)
    assert abs(u) < 1e-3 and abs(v - f * math.pi / 2) < 1e-3
    bx, by = o.map_backward(P, u, v)
    assert abs(bx - w / 2) < 1e-2 and abs(by - h / 2) < 1e-2
    img = np.random.default_rng(2).integers(0, 256, (h, w, 3), dtype=np.uint8)
    dst, tl = o.warp_spherical(img, f, K, R)
    msk, tl2 = o.warp_spherical(np.full((h, w), 255, np.uint8), f, K, R, o.INTER_NEAREST, o.BORDER_CONSTANT)
    assert tl == tl2 == (x, y) and dst.shape[:2] == (rh, rw) == msk.shape
    assert 0.5 < (msk == 255).mean() <= 1.0


def test_blender_single_image_roundtrip(oracle_mod):
    """Feeding one fully-masked image must reproduce it (Laplacian pyramid is exactly invertible)."""
    o = oracle_mod
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (96, 128, 3)).astype(np.int16)
    mask = np.full((96, 128), 255, np.uint8)
    for btype in (o.BLEND_MULTI_BAND, o.BLEND_FEATHER, o.BLEND_NO):
        b = o.Blender(btype, 3, 0.05)
        b.prepare([(10, 20)], [(128, 96)])
        b.feed(img, mask, (10, 20))
        dst, m = b.blend()
        assert np.all(m == 255)
        # every level loses < 1 LSB to the (short)(x / (w + 1e-5)) truncation, as in OpenCV
        tol = {o.BLEND_MULTI_BAND: 4, o.BLEND_FEATHER: 1, o.BLEND_NO: 0}[btype]
        assert np.abs(dst.astype(int) - img.astype(int)).max() <= tol


def test_blend_config_reference_sizing(oracle_mod):
    o = oracle_mod
    t, nb, sh = o.blend_config(o.BLEND_MULTI_BAND, 5.0, 16500, 2200)
    assert t == o.BLEND_MULTI_BAND and nb == 8  # 8 bands for the config-3 panorama
    t, nb, sh = o.blend_config(o.BLEND_FEATHER, 5.0, 1000, 1000)
    assert t == o.BLEND_FEATHER and abs(sh - 1 / 50.0) < 1e-7
    assert o.blend_config(o.BLEND_MULTI_BAND, 5.0, 10, 10)[0] == o.BLEND_NO


# ---- image operators (mo_imgops.c): closed-form known answers ---------------------------------
def test_resize_exact_known_answers(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    # identity scale reproduces the image; a constant image stays constant for any scale
    assert np.array_equal(o.resize_exact(img, fx=1.0, fy=1.0), img)
    c = np.full((9, 14), 173, np.uint8)
    assert np.all(o.resize_exact(c, fx=0.37, fy=0.61) == 173)
    assert np.all(o.resize_exact(c, dsize=(31, 17)) == 173)
    # exact 2x decimation: destination i samples 2i + 0.5 -> mean of two neighbours (weights 128/128), rounded half up
    a = np.arange(16, dtype=np.uint8).reshape(1, 16) * 10
    a = np.repeat(a, 2, 0)
    r = o.resize_exact(a, fx=0.5, fy=0.5)
    assert r.shape == (1, 8) and list(r[0]) == [5, 25, 45, 65, 85, 105, 125, 145]
    # dsize by factor rounds half to even (cvRound): 5 * 0.5 = 2.5 -> 2, 7 * 0.5 = 3.5 -> 4
    assert o.resize_exact(np.zeros((5, 7), np.uint8), fx=0.5, fy=0.5).shape == (2, 4)


def test_rotate_dilate_and_seam_mask_known_answers(oracle_mod):
    o = oracle_mod
    a = np.arange(6, dtype=np.uint8).reshape(2, 3)
    assert o.rotate(a, 0).tolist() == [[3, 0], [4, 1], [5, 2]]          # 90 clockwise
    assert o.rotate(a, 1).tolist() == [[5, 4, 3], [2, 1, 0]]            # 180
    assert o.rotate(a, 2).tolist() == [[2, 5], [1, 4], [0, 3]]          # 90 counter-clockwise
    rgb = np.arange(24, dtype=np.uint8).reshape(2, 4, 3)
    assert np.array_equal(o.rotate(o.rotate(rgb, 0), 2), rgb) and np.array_equal(o.rotate(o.rotate(rgb, 1), 1), rgb)
    m = np.zeros((5, 5), np.uint8)
    m[2, 2] = 255
    d = o.dilate3x3(m)
    assert d[1:4, 1:4].min() == 255 and d.sum() == 9 * 255
    m[0, 0] = 200                                                       # the border contributes nothing
    assert o.dilate3x3(m)[0:2, 0:2].min() == 200
    # seam mask: all-255 seam leaves the mask untouched, all-zero seam clears it
    mask = np.full((40, 60), 255, np.uint8)
    mask[:, :7] = 0
    assert np.array_equal(o.seam_mask_apply(np.full((10, 15), 255, np.uint8), mask), mask)
    assert o.seam_mask_apply(np.zeros((10, 15), np.uint8), mask).max() == 0
    # a one-pixel hole in the seam mask is closed by the dilation
    seam = np.full((10, 15), 255, np.uint8)
    seam[5, 7] = 0
    assert np.array_equal(o.seam_mask_apply(seam, mask), mask)


# ---- SIFT (mo_sift.c): closed-form expectations of the published algorithm --------------------
def test_sift_blob_scale_and_position(oracle_mod):
    o = oracle_mod
    h, w, s0 = 128, 160, 6.0
    yy, xx = np.mgrid[0:h, 0:w]
    g = 200 * np.exp(-((xx - 80) ** 2 + (yy - 64) ** 2) / (2 * s0 ** 2))
    img = np.repeat(g[..., None], 3, 2).astype(np.uint8)
    s = o.Sift(w, h)
    k, d = s.run(img)
    assert s.num_octaves() == 7 and len(k) >= 1
    # the blob's DoG extremum: centre within half a pixel, characteristic scale sigma ~ s0 (size = 2 sigma, within 20 %)
    assert np.all(np.abs(k["x"] - 80) < 0.75) and np.all(np.abs(k["y"] - 64) < 0.75)
    assert np.all(np.abs(k["size"] / 2 - s0) < 0.2 * s0)
    assert np.array_equal(d, np.rint(d)) and d.max() <= 255 and d.min() >= 0
    assert np.all(np.abs(np.linalg.norm(d, axis=1) - 512) < 24)
    # a flat image has no extrema
    assert len(o.Sift(64, 48).run(np.full((48, 64, 3), 31, np.uint8))[0]) == 0


def test_sift_helpers(oracle_mod):
    import math
    o = oracle_mod
    assert 0.0 <= o.expf(-103.0) < 1e-44                                  # denormal range: coarse
    for x in [-80.0, -20.5, -1.0, -1e-3, 0.0, 0.3, 1.0, 10.0, 88.0]:
        assert abs(o.expf(x) - math.exp(x)) <= 2e-7 * math.exp(x)
    assert o.expf(-200.0) == 0.0
    t = o.gaussian_taps_f32(1.6)
    assert len(t) == 15 and abs(float(t.sum()) - 1.0) < 1e-6 and np.array_equal(t, t[::-1]) and t.argmax() == 7
    assert len(o.gaussian_taps_f32(1.2489996)) == 11


# ---- exposure compensation / Voronoi seams (mo_expos.c; parity unpinned: analytic checks only) ----
def test_oracle_lu_solve_matches_numpy(oracle_mod):
    oracle = oracle_mod
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 40):
        A = rng.normal(size=(n, n)) + np.eye(n) * 3
        x = rng.normal(size=n)
        got = oracle.solve_lu(A, A @ x)
        assert got is not None and np.abs(got - x).max() < 1e-10
    assert oracle.solve_lu(np.zeros((3, 3)), np.ones(3)) is None      # singular -> cv::solve returns false


def test_oracle_gain_compensator_pulls_two_exposures_together(oracle_mod):
    oracle = oracle_mod
    rng = np.random.default_rng(0)
    base = rng.integers(40, 200, (300, 700, 3)).astype(np.float32)
    a = np.clip(base[:, :400] * 0.8, 0, 255).astype(np.uint8)
    b = np.clip(base[:, 300:] * 1.2, 0, 255).astype(np.uint8)
    ma, mb = np.full(a.shape[:2], 255, np.uint8), np.full(b.shape[:2], 255, np.uint8)
    c = oracle.Compensator(64, 64, 2)
    c.feed([(0, 0), (300, 0)], [a, b], [ma, mb])
    ga, gb = c.gain_map(0), c.gain_map(1)
    assert ga.shape == (5, 7) and gb.shape == (5, 7)                   # ceil(300/64) x ceil(400/64)
    assert np.all(ga[:, :3] == 1.0) and np.all(gb[:, 4:] == 1.0)       # blocks that meet no other image keep gain 1
    assert ga[:, -1].min() > 1.05 and gb[:, 0].max() < 0.9              # the dark image is lifted, the bright one lowered
    oa, ob = c.apply(0, a), c.apply(1, b)
    before = np.abs(a[:, 300:].astype(int) - b[:, :100].astype(int)).mean()
    after = np.abs(oa[:, 300:].astype(int) - ob[:, :100].astype(int)).mean()
    assert after < 0.4 * before
    # a single image: every gain stays exactly 1 and apply is the identity
    c.feed([(0, 0)], [a], [ma])
    assert np.all(c.gain_map(0) == 1.0) and np.array_equal(c.apply(0, a), a)


def test_oracle_voronoi_splits_a_symmetric_overlap_in_the_middle(oracle_mod):
    oracle = oracle_mod
    ma, mb = np.full((50, 400), 255, np.uint8), np.full((50, 400), 255, np.uint8)
    a, b = oracle.voronoi_seams([(0, 0), (300, 0)], [ma, mb])
    # overlap = pano x 300..399; equidistant pixels go to the first image ("dist1 < dist2" is strict -> mask1 is cleared)
    assert np.all(a[:, :349] == 255) and np.all(a[:, 350:] == 0)
    assert np.all(b[:, :49] == 0) and np.all(b[:, 50:] == 255)
    assert ((a[:, 300:] > 0).astype(int) + (b[:, :100] > 0).astype(int)).max() == 1
    # disjoint images are left alone
    c, d = oracle.voronoi_seams([(0, 0), (500, 0)], [ma, mb])
    assert np.array_equal(c, ma) and np.array_equal(d, mb)
