"""GPU parity: spherical warp (HIP, through the C ABI) vs the CPU oracle, bit-exact."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cam(w, h, hfov, yaw, pitch=0.0, roll=0.0):
    import synth
    return synth.make_camera(w, h, hfov, yaw, pitch, roll)


CASES = [
    (480, 270, 60.0, 0.0, 0.0, 0.0),
    (480, 270, 60.0, 25.0, 1.0, -0.5),
    (333, 217, 75.0, -100.0, -8.0, 3.0),     # odd sizes, far yaw
    (256, 256, 90.0, 10.0, 70.0, 0.0),       # pole inside the frame: roi reaches v = pi*scale
    (640, 200, 50.0, 170.0, 0.0, 0.0),       # straddles the +-180 seam: full-width roi
]


@pytest.mark.parametrize("case", CASES)
def test_warp_fused_bit_exact(ctx, oracle_mod, case):
    import torch
    import image_stitching_amd as isa
    w, h, hfov, yaw, pitch, roll = case
    cam = _cam(w, h, hfov, yaw, pitch, roll)
    rng = np.random.default_rng(hash(case) & 0xffff)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
    scale = float(np.float32(cam["f"]))
    ref_img, ref_tl = oracle_mod.warp_spherical(img, scale, K, R)
    ref_msk, _ = oracle_mod.warp_spherical(np.full((h, w), 255, np.uint8), scale, K, R, oracle_mod.INTER_NEAREST,
                                           oracle_mod.BORDER_CONSTANT)
    assert isa.warp_roi(scale, (w, h), K, R) == oracle_mod.warp_roi(scale, w, h, K, R)
    warper = isa.SphericalWarper(ctx, scale)
    tl, out, msk = warper.warp_fused(torch.from_numpy(img).cuda(), K, R)
    ctx.synchronize()
    assert tl == ref_tl
    assert np.array_equal(out.cpu().numpy(), ref_img.astype(np.int16))
    assert np.array_equal(msk.cpu().numpy(), ref_msk)


@pytest.mark.parametrize("cn", [1, 3])
def test_warp_general_modes_bit_exact(ctx, oracle_mod, cn):
    import torch
    import image_stitching_amd as isa
    w, h = 301, 173
    cam = _cam(w, h, 65.0, -33.0, 4.0, 2.0)
    rng = np.random.default_rng(11 + cn)
    img = rng.integers(0, 256, (h, w, 3) if cn == 3 else (h, w), dtype=np.uint8)
    K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
    scale = float(np.float32(cam["f"] * 0.37))   # seam-scale style warper (scale * seam_work_aspect)
    Ks = K.copy()
    Ks[0, 0] *= 0.37; Ks[0, 2] *= 0.37; Ks[1, 1] *= 0.37; Ks[1, 2] *= 0.37
    warper = isa.SphericalWarper(ctx, scale)
    for interp, border in ((isa.INTER_LINEAR, isa.BORDER_REFLECT), (isa.INTER_NEAREST, isa.BORDER_CONSTANT)):
        ref, ref_tl = oracle_mod.warp_spherical(img, scale, Ks, R, interp, border)
        tl, out = warper.warp(torch.from_numpy(img).cuda(), Ks, R, interp, border)
        assert tl == ref_tl
        assert np.array_equal(out.cpu().numpy(), ref)
        # host-buffer path of the ABI (staged through HBM) gives the same bytes
        tl2, out2 = warper.warp(img, Ks, R, interp, border)
        assert tl2 == ref_tl and np.array_equal(out2.cpu().numpy(), ref)


def test_warp_full_size_properties(ctx):
    """4K frame (BASELINE config 3 size): size-independent properties instead of the slow oracle."""
    import torch
    import image_stitching_amd as isa
    w, h = 3840, 2160
    cam = _cam(w, h, 60.0, 15.0, 0.3, -0.2)
    K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
    scale = float(np.float32(cam["f"]))
    warper = isa.SphericalWarper(ctx, scale)
    const = torch.full((h, w, 3), 137, dtype=torch.uint8, device="cuda")
    tl, out, msk = warper.warp_fused(const, K, R)
    x, y, rw, rh = isa.warp_roi(scale, (w, h), K, R)
    assert tl == (x, y) and tuple(out.shape) == (rh, rw, 3) and tuple(msk.shape) == (rh, rw)
    # a constant image warps to the same constant everywhere (weights sum to 32768, REFLECT border)
    assert int(out.min()) == 137 and int(out.max()) == 137
    m = msk.cpu().numpy()
    assert set(np.unique(m)) <= {0, 255}
    # mask area ~ image area at mid latitudes (spherical projection is near-isometric at scale = f)
    assert 0.75 < (m == 255).sum() / (w * h) < 1.1
    # linearity in the source: warp(a) + warp(b) == warp(a + b) up to the two rounding steps
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randint(0, 100, (h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 100, (h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    _, wa, _ = warper.warp_fused(a, K, R)
    _, wb, _ = warper.warp_fused(b, K, R)
    _, wab, _ = warper.warp_fused(a + b, K, R)
    assert int((wa.int() + wb.int() - wab.int()).abs().max()) <= 1


def test_fused_batch_equals_single_launches(ctx):
    """mis_warp_spherical_fused_batch (all frames in one grid per 16 frames; 18 frames = two grids, two frame sizes' worth of
    rois) writes what 18 mis_warp_spherical_fused_roi calls write."""
    import ctypes as C
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd import _capi as capi
    from image_stitching_amd.stitching import as_image
    w, h = 640, 360
    cams = [synth.make_camera(w, h, 60.0, 9.0 * i - 70.0, 0.6 * ((i % 3) - 1), 0.5 * ((i % 2) - 0.5)) for i in range(18)]
    frames = [torch.from_numpy(synth.render_frame(c)).cuda() for c in cams]
    scale = isa.Stitcher.warped_image_scale(cams)
    warper = isa.SphericalWarper(ctx, scale)
    rois = isa.stitching.warp_rois(ctx, scale, (w, h), cams)
    single = [warper.warp_fused(f, c["K"], c["R"], r) for f, c, r in zip(frames, cams, rois)]
    outs = [warper.alloc_fused(r) for r in rois]
    for o in outs:
        o[0].fill_(-7); o[1].fill_(9)
    n = len(cams)
    im = (capi.MisImage * n)(*[as_image(f) for f in frames])
    ds = (capi.MisImage * n)(*[as_image(o[0]) for o in outs])
    ms = (capi.MisImage * n)(*[as_image(o[1]) for o in outs])
    Ks = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float32).reshape(9) for c in cams]))
    Rs = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float32).reshape(9) for c in cams]))
    rr = (capi.MisRect * n)(*[capi.MisRect(*[int(v) for v in r]) for r in rois])
    tls = (capi.MisPoint * n)()
    fp = C.POINTER(C.c_float)
    ctx.check(ctx.lib.mis_warp_spherical_fused_batch(ctx.h, im, n, float(scale), Ks.ctypes.data_as(fp), Rs.ctypes.data_as(fp), rr, ds, ms, tls))
    ctx.synchronize()
    for (tl, img, msk), (bi, bm), t, r in zip(single, outs, tls, rois):
        assert (t.x, t.y) == tuple(tl) == (r[0], r[1])
        assert torch.equal(bi, img) and torch.equal(bm, msk)
