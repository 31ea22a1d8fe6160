"""GPU parity: multi-band / feather / plain blender (HIP, through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(rng, n, w, h):
    """n overlapping warped-like frames: s16 image, 0/255 mask with ragged borders, corners."""
    out = []
    for i in range(n):
        ww, hh = w + int(rng.integers(-20, 20)), h + int(rng.integers(-15, 15))
        img = rng.integers(0, 256, (hh, ww, 3)).astype(np.int16)
        yy, xx = np.mgrid[0:hh, 0:ww]
        mask = ((xx > 3 + (yy // 7) % 5) & (xx < ww - 4) & (yy > 2) & (yy < hh - 3 - (xx // 9) % 4)).astype(np.uint8) * 255
        tl = (int(-40 + i * (w * 0.6) + rng.integers(-5, 5)), int(100 + rng.integers(-12, 12)))
        out.append((img, mask, tl))
    return out


def _run_both(ctx, oracle_mod, btype, frames, bands=4, sharp=0.03, check_levels=False):
    import torch
    import image_stitching_amd as isa
    corners = [f[2] for f in frames]
    sizes = [(f[0].shape[1], f[0].shape[0]) for f in frames]
    ob = oracle_mod.Blender(btype, bands, sharp)
    ob.prepare(corners, sizes)
    gb = {isa.BLEND_MULTI_BAND: lambda: isa.MultiBandBlender(ctx, bands), isa.BLEND_FEATHER: lambda: isa.FeatherBlender(ctx, sharp),
          isa.BLEND_NO: lambda: isa.Blender(ctx)}[btype]()
    gb.prepare(corners, sizes)
    for img, mask, tl in frames:
        ob.feed(img, mask, tl)
        gb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), tl)
    if check_levels:
        assert gb.numBands() == ob.num_bands
        for l in range(ob.num_bands + 1):
            olap, owgt = ob.level(l)
            glap, gwgt = gb.level(l)
            assert np.array_equal(glap, olap), "laplacian level %d" % l
            assert np.array_equal(gwgt.view(np.uint32), owgt.view(np.uint32)), "weight level %d" % l
    oref, omask = ob.blend()
    gout, gmask = gb.blend()
    ctx.synchronize()
    assert np.array_equal(gmask.cpu().numpy(), omask)
    assert np.array_equal(gout.cpu().numpy(), oref)
    return oref


def test_multiband_bit_exact(ctx, oracle_mod):
    rng = np.random.default_rng(21)
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, _frames(rng, 3, 200, 150), bands=4, check_levels=True)


def test_multiband_full_s16_range_bit_exact(ctx, oracle_mod):
    """feed() takes any 16SC3 image: values outside 0..255 leave the packed 16-bit pyrDown path (per tile: one frame is
    all wide-range, one is 8-bit except for a patch, one has negative values only in a corner)."""
    rng = np.random.default_rng(27)
    frames = _frames(rng, 3, 260, 190)
    a = rng.integers(-32768, 32768, frames[0][0].shape).astype(np.int16)
    b = frames[1][0].copy(); b[60:90, 100:140] = rng.integers(-3000, 3000, (30, 40, 3)).astype(np.int16)
    c = frames[2][0].copy(); c[:9, :11] = -7
    frames = [(a, frames[0][1], frames[0][2]), (b, frames[1][1], frames[1][2]), (c, frames[2][1], frames[2][2])]
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, frames, bands=5, check_levels=True)


def test_multiband_band_crop_and_single_frame(ctx, oracle_mod):
    rng = np.random.default_rng(22)
    # more bands requested than the panorama supports -> prepare() crops them
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, _frames(rng, 1, 90, 70), bands=9, check_levels=True)
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, _frames(rng, 2, 64, 48), bands=0, check_levels=True)


def test_feather_and_plain_bit_exact(ctx, oracle_mod):
    rng = np.random.default_rng(23)
    fr = _frames(rng, 3, 180, 120)
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_FEATHER, fr, sharp=0.05)
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_NO, fr)


def test_multiband_empty_mask_frame(ctx, oracle_mod):
    rng = np.random.default_rng(24)
    fr = _frames(rng, 2, 150, 100)
    img, mask, tl = fr[1]
    fr[1] = (img, np.zeros_like(mask), tl)       # a frame that contributes nothing
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, fr, bands=3, check_levels=True)


def test_warp_then_blend_pair(ctx, oracle_mod, small_pair):
    """config-2 style slice: two synthetic frames through warp + multiband blend on both sides."""
    import torch
    import image_stitching_amd as isa
    cams, frames = small_pair
    scale = isa.Stitcher.warped_image_scale(cams)
    warper = isa.SphericalWarper(ctx, scale)
    items, oitems = [], []
    for cam, f in zip(cams, frames):
        K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
        tl, img_s, msk = warper.warp_fused(torch.from_numpy(f).cuda(), K, R)
        items.append((img_s.cpu().numpy(), msk.cpu().numpy(), tl))
        oi, otl = oracle_mod.warp_spherical(f, scale, K, R)
        om, _ = oracle_mod.warp_spherical(np.full(f.shape[:2], 255, np.uint8), scale, K, R, 0, 0)
        assert otl == tl and np.array_equal(items[-1][0], oi.astype(np.int16)) and np.array_equal(items[-1][1], om)
    pano = _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, items, bands=3, check_levels=True)
    assert pano.shape[1] > frames[0].shape[1]


def test_multiband_full_size_properties(ctx):
    """4K-size frames, 8 bands (BASELINE config 3 shape): properties instead of the slow oracle."""
    import torch
    import image_stitching_amd as isa
    h, w = 2160, 3840
    corners, sizes = [(0, 0), (2600, 40)], [(w, h), (w, h)]
    b = isa.MultiBandBlender(ctx, 8)
    b.prepare(corners, sizes)
    full = torch.full((h, w), 255, dtype=torch.uint8, device="cuda")
    for c, val in zip(corners, (80, 160)):
        b.feed(torch.full((h, w, 3), val, dtype=torch.int16, device="cuda"), full, c)
    out, msk = b.blend()
    o = out.cpu().numpy().astype(int)
    m = msk.cpu().numpy()
    assert m[:h, :w].all() and m[40:, 2600:].all() and not m[:40, w:].any()
    # away from the overlap each constant frame is reproduced (minus the per-level truncation loss)
    assert abs(o[1000, 500, 0] - 80) <= 9 and abs(o[1000, 6000, 0] - 160) <= 9
    # inside the overlap the blend is between the two inputs
    ov = o[1000, 2700:3800, 0]
    assert ov.min() >= 70 and ov.max() <= 170


def test_warp_then_blend_4k_pair_bit_exact(ctx, oracle_mod):
    """Two adjacent BASELINE config-3 frames at full size through the fused warp and the multiband blender (the
    band count the job uses for them) against the oracle: every pyramid level and the final panorama."""
    import torch
    import synth
    import image_stitching_amd as isa
    cams = synth.workload("config3")[7:9]
    frames = [synth.render_frame(c) for c in cams]
    scale = isa.Stitcher.warped_image_scale(cams)
    warper = isa.SphericalWarper(ctx, scale)
    items = []
    for cam, f in zip(cams, frames):
        K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
        tl, img_s, msk = warper.warp_fused(torch.from_numpy(f).cuda(), K, R)
        oi, otl = oracle_mod.warp_spherical(f, scale, K, R)
        om, _ = oracle_mod.warp_spherical(np.full(f.shape[:2], 255, np.uint8), scale, K, R, 0, 0)
        got_i, got_m = img_s.cpu().numpy(), msk.cpu().numpy()
        assert otl == tl and np.array_equal(got_i, oi.astype(np.int16)) and np.array_equal(got_m, om)
        items.append((got_i, got_m, tl))
    cs = [i[2] for i in items]; ss = [(i[1].shape[1], i[1].shape[0]) for i in items]
    x0 = min(c[0] for c in cs); y0 = min(c[1] for c in cs)
    x1 = max(c[0] + s[0] for c, s in zip(cs, ss)); y1 = max(c[1] + s[1] for c, s in zip(cs, ss))
    _, bands, _ = oracle_mod.blend_config(oracle_mod.BLEND_MULTI_BAND, 5.0, x1 - x0, y1 - y0)
    pano = _run_both(ctx, oracle_mod, oracle_mod.BLEND_MULTI_BAND, items, bands=bands, check_levels=True)
    assert pano.shape[1] > 3840 and bands >= 6


@pytest.mark.parametrize("n,bands", [(3, 4), (19, 3), (5, 0)])
def test_feed_batch_equals_single_feeds(ctx, oracle_mod, n, bands):
    """mis_blender_feed_batch (the frames' pyramids built together, > FB_MAX frames in two groups, frames of different sizes, one
    with values outside 0..255) leaves every accumulator level and the blend identical to n single feeds -- and to the oracle."""
    import torch
    import image_stitching_amd as isa
    rng = np.random.default_rng(33 + n)
    frames = _frames(rng, n, 150, 110)
    wide = frames[1][0].copy(); wide[20:50, 30:70] = rng.integers(-5000, 5000, (30, 40, 3)).astype(np.int16)
    frames[1] = (wide, frames[1][1], frames[1][2])
    corners = [f[2] for f in frames]
    sizes = [(f[0].shape[1], f[0].shape[0]) for f in frames]
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda(), f[2]) for f in frames]
    single = isa.MultiBandBlender(ctx, bands); single.prepare(corners, sizes)
    for img, mask, tl in dev:
        single.feed(img, mask, tl)
    batch = isa.MultiBandBlender(ctx, bands); batch.prepare(corners, sizes)
    batch.feed_batch([d[0] for d in dev], [d[1] for d in dev], [d[2] for d in dev])
    assert batch.numBands() == single.numBands()
    for l in range(single.numBands() + 1):
        sl, sw = single.level(l)
        bl, bw = batch.level(l)
        assert np.array_equal(bl, sl), "laplacian level %d" % l
        assert np.array_equal(bw.view(np.uint32), sw.view(np.uint32)), "weight level %d" % l
    s_out, s_mask = single.blend()
    b_out, b_mask = batch.blend()
    ctx.synchronize()
    assert torch.equal(b_out, s_out) and torch.equal(b_mask, s_mask)
    ob = oracle_mod.Blender(oracle_mod.BLEND_MULTI_BAND, bands, 0.0)
    ob.prepare(corners, sizes)
    for img, mask, tl in frames:
        ob.feed(img, mask, tl)
    oref, omask = ob.blend()
    assert np.array_equal(b_out.cpu().numpy(), oref) and np.array_equal(b_mask.cpu().numpy(), omask)


def test_feed_batch_edge_cases(ctx, oracle_mod):
    """An empty batch is a no-op; a batch with a frame whose mask is all zero, a one-frame batch after single feeds (the
    accumulators are no longer fresh: read-modify-write mode) and a second batch on top all equal the oracle's sequential feeds."""
    import torch
    import image_stitching_amd as isa
    rng = np.random.default_rng(77)
    frames = _frames(rng, 5, 140, 100)
    frames[2] = (frames[2][0], np.zeros_like(frames[2][1]), frames[2][2])
    corners = [f[2] for f in frames]
    sizes = [(f[0].shape[1], f[0].shape[0]) for f in frames]
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda(), f[2]) for f in frames]
    gb = isa.MultiBandBlender(ctx, 3); gb.prepare(corners, sizes)
    gb.feed_batch([], [], [])
    gb.feed(*dev[0])                                                   # single feed first: the batch that follows must add to it
    gb.feed_batch([dev[1][0]], [dev[1][1]], [dev[1][2]])               # batch of one
    gb.feed_batch([d[0] for d in dev[2:]], [d[1] for d in dev[2:]], [d[2] for d in dev[2:]])   # empty mask inside
    ob = oracle_mod.Blender(oracle_mod.BLEND_MULTI_BAND, 3, 0.0)
    ob.prepare(corners, sizes)
    for img, mask, tl in frames:
        ob.feed(img, mask, tl)
    for l in range(ob.num_bands + 1):
        olap, owgt = ob.level(l)
        glap, gwgt = gb.level(l)
        assert np.array_equal(glap, olap), "laplacian level %d" % l
        assert np.array_equal(gwgt.view(np.uint32), owgt.view(np.uint32)), "weight level %d" % l
    oref, omask = ob.blend()
    gout, gmask = gb.blend()
    ctx.synchronize()
    assert np.array_equal(gout.cpu().numpy(), oref) and np.array_equal(gmask.cpu().numpy(), omask)


def test_feather_wide_ragged_masks_bit_exact(ctx, oracle_mod):
    """K15's scans outside the easy case: rows wider than one 4096-pixel chunk, rows without any zero, zeros only in a later chunk,
    columns whose nearest zero lies several 32-row segments away, an all-zero and an all-255 frame."""
    rng = np.random.default_rng(31)
    w, h = 9100, 150
    img = rng.integers(0, 256, (h, w, 3)).astype(np.int16)
    mask = np.full((h, w), 255, np.uint8)
    mask[0:3, 5000:5003] = 0                  # the only zeros of many rows' columns: far above
    mask[40, 8800] = 0                        # a row whose only zero sits in the third chunk
    mask[90:95, 10:4000] = 0
    mask[120, :] = 0
    mask[rng.integers(0, h, 40), rng.integers(0, w, 40)] = 0
    frames = [(img, mask, (0, 0)),
              (img[:, :300].copy(), np.zeros((h, 300), np.uint8), (50, 3)),
              (img[:, 300:700].copy(), np.full((h, 400), 255, np.uint8), (200, -2))]
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_FEATHER, frames, sharp=1.0 / 37.0, check_levels=False)
    _run_both(ctx, oracle_mod, oracle_mod.BLEND_FEATHER, frames, sharp=0.0004)      # weights below 1 across thousands of pixels


def test_feather_config3_4k_pair_bit_exact(ctx, oracle_mod):
    """FeatherBlender at frame size: two adjacent 3840 x 2160 frames of config 3 through the fused warp and the
    feather blender with the sharpness image_stitching.cpp:1186-1190 gives for the 16-frame panorama: accumulated image and
    weight sums (f32 as bits) after the feeds, then the blended panorama and mask."""
    import torch
    import synth
    import image_stitching_amd as isa
    cams_all = synth.workload("config3")
    w, h = 3840, 2160
    scale = isa.Stitcher.warped_image_scale(cams_all)
    rois_all = isa.stitching.warp_rois(ctx, scale, (w, h), cams_all)
    x0 = min(r[0] for r in rois_all); y0 = min(r[1] for r in rois_all)
    x1 = max(r[0] + r[2] for r in rois_all); y1 = max(r[1] + r[3] for r in rois_all)
    btype, _, sharp = oracle_mod.blend_config(oracle_mod.BLEND_FEATHER, 5.0, x1 - x0, y1 - y0)
    assert btype == oracle_mod.BLEND_FEATHER and 0 < sharp < 0.01
    pair = [7, 8]
    cams = [cams_all[i] for i in pair]
    dev = [synth.render_frame_gpu(c) for c in cams]
    torch.cuda.synchronize()
    warper = isa.SphericalWarper(ctx, scale)
    batch = warper.warp_fused_batch(dev, cams, [rois_all[i] for i in pair])
    gb, ob = isa.FeatherBlender(ctx, sharp), oracle_mod.Blender(oracle_mod.BLEND_FEATHER, 0, sharp)
    corners = [b[0] for b in batch]
    sizes = [(b[2].shape[1], b[2].shape[0]) for b in batch]
    gb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    for cam, fd, (tl, img_s, msk) in zip(cams, dev, batch):
        K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
        oi, otl = oracle_mod.warp_spherical(fd.cpu().numpy(), scale, K, R)
        om, _ = oracle_mod.warp_spherical(np.full((h, w), 255, np.uint8), scale, K, R, 0, 0)
        assert otl == tl and np.array_equal(img_s.cpu().numpy(), oi.astype(np.int16)) and np.array_equal(msk.cpu().numpy(), om)
        gb.feed(img_s, msk, tl)
        ob.feed(oi.astype(np.int16), om, tl)
    gl, gw = gb.level(0)
    ol, ow = ob.level(0)
    assert np.array_equal(gw.view(np.uint32), ow.view(np.uint32)), "feather weight sums"
    assert np.array_equal(gl, ol), "feather accumulated image"
    gp, gm = gb.blend()
    op, om_ = ob.blend()
    assert np.array_equal(gm.cpu().numpy(), om_) and np.array_equal(gp.cpu().numpy(), op)
