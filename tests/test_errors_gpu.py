"""GPU: error behaviour of the C ABI on the hot path -- a bad call returns a negative status with a message (MisError here), does
nothing, and leaves the object usable: the next correct call gives the result of a run without the bad call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_INVALID, E_STATE, E_UNSUPPORTED = -1, -5, -6


def _cam_frames(n, w=320, h=180):
    import torch
    import synth
    cams = [synth.make_camera(w, h, 60.0, 14.0 * i - 10.0, 0.3 * (i - 1), 0.0) for i in range(n)]
    return cams, [torch.from_numpy(synth.render_frame(c)).cuda() for c in cams]


def test_blender_call_order_and_argument_errors(ctx):
    import torch
    import image_stitching_amd as isa
    cams, frames = _cam_frames(2)
    scale = isa.Stitcher.warped_image_scale(cams)
    w = isa.SphericalWarper(ctx, scale)
    warped = [w.warp_fused(f, c["K"], c["R"]) for f, c in zip(frames, cams)]
    corners = [t[0] for t in warped]
    sizes = [(t[1].shape[1], t[1].shape[0]) for t in warped]
    b = isa.MultiBandBlender(ctx, 3)
    with pytest.raises(isa.MisError) as e:
        b.feed(warped[0][1], warped[0][2], warped[0][0])          # feed before prepare
    assert e.value.code == E_STATE
    with pytest.raises(isa.MisError) as e:
        b.blend()                                                 # blend before prepare
    assert e.value.code == E_STATE
    b.prepare(corners, sizes)
    with pytest.raises(isa.MisError) as e:                        # an 8-bit image where 16SC3 is required
        b.feed(warped[0][1].to(torch.uint8), warped[0][2], warped[0][0])
    assert e.value.code == E_INVALID
    with pytest.raises(isa.MisError) as e:                        # mask of another size
        b.feed(warped[0][1], warped[0][2][:-1], warped[0][0])
    assert e.value.code == E_INVALID
    with pytest.raises(isa.MisError) as e:                        # a corner outside the prepared roi
        b.feed(warped[0][1], warped[0][2], (corners[0][0] - 10000, corners[0][1]))
    assert e.value.code == E_INVALID
    with pytest.raises(isa.MisError) as e:                        # one bad frame fails the whole batch before anything is fed
        b.feed_batch([warped[0][1], warped[1][1]], [warped[0][2], warped[1][2][:-1]], corners)
    assert e.value.code == E_INVALID
    for tl, img, msk in warped:                                   # the blender is as it was after prepare
        b.feed(img, msk, tl)
    got, gmask = b.blend()
    ref = isa.MultiBandBlender(ctx, 3); ref.prepare(corners, sizes)
    for tl, img, msk in warped:
        ref.feed(img, msk, tl)
    want, wmask = ref.blend()
    ctx.synchronize()
    assert torch.equal(got, want) and torch.equal(gmask, wmask)


def test_warp_and_feature_argument_errors(ctx):
    import torch
    import image_stitching_amd as isa
    cams, frames = _cam_frames(2)
    scale = isa.Stitcher.warped_image_scale(cams)
    w = isa.SphericalWarper(ctx, scale)
    with pytest.raises(isa.MisError) as e:                        # the fused warp needs a colour frame
        w.warp_fused(frames[0][:, :, 0].contiguous(), cams[0]["K"], cams[0]["R"])
    assert e.value.code in (E_UNSUPPORTED, E_INVALID)
    with pytest.raises(isa.MisError) as e:
        isa.SphericalWarper(ctx, -1.0).warp_fused(frames[0], cams[0]["K"], cams[0]["R"])
    assert e.value.code == E_INVALID
    good = w.warp_fused(frames[0], cams[0]["K"], cams[0]["R"])
    again = w.warp_fused(frames[0], cams[0]["K"], cams[0]["R"])
    assert torch.equal(good[1], again[1]) and torch.equal(good[2], again[2])
    finder = isa.OrbFeatureFinder(ctx, (320, 180))
    with pytest.raises(isa.MisError) as e:                        # frames of a batch share one size
        finder.detect_batch([frames[0], frames[1][:-4].contiguous()])
    assert e.value.code == E_INVALID
    with pytest.raises(isa.MisError) as e:                        # larger than the finder was created for
        finder.detect(torch.zeros((400, 700, 3), dtype=torch.uint8, device="cuda"))
    assert e.value.code in (E_INVALID, E_UNSUPPORTED)
    a = finder.detect_batch(frames)
    b = finder.detect_batch(frames)
    for x, y in zip(a, b):
        kx, dx = x.download(); ky, dy = y.download()
        assert np.array_equal(kx, ky) and np.array_equal(dx, dy)


def test_matcher_refuses_mixed_descriptors_and_recovers(ctx):
    import torch
    import image_stitching_amd as isa
    cams, frames = _cam_frames(2)
    orb = isa.OrbFeatureFinder(ctx, (320, 180))
    sift = isa.SiftFeatureFinder(ctx, (320, 180))
    fo = [isa.computeImageFeatures(orb, f, i) for i, f in enumerate(frames)]
    fs = isa.computeImageFeatures(sift, frames[1], 1)
    m = isa.BestOf2NearestMatcher(ctx, 0.32)
    with pytest.raises(isa.MisError) as e:
        m([fo[0], fs])                                            # binary and float descriptors in one call
    assert e.value.code == E_INVALID
    a, b = m(fo), m(fo)
    for x, y in zip(a, b):
        assert np.array_equal(x.matches, y.matches) and x.confidence == y.confidence


def test_batched_warp_error_hands_back_what_it_allocated(ctx):
    """mis_warp_spherical_fused_batch with library-allocated outputs (data == NULL) and a bad second frame: the call fails with a
    code, frame 0's freshly allocated outputs are released and the caller's structs are back to data == NULL; the same call with
    good frames then works."""
    import ctypes as C
    import torch
    import image_stitching_amd as isa
    from image_stitching_amd import _capi as capi
    from image_stitching_amd.stitching import as_image
    cams, frames = _cam_frames(2)
    scale = isa.Stitcher.warped_image_scale(cams)
    rois = isa.stitching.warp_rois(ctx, scale, (frames[0].shape[1], frames[0].shape[0]), cams)
    Ks = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float32).reshape(9) for c in cams]))
    Rs = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float32).reshape(9) for c in cams]))
    rs = (capi.MisRect * 2)(*[capi.MisRect(int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in rois])
    fp = C.POINTER(C.c_float)

    def call(imgs):
        im = (capi.MisImage * 2)(*[as_image(i) for i in imgs])
        ds, ms, tls = (capi.MisImage * 2)(), (capi.MisImage * 2)(), (capi.MisPoint * 2)()
        rc = ctx.lib.mis_warp_spherical_fused_batch(ctx.h, im, 2, float(scale), Ks.ctypes.data_as(fp), Rs.ctypes.data_as(fp), rs, ds, ms, tls)
        return rc, ds, ms
    gray = frames[1][:, :, 0].contiguous()
    rc, ds, ms = call([frames[0], gray])
    assert rc in (E_UNSUPPORTED, E_INVALID)
    assert not ds[0].data and not ms[0].data and not ds[1].data and not ms[1].data
    rc, ds, ms = call(frames)
    assert rc == 0 and ds[0].data and ms[1].data
    for k in range(2):
        ctx.lib.mis_image_free(ctx.h, C.byref(ds[k])); ctx.lib.mis_image_free(ctx.h, C.byref(ms[k]))
