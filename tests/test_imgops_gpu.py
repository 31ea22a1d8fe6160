"""GPU parity of the image operators either side of the path (rows a14, N1c of SURVEY section 8) against the
oracle: bit-exact (byte arithmetic).  Reference call sites: image_stitching.cpp:571-580, :619, :1144, :1169-1171."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _img(h, w, c, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, c) if c > 1 else (h, w), dtype=np.uint8)
    return a


@pytest.mark.parametrize("h,w,c,fx,fy", [(270, 480, 3, 0.6, 0.6), (271, 483, 3, 0.2581988897, 0.2581988897), (133, 77, 1, 0.5, 0.5),
                                          (64, 64, 3, 1.0, 1.0), (50, 90, 1, 1.7, 2.3), (1080, 1920, 3, 0.3162277660, 0.3162277660)])
def test_resize_by_factor_bit_exact(ctx, oracle_mod, h, w, c, fx, fy):
    import torch
    import image_stitching_amd as isa
    a = _img(h, w, c, h + w)
    want = oracle_mod.resize_exact(a, fx=fx, fy=fy)
    got = isa.resize(ctx, torch.from_numpy(a).cuda(), fx=fx, fy=fy).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("h,w,c,dsize", [(100, 150, 1, (611, 407)), (270, 480, 3, (123, 45)), (31, 17, 1, (17, 31)), (8, 8, 3, (1, 1))])
def test_resize_to_size_bit_exact_device_and_host_buffers(ctx, oracle_mod, h, w, c, dsize):
    import torch
    import ctypes as C
    import image_stitching_amd as isa
    from image_stitching_amd import _capi as capi
    from image_stitching_amd.stitching import as_image
    a = _img(h, w, c, 3 * h + w)
    want = oracle_mod.resize_exact(a, dsize=dsize)
    got = isa.resize(ctx, torch.from_numpy(a).cuda(), dsize=dsize).cpu().numpy()
    assert np.array_equal(got, want)
    # host buffers through the same C entry (the boundary accepts both)
    out = np.zeros_like(want)
    simg, dimg = as_image(a), as_image(out)
    ctx.check(ctx.lib.mis_resize_linear_exact(ctx.h, C.byref(simg), dsize[0], dsize[1], 0.0, 0.0, C.byref(dimg)))
    assert np.array_equal(out, want)


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("code", [0, 1, 2])
def test_rotate_bit_exact(ctx, oracle_mod, c, code):
    import torch
    import image_stitching_amd as isa
    a = _img(97, 161, c, 10 * c + code)
    got = isa.rotate(ctx, torch.from_numpy(a).cuda(), code).cpu().numpy()
    assert np.array_equal(got, oracle_mod.rotate(a, code))


def test_seam_mask_apply_bit_exact(ctx, oracle_mod):
    import torch
    import image_stitching_amd as isa
    rng = np.random.default_rng(11)
    # seam-scale mask (0.1 MP class) with a ragged valid region and holes; compose-size mask of a warped frame
    seam = np.zeros((217, 331), np.uint8)
    seam[20:190, 15:300] = 255
    seam[rng.integers(0, 217, 400), rng.integers(0, 331, 400)] = 0
    seam[100:110, 150:170] = 0
    mask = np.full((1303, 1987), 255, np.uint8)
    mask[:, :40] = 0
    mask[rng.integers(0, 1303, 1000), rng.integers(0, 1987, 1000)] = 0
    want = oracle_mod.seam_mask_apply(seam, mask)
    m = torch.from_numpy(mask).cuda()
    isa.seam_mask_apply(ctx, torch.from_numpy(seam).cuda(), m)
    assert np.array_equal(m.cpu().numpy(), want)
    assert 0 < int((want == 255).sum()) < want.size                     # neither trivial outcome
    # host in/out buffer
    mh = mask.copy()
    isa.seam_mask_apply(ctx, seam, mh)
    assert np.array_equal(mh, want)


def test_resize_rejects_bad_arguments(ctx):
    import torch
    import image_stitching_amd as isa
    a = torch.zeros((8, 8, 2), dtype=torch.uint8, device="cuda")
    with pytest.raises(isa.MisError):
        isa.resize(ctx, a, dsize=(4, 4))
    with pytest.raises(isa.MisError):
        isa.rotate(ctx, torch.zeros((8, 8), dtype=torch.uint8, device="cuda"), 7)
