"""GPU parity: ORB detect + describe (HIP, through the C ABI) vs the CPU oracle, bit-exact at every stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_frame(ctx, oracle_mod, finder, orb, frame, stages=True):
    import torch
    feats = finder.detect(torch.from_numpy(frame).cuda())
    okps, odesc = orb.run(frame)
    if stages:
        for l in range(orb.params.nlevels):
            assert np.array_equal(finder.debug_level(l, 0), orb.level_gray(l)), "gray level %d" % l
            assert np.array_equal(finder.debug_level(l, 1), orb.level_nms(l)), "nms level %d" % l
            assert np.array_equal(finder.debug_level(l, 2), orb.level_blur(l)), "blur level %d" % l
    kps, desc = feats.download()
    assert len(kps) == len(okps)
    for f in ("octave", "x", "y", "size", "response", "angle"):
        assert np.array_equal(kps[f], okps[f]), f
    assert np.array_equal(desc, odesc)
    assert feats.img_size == (frame.shape[1], frame.shape[0])
    return kps, desc


def test_orb_small_pair_bit_exact(ctx, oracle_mod, small_pair):
    import image_stitching_amd as isa
    cams, frames = small_pair
    h, w = frames[0].shape[:2]
    finder = isa.OrbFeatureFinder(ctx, (w, h))
    orb = oracle_mod.Orb(w, h)
    for f in frames:
        kps, desc = _check_frame(ctx, oracle_mod, finder, orb, f)
        assert 3000 < len(kps) <= 4000


def test_orb_few_corners_and_flat(ctx, oracle_mod):
    """Fewer corners than the budget (retainBest keeps all) and a flat image (no keypoints)."""
    import torch
    import image_stitching_amd as isa
    w, h = 200, 150
    finder = isa.OrbFeatureFinder(ctx, (w, h))
    orb = oracle_mod.Orb(w, h)
    img = np.full((h, w, 3), 90, np.uint8)
    img[40:80, 50:120] = (200, 30, 60)
    img[100:120, 20:40] = (10, 220, 130)
    kps, _ = _check_frame(ctx, oracle_mod, finder, orb, img)
    assert 0 < len(kps) < 200
    flat = np.full((h, w, 3), 128, np.uint8)
    kps, desc = _check_frame(ctx, oracle_mod, finder, orb, flat)
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_orb_resizes_workspace_and_other_params(ctx, oracle_mod):
    """A finder created for a larger size handles smaller frames; non-default budget / levels."""
    import synth
    import image_stitching_amd as isa
    finder = isa.OrbFeatureFinder(ctx, (640, 480), isa.stitching.orb_params(nfeatures=700, nlevels=5, fast_threshold=30))
    for (w, h) in ((400, 300), (333, 257)):
        frame = synth.render_frame(synth.make_camera(w, h, 60.0, 40.0))
        orb = oracle_mod.Orb(w, h, oracle_mod.orb_default_params(nfeatures=700, nlevels=5, fast_threshold=30))
        _check_frame(ctx, oracle_mod, finder, orb, frame)


def test_orb_1080p_and_batch(ctx, oracle_mod):
    """BASELINE config 2 frame size; the batched entry point gives the same features."""
    import torch
    import synth
    import image_stitching_amd as isa
    cams = synth.workload("config2")
    frames = [synth.render_frame(c) for c in cams]
    finder = isa.OrbFeatureFinder(ctx, (1920, 1080))
    orb = oracle_mod.Orb(1920, 1080)
    single = [_check_frame(ctx, oracle_mod, finder, orb, f, stages=(i == 0)) for i, f in enumerate(frames)]
    batch = finder.detect_batch([torch.from_numpy(f).cuda() for f in frames])
    for (kps, desc), fb in zip(single, batch):
        bk, bd = fb.download()
        assert np.array_equal(bk, kps) and np.array_equal(bd, desc)


def test_orb_4k_properties(ctx):
    """4K frame (config 3 size): budget respected, canonical order, determinism across runs."""
    import synth
    import image_stitching_amd as isa
    cam = synth.workload("config3")[5]
    frame = synth.render_frame_gpu(cam)
    finder = isa.OrbFeatureFinder(ctx, (3840, 2160))
    k1, d1 = finder.detect(frame).download()
    k2, d2 = finder.detect(frame).download()
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    assert len(k1) == 4000
    budget = [869, 724, 603, 503, 419, 349, 291, 242]
    for l in range(8):
        sel = k1[k1["octave"] == l]
        assert len(sel) == budget[l]
        r = sel["response"]
        assert np.all(r[:-1] >= r[1:])                      # response descending inside a level
        assert np.all((sel["angle"] >= 0) & (sel["angle"] < 360))
    assert np.all(np.diff(k1["octave"]) >= 0)
    assert k1["x"].min() >= 0 and k1["x"].max() < 3840 and k1["y"].max() < 2160


def test_orb_4k_frame_bit_exact(ctx, oracle_mod):
    """One BASELINE config-3 frame at full size against the oracle (all pyramid levels, keypoints, descriptors)."""
    import synth
    import image_stitching_amd as isa
    cam = synth.workload("config3")[7]
    frame = synth.render_frame(cam)
    finder = isa.OrbFeatureFinder(ctx, (3840, 2160))
    kps, desc = _check_frame(ctx, oracle_mod, finder, oracle_mod.Orb(3840, 2160), frame)
    assert len(kps) == 4000


def test_orb_on_enqueued_hook_runs_once_inside_the_next_batch(ctx, small_pair):
    """mis_orb_on_enqueued: the hook runs once, on the calling thread, inside the NEXT detect_batch (the job sizes its blender there);
    a cleared hook does not run, the hook does not change the features, an exception in it surfaces after the call."""
    import threading
    import torch
    import image_stitching_amd as isa
    cams, frames = small_pair
    h, w = frames[0].shape[:2]
    finder = isa.OrbFeatureFinder(ctx, (w, h))
    imgs = [torch.from_numpy(f).cuda() for f in frames]
    plain = [f.download() for f in finder.detect_batch(imgs)]
    ran = []
    finder.on_enqueued(lambda: ran.append(threading.get_ident()))
    hooked = [f.download() for f in finder.detect_batch(imgs)]
    assert ran == [threading.get_ident()]
    for (k0, d0), (k1, d1) in zip(plain, hooked):
        assert np.array_equal(d0, d1) and all(np.array_equal(k0[f], k1[f]) for f in ("x", "y", "angle", "response", "octave"))
    finder.detect_batch(imgs)                      # one-shot: not again
    assert len(ran) == 1
    finder.on_enqueued(lambda: ran.append(-1))
    finder.on_enqueued(None)                       # cleared before the call
    finder.detect_batch(imgs)
    assert len(ran) == 1
