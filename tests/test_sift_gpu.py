"""GPU parity of the SIFT detector / descriptor (image_stitching.cpp:559, :613) against the oracle.
Every stage is compared bit for bit: the oracle and the HIP kernels evaluate the same float expressions in the same
order (no FMA contraction, shared exp / atan2 / sin / cos polynomials), and the keypoint order is the total order
of KeyPointsFilter::removeDuplicatedSorted."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(w, h, yaw=0.0, seed_pitch=0.0):
    import synth
    return synth.render_frame(synth.make_camera(w, h, 60.0, yaw, seed_pitch))


@pytest.mark.parametrize("w,h", [(320, 200), (333, 251), (1300, 72)])   # the last: rows wider than one 2048-output block of the row pass
def test_scale_space_bit_exact(ctx, oracle_mod, w, h):
    import torch
    import image_stitching_amd as isa
    fr = _frame(w, h, 5.0)
    o = oracle_mod.Sift(w, h)
    o.run(fr)
    f = isa.SiftFeatureFinder(ctx, (w, h))
    img = torch.from_numpy(fr).cuda()
    nl = 3
    for octave in range(o.num_octaves()):
        for layer in range(nl + 3):
            assert np.array_equal(f.debug_level(img, octave, layer, dog=False).view(np.uint32), o.gauss(octave, layer).view(np.uint32)), (octave, layer)
        for layer in range(nl + 2):
            assert np.array_equal(f.debug_level(img, octave, layer, dog=True).view(np.uint32), o.dog(octave, layer).view(np.uint32)), (octave, layer)


@pytest.mark.parametrize("w,h,yaw", [(480, 270, 0.0), (333, 251, 40.0), (640, 360, -75.0), (1300, 72, 10.0)])
def test_keypoints_and_descriptors_bit_exact(ctx, oracle_mod, w, h, yaw):
    import torch
    import image_stitching_amd as isa
    fr = _frame(w, h, yaw, 3.0)
    ko, do = oracle_mod.Sift(w, h).run(fr)
    f = isa.SiftFeatureFinder(ctx, (w, h))
    kg, dg = f.detect(torch.from_numpy(fr).cuda()).download()
    assert len(ko) > 200
    assert len(kg) == len(ko)
    for name in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kg[name].view(np.uint32), ko[name].view(np.uint32)), name
    assert np.array_equal(kg["octave"], ko["octave"])
    assert dg.dtype == np.float32 and dg.shape == (len(ko), 128)
    assert np.array_equal(dg, do)
    # descriptor invariants of calcSIFTDescriptor: integers 0..255, norm close to 512
    assert np.array_equal(dg, np.rint(dg)) and dg.min() >= 0 and dg.max() <= 255
    nrm = np.linalg.norm(dg, axis=1)
    assert np.all(np.abs(nrm - 512) < 24)


def test_flat_and_tiny_images(ctx, oracle_mod):
    import torch
    import image_stitching_amd as isa
    f = isa.SiftFeatureFinder(ctx, (64, 48))
    flat = np.full((48, 64, 3), 90, np.uint8)
    assert len(f.detect(torch.from_numpy(flat).cuda())) == 0
    assert len(oracle_mod.Sift(64, 48).run(flat)[0]) == 0
    # a smaller image through a finder planned for a larger one
    g = isa.SiftFeatureFinder(ctx, (200, 160))
    fr = _frame(96, 80, 10.0)
    ko, do = oracle_mod.Sift(96, 80).run(fr)
    kg, dg = g.detect(torch.from_numpy(fr).cuda()).download()
    assert len(kg) == len(ko) and np.array_equal(dg, do)


def test_sift_features_feed_the_l2_matcher(ctx, oracle_mod):
    """Two overlapping views: SIFT -> exact L2 2-NN on MFMA (K8) -> RANSAC; the matcher output equals the oracle's on
    the oracle's features (config-5 style path end to end at a small size)."""
    import torch
    import synth
    import image_stitching_amd as isa
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 0.0), synth.make_camera(w, h, 60.0, 12.0)]
    frames = [synth.render_frame(c) for c in cams]
    f = isa.SiftFeatureFinder(ctx, (w, h))
    feats = [f.detect(torch.from_numpy(fr).cuda()) for fr in frames]
    for i, ft in enumerate(feats):
        ft.img_idx = i
    pm = isa.BestOf2NearestMatcher(ctx, 0.65)(feats)          # match_conf of the float-descriptor branch (:59)
    so = oracle_mod.Sift(w, h)
    of = []
    for fr in frames:
        k, d = so.run(fr)
        of.append(dict(img_w=w, img_h=h, xy=np.stack([k["x"], k["y"]], 1), desc=d))
    ref = oracle_mod.match_pair(of[0], of[1], oracle_mod.match_default_params(match_conf=0.65))
    m = pm[1]
    assert m.num_inliers == ref["num_inliers"] and m.num_inliers >= 10
    assert np.array_equal(np.asarray(m.H, np.float64).view(np.uint64).reshape(-1), np.asarray(ref["H"], np.float64).view(np.uint64).reshape(-1))


def test_batch_of_frames_equals_single_calls(ctx):
    """mis_sift_detect_batch keeps several frames in flight (three lanes by default; every lane: own scale space, stream and host
    thread, output blocks from the caller's pool); results and their order must be those of one call per frame.  Device and host inputs."""
    import torch
    import image_stitching_amd as isa
    w, h = 480, 270
    frames = [_frame(w, h, yaw, 1.0) for yaw in (0.0, 20.0, 40.0, 60.0, 80.0)]
    f = isa.SiftFeatureFinder(ctx, (w, h))
    singles = [f.detect(torch.from_numpy(fr).cuda()).download() for fr in frames]
    for inputs in ([torch.from_numpy(fr).cuda() for fr in frames], frames):
        batch = f.detect_batch(inputs)
        assert [b.img_idx for b in batch] == list(range(len(frames)))
        for (ks, ds), b in zip(singles, batch):
            kb, db = b.download()
            assert len(kb) == len(ks) and len(ks) > 100
            assert kb.tobytes() == ks.tobytes() and np.array_equal(db, ds)
    assert f.detect_batch([]) == []


_LANES_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import synth, image_stitching_amd as isa
ctx = isa.Context(0)
w, h = 640, 360
frames = [synth.render_frame_gpu(synth.make_camera(w, h, 60.0, 10.0 * k)) for k in range(7)]
f = isa.SiftFeatureFinder(ctx, (w, h))
out = {}
for rep in range(2):      # the second batch reuses the recycled output blocks of the first
    batch = f.detect_batch(frames)
    for i, b in enumerate(batch):
        k, d = b.download()
        out["k%d_%d" % (rep, i)] = np.frombuffer(k.tobytes(), np.uint8); out["d%d_%d" % (rep, i)] = d
    del batch
np.savez(sys.argv[2], **out)
"""


def test_batch_lanes_one_to_four_give_identical_features(tmp_path):
    """MIS_SIFT_LANES = 1, 3 (the default) and 4: seven frames, two batches each (the second on recycled output blocks): every
    keypoint and descriptor byte for byte.  The switch is read once per process, hence child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for lanes in ("1", "3", "4"):
        path = str(tmp_path / ("lanes%s.npz" % lanes))
        r = subprocess.run([sys.executable, "-c", _LANES_SCRIPT, root, path], env=dict(os.environ, MIS_SIFT_LANES=lanes), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(path))
    a = outs[0]
    assert len(a.files) == 28 and all(a["k0_%d" % i].size > 24 * 50 for i in range(7))
    for b in outs[1:]:
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k
    for i in range(7):
        assert a["k0_%d" % i].tobytes() == a["k1_%d" % i].tobytes() and a["d0_%d" % i].tobytes() == a["d1_%d" % i].tobytes()
