"""GPU parity: exact 2-NN, ratio/union, RANSAC + LM homography, all-pairs matcher vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _feat_dict(kps, desc, size):
    return dict(img_w=size[0], img_h=size[1], xy=np.stack([kps["x"], kps["y"]], 1), desc=desc)


def test_knn2_hamming_bit_exact(ctx, oracle_mod):
    import image_stitching_amd as isa
    from image_stitching_amd.stitching import KP_DTYPE
    import ctypes as C
    rng = np.random.default_rng(31)
    for nq, nt in ((700, 1300), (257, 2), (5, 300)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nt > 260:
            t[7] = q[3]; t[259] = q[3]          # duplicates across LDS tiles: tie -> smaller index
            t[100] = t[101]
        fq = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(nq, KP_DTYPE), q)
        ft = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(nt, KP_DTYPE), t)
        idx = np.zeros((nq, 2), np.int32)
        dist = np.zeros((nq, 2), np.float32)
        ctx.check(ctx.lib.mis_knn2(ctx.h, C.byref(fq.raw), C.byref(ft.raw), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
        oi, od = oracle_mod.knn2_hamming(q, t)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od.astype(np.float32))


def _synthetic_correspondences(rng, n, outlier_frac, noise=0.4):
    H = np.array([[0.97, 0.03, 25.0], [-0.02, 1.03, -14.0], [2e-5, -1e-5, 1.0]])
    src = rng.uniform(-900, 900, (n, 2)).astype(np.float32)
    p = np.c_[src, np.ones(n)] @ H.T
    dst = (p[:, :2] / p[:, 2:] + rng.normal(0, noise, (n, 2))).astype(np.float32)
    no = int(n * outlier_frac)
    dst[:no] = rng.uniform(-900, 900, (no, 2)).astype(np.float32)
    return src, dst


@pytest.mark.parametrize("n,frac", [(600, 0.1), (900, 0.5), (300, 0.75), (40, 0.3), (2500, 0.2), (5, 0.0), (4, 0.0), (3, 0.0)])
def test_find_homography_bit_exact(ctx, oracle_mod, n, frac):
    import image_stitching_amd as isa
    rng = np.random.default_rng(n * 7 + int(frac * 100))
    src, dst = _synthetic_correspondences(rng, n, frac)
    ok_o, H_o, mask_o, iters_o = oracle_mod.find_homography_ransac(src, dst)
    ok_g, H_g, mask_g = isa.find_homography(ctx, src, dst)
    assert ok_g == ok_o
    assert np.array_equal(mask_g, mask_o)
    if ok_o:
        assert np.array_equal(_bits(H_g), _bits(H_o)), (H_g - H_o)


def test_find_homography_degenerate_inputs(ctx, oracle_mod):
    import image_stitching_amd as isa
    rng = np.random.default_rng(77)
    # all points collinear: checkSubset rejects every subset -> no model
    x = rng.uniform(-100, 100, 50).astype(np.float32)
    src = np.stack([x, 2 * x], 1).astype(np.float32)
    dst = src + 1
    ok_o, H_o, mask_o, _ = oracle_mod.find_homography_ransac(src, dst, max_iters=50)
    ok_g, H_g, mask_g = isa.find_homography(ctx, src, dst, max_iters=50)
    assert ok_g == ok_o and np.array_equal(mask_g, mask_o)
    # pure noise: RANSAC runs all iterations
    src = rng.uniform(-500, 500, (200, 2)).astype(np.float32)
    dst = rng.uniform(-500, 500, (200, 2)).astype(np.float32)
    ok_o, H_o, mask_o, it = oracle_mod.find_homography_ransac(src, dst, max_iters=300)
    ok_g, H_g, mask_g = isa.find_homography(ctx, src, dst, max_iters=300)
    assert ok_g == ok_o and np.array_equal(mask_g, mask_o)
    if ok_o:
        assert np.array_equal(_bits(H_g), _bits(H_o))


def _compare_matches_info(g, o):
    assert g.src_img_idx == o["src_img_idx"] and g.dst_img_idx == o["dst_img_idx"]
    assert np.array_equal(g.matches, o["matches"].astype(g.matches.dtype))
    assert np.array_equal(g.inliers_mask, o["inliers_mask"])
    assert g.num_inliers == o["num_inliers"]
    assert (g.H is not None) == o["has_H"]
    if o["has_H"]:
        assert np.array_equal(_bits(g.H), _bits(o["H"]))
    assert g.confidence == o["confidence"]


def test_match_all_pairs_bit_exact(ctx, oracle_mod):
    """4 synthetic frames (three overlapping, one looking elsewhere): every MatchesInfo field."""
    import torch
    import synth
    import image_stitching_amd as isa
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, y, p, r) for y, p, r in ((0, 0, 0), (28, 0.6, -0.3), (50, -0.4, 0.5), (150, 0, 0))]
    frames = [synth.render_frame(c) for c in cams]
    finder = isa.OrbFeatureFinder(ctx, (w, h))
    feats = [isa.computeImageFeatures(finder, torch.from_numpy(f).cuda(), i) for i, f in enumerate(frames)]
    host = [f.download() for f in feats]
    pm = isa.BestOf2NearestMatcher(ctx, 0.32)(feats)
    ref = oracle_mod.match_all_pairs([_feat_dict(k, d, (w, h)) for k, d in host])
    assert len(pm) == 16
    for g, o in zip(pm, ref):
        _compare_matches_info(g, o)
    # pairs (0,1), (0,2) connect the frames; (1,2) matches so well that the reference's `confidence > 3 -> 0` rule zeroes it
    assert pm[1].confidence > 1.0 and pm[2].confidence > 1.0 and pm[0 * 4 + 3].confidence < 0.95
    idx_g = isa.leaveBiggestComponent(pm, 4, 0.95)
    conf = np.array([m["confidence"] for m in ref]).reshape(4, 4)
    assert list(idx_g) == list(oracle_mod.leave_biggest_component(conf, 0.95)) == [0, 1, 2]
    # sharded matcher: the union over ranks equals the single-rank result
    parts = [isa.BestOf2NearestMatcher(ctx, 0.32)(feats, rank=r, world_size=3) for r in range(3)]
    for k in range(16):
        owners = [p[k] for p in parts if p[k].src_img_idx >= 0]
        if pm[k].src_img_idx < 0:
            assert not owners
        else:
            assert len(owners) == 1
            assert np.array_equal(owners[0].matches, pm[k].matches) and owners[0].confidence == pm[k].confidence


def test_match_empty_and_tiny_feature_sets(ctx, oracle_mod):
    import image_stitching_amd as isa
    from image_stitching_amd.stitching import KP_DTYPE
    rng = np.random.default_rng(41)

    def mk(n):
        k = np.zeros(n, KP_DTYPE)
        k["x"] = rng.uniform(0, 200, n)
        k["y"] = rng.uniform(0, 100, n)
        return k, rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sets = [mk(0), mk(1), mk(30), mk(200)]
    sets[3][1][:30] = sets[2][1]          # 30 identical descriptors -> matches but random geometry
    feats = [isa.ImageFeatures.upload(ctx, (200, 100), k, d, i) for i, (k, d) in enumerate(sets)]
    pm = isa.BestOf2NearestMatcher(ctx, 0.32)(feats)
    ref = oracle_mod.match_all_pairs([_feat_dict(k, d, (200, 100)) for k, d in sets])
    for g, o in zip(pm, ref):
        _compare_matches_info(g, o)


def test_match_full_size_sets_repeatable_and_exact(ctx, oracle_mod):
    """8 frames x 4096 descriptors (128 train tiles per pass, every compute unit holding several workgroups of the matrix pass):
    three runs give identical lists, and sampled pairs equal the oracle's.  Frame i holds a shared base set with a few bits flipped
    per descriptor, so the ratio test passes and the lists are long; the first rows (train tile 0) are part of every list."""
    import image_stitching_amd as isa
    from image_stitching_amd.stitching import KP_DTYPE
    rng = np.random.default_rng(97)
    n, nf = 4096, 8
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sets = []
    for i in range(nf):
        d = base.copy()
        flips = rng.integers(0, 256, (n, 6))              # up to 6 flipped bits per descriptor
        for c in range(6):
            d[np.arange(n), flips[:, c] >> 3] ^= (1 << (flips[:, c] & 7)).astype(np.uint8)
        d = d[rng.permutation(n)] if i else d
        k = np.zeros(n, KP_DTYPE)
        k["x"] = rng.uniform(0, 3840, n)
        k["y"] = rng.uniform(0, 2160, n)
        sets.append((k, d))
    feats = [isa.ImageFeatures.upload(ctx, (3840, 2160), k, d, i) for i, (k, d) in enumerate(sets)]
    runs = [isa.BestOf2NearestMatcher(ctx, 0.32)(feats) for _ in range(3)]
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.array_equal(a.matches, b.matches) and a.confidence == b.confidence
    for i, j in ((0, 1), (0, 7), (2, 5), (3, 4), (6, 7), (1, 6)):
        o = oracle_mod.match_pair(_feat_dict(*sets[i], (3840, 2160)), _feat_dict(*sets[j], (3840, 2160)))
        g = runs[0][i * nf + j]
        assert len(g.matches) > 2000
        assert np.array_equal(g.matches, o["matches"].astype(g.matches.dtype))
        assert np.array_equal(g.inliers_mask, o["inliers_mask"]) and g.confidence == o["confidence"]


def _sift_like(rng, n):
    """SIFT-style descriptors: integer valued 0..255 stored as f32 (SURVEY config 5 surrogate)."""
    d = rng.gamma(0.6, 30.0, (n, 128))
    return np.clip(np.rint(d), 0, 255).astype(np.float32)


def test_knn2_l2_mfma_bit_exact(ctx, oracle_mod):
    import ctypes as C
    import image_stitching_amd as isa
    from image_stitching_amd.stitching import KP_DTYPE
    rng = np.random.default_rng(51)
    # the last two: 8 and 16 train slices (the XCD-mapped orders of the kernel), ragged last tile, duplicates in different slices
    for nq, nt in ((700, 1300), (65, 33), (31, 2), (4096, 4096), (70, 9001), (300, 16411)):
        q, t = _sift_like(rng, nq), _sift_like(rng, nt)
        if nt > 40:
            t[7] = q[3]; t[33] = q[3]      # exact duplicates in different lanes / tiles: tie -> smaller index
            if nt > 9000:
                t[nt - 5] = q[3]; t[nt // 2 + 1] = q[5]; t[11] = q[5]   # ... and in different slices
        fq = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(nq, KP_DTYPE), q)
        ft = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(nt, KP_DTYPE), t)
        idx = np.zeros((nq, 2), np.int32)
        dist = np.zeros((nq, 2), np.float32)
        ctx.check(ctx.lib.mis_knn2(ctx.h, C.byref(fq.raw), C.byref(ft.raw), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
        oi, od = oracle_mod.knn2_l2(q, t)
        assert np.array_equal(idx, oi)
        assert np.array_equal(dist.view(np.uint32), od.view(np.uint32))
    # non-integer descriptors are refused (the fp16 MFMA path would not be exact)
    bad = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(8, KP_DTYPE), rng.uniform(0, 1, (8, 128)).astype(np.float32))
    with pytest.raises(isa.MisError):
        ctx.check(ctx.lib.mis_knn2(ctx.h, C.byref(bad.raw), C.byref(bad.raw), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))


def test_match_all_pairs_float_descriptors(ctx, oracle_mod):
    """BestOf2NearestMatcher over SIFT-style float descriptors (match_conf 0.65 as with xfeatures2d)."""
    import image_stitching_amd as isa
    from image_stitching_amd.stitching import KP_DTYPE
    rng = np.random.default_rng(52)
    H = np.array([[0.98, 0.02, 40.0], [-0.01, 1.01, -9.0], [1e-5, 0.0, 1.0]])
    n = 500
    base = _sift_like(rng, n)
    k1 = np.zeros(n, KP_DTYPE); k1["x"] = rng.uniform(0, 640, n); k1["y"] = rng.uniform(0, 480, n)
    p = np.c_[k1["x"] - 320, k1["y"] - 240, np.ones(n)] @ H.T
    k2 = np.zeros(n, KP_DTYPE); k2["x"] = p[:, 0] / p[:, 2] + 320; k2["y"] = p[:, 1] / p[:, 2] + 240
    d2 = np.clip(base + rng.integers(-2, 3, base.shape), 0, 255).astype(np.float32)
    d2[:80] = _sift_like(rng, 80)       # outliers
    k3, d3 = np.zeros(300, KP_DTYPE), _sift_like(rng, 300)
    k3["x"] = rng.uniform(0, 640, 300); k3["y"] = rng.uniform(0, 480, 300)
    sets = [(k1, base), (k2, d2), (k3, d3)]
    feats = [isa.ImageFeatures.upload(ctx, (640, 480), k, d, i) for i, (k, d) in enumerate(sets)]
    pm = isa.BestOf2NearestMatcher(ctx, 0.65)(feats)
    ref = oracle_mod.match_all_pairs([_feat_dict(k, d, (640, 480)) for k, d in sets], oracle_mod.match_default_params(match_conf=0.65))
    for g, o in zip(pm, ref):
        _compare_matches_info(g, o)
    # 420 exact inliers of 420 matches: inliers / (8 + 0.3 m) > 3 -> the "too similar images" rule zeroes it
    assert pm[1].num_inliers >= 400 and pm[1].confidence == 0.0 and pm[2].num_inliers < 20


@pytest.mark.parametrize("n", [5, 6, 8, 10])
def test_find_homography_tiny_problems_incl_infeasible(ctx, oracle_mod, n):
    """Tiny correspondence sets: (a) every 4-subset fails checkSubset (collinear source points) -> getSubset gives up,
    no model -- the GPU short-cuts the 10000 attempts by testing all ordered 4-tuples; (b) a feasible tiny set must not
    take the short cut and still equals the oracle bit for bit."""
    import image_stitching_amd as isa
    rng = np.random.default_rng(40 + n)
    # (a) infeasible: all source points on one line
    t = rng.uniform(-200, 200, n).astype(np.float32)
    src = np.stack([t, (0.5 * t + 3).astype(np.float32)], 1).astype(np.float32)
    dst = rng.uniform(-200, 200, (n, 2)).astype(np.float32)
    ok_g, H_g, m_g = isa.find_homography(ctx, src, dst)
    ok_o, H_o, m_o, _ = oracle_mod.find_homography_ransac(src, dst)
    assert bool(ok_o) is False and ok_g is False
    assert not m_g.any() and not np.asarray(m_o).any()
    # (b) feasible: an exact homography plus one or two outliers
    H = np.array([[1.01, 0.02, 12.0], [-0.015, 0.99, -7.0], [2e-5, -1e-5, 1.0]])
    src = rng.uniform(-300, 300, (n, 2)).astype(np.float32)
    p = np.c_[src, np.ones(n)] @ H.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst[0] += 40
    ok_g, H_g, m_g = isa.find_homography(ctx, src, dst)
    ok_o, H_o, m_o, _ = oracle_mod.find_homography_ransac(src, dst)
    assert ok_g == bool(ok_o)
    assert np.array_equal(m_g, m_o)
    if ok_g:
        assert np.array_equal(_bits(H_g), _bits(H_o))


_CHAINS3_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import synth, image_stitching_amd as isa
ctx = isa.Context(0)
cams = synth.workload("config3")[4:10]
frames = [synth.render_frame_gpu(c) for c in cams]
feats = isa.OrbFeatureFinder(ctx, (3840, 2160)).detect_batch(frames)
pm = isa.BestOf2NearestMatcher(ctx, 0.32)(feats)
out = {}
for k, m in enumerate(pm):
    out["c%d" % k] = np.float64(m.confidence); out["n%d" % k] = np.int64(m.num_inliers)
    out["H%d" % k] = np.asarray(m.H, np.float64) if m.H is not None else np.zeros(0)
    out["m%d" % k] = np.asarray(m.inliers_mask, np.uint8) if m.inliers_mask is not None else np.zeros(0, np.uint8)
np.savez(sys.argv[2], **out)
'''


def test_three_chain_matcher_flow_gives_identical_results(tmp_path):
    """MIS_MATCH_CHAINS=3 (second estimation from the RANSAC mask while the first H is still being refined, the |det H| test
    on the host) against the default flow: six 4K frames, every pair's confidence, inlier count, mask and H bit for bit.
    The switch is read once per process, hence two child processes (one after the other)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for chains in ("2", "3"):
        path = str(tmp_path / ("chains%s.npz" % chains))
        env = dict(os.environ, MIS_MATCH_CHAINS=chains)
        r = subprocess.run([sys.executable, "-c", _CHAINS3_SCRIPT, root, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(path))
    a, b = outs
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 4 * 36
    ran = 0
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k
        ran += k.startswith("n") and int(a[k]) > 0
    assert ran >= 10      # adjacent frames do have inliers: the second estimation was exercised


def test_pipelined_ordered_sums_equal_the_plain_loop(tmp_path):
    """The DLT / LM passes of the tails add their terms through ordered_sums (producer waves form the terms, one wave adds them in
    point order, structurally-zero terms are not formed; homography.hip) -- MIS_TAIL_PLAIN=1 sends every sum through the plain loop
    that forms all 90 products per point instead (the path a chunk with a non-finite record takes).  Six 4K frames: every pair's
    confidence, inlier count, mask and H bit for bit; the default flow itself is checked against the oracle by the tests above."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for plain in ("0", "1"):
        path = str(tmp_path / ("plain%s.npz" % plain))
        env = dict(os.environ, MIS_TAIL_PLAIN=plain)
        r = subprocess.run([sys.executable, "-c", _CHAINS3_SCRIPT, root, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(path))
    a, b = outs
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 4 * 36
    ran = 0
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k
        ran += k.startswith("n") and int(a[k]) > 200      # (tails with hundreds of inliers: several 256-point chunks per pass)
    assert ran >= 5


def test_enqueue_hook_runs_once_inside_the_next_call(ctx, oracle_mod):
    """mis_match_on_enqueued: the hook runs on the calling thread of the next matcher call (after its work is enqueued), once;
    a cleared hook does not run; the matches are those of a call without a hook."""
    import ctypes as C
    import threading
    import torch
    import synth
    import image_stitching_amd as isa
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 14.0 * i - 14.0, 0.0, 0.0) for i in range(3)]
    finder = isa.OrbFeatureFinder(ctx, (w, h))
    feats = [isa.computeImageFeatures(finder, torch.from_numpy(synth.render_frame(c)).cuda(), i) for i, c in enumerate(cams)]
    matcher = isa.BestOf2NearestMatcher(ctx, 0.32)
    ref = matcher(feats)
    seen = []
    cb = C.CFUNCTYPE(None, C.c_void_p)(lambda _u: seen.append((threading.get_ident(), int(ctx.lib.mis_match_sequence(ctx.h)))))
    seq0 = int(ctx.lib.mis_match_sequence(ctx.h))
    ctx.check(ctx.lib.mis_match_on_enqueued(ctx.h, C.cast(cb, C.c_void_p), None))
    a = matcher(feats)
    b = matcher(feats)                      # the hook was one-shot
    assert seen == [(threading.get_ident(), seq0 + 1)]
    ctx.check(ctx.lib.mis_match_on_enqueued(ctx.h, C.cast(cb, C.c_void_p), None))
    ctx.check(ctx.lib.mis_match_on_enqueued(ctx.h, None, None))
    matcher(feats)
    assert len(seen) == 1
    for x, y, z in zip(ref, a, b):
        assert np.array_equal(x.matches, y.matches) and np.array_equal(x.matches, z.matches)
        assert x.confidence == y.confidence == z.confidence
