"""Camera refinement (row N1 of SURVEY section 8; image_stitching.cpp:681-726): bundle adjustment with the
reprojection cost and wave correction.  Host logic restated from OpenCV (parity unpinned): bit-exact against the oracle's
independent restatement (oracle/mo_motion.c) and behavioural checks -- perturbed cameras come back to the ground truth, the
reprojection error collapses, the horizon is levelled."""
import math

import numpy as np
import pytest


def _rot_err_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return math.degrees(math.acos(max(-1.0, min(1.0, c))))


def test_wave_correct_levels_a_tilted_sweep():
    import synth
    import image_stitching_amd as isa
    # a yaw sweep recorded with a common roll: after the correction every camera's x axis is horizontal again
    Rs = [synth.rotation_yxz(math.radians(12.0 * i - 30), math.radians(1.5), math.radians(6.0)) for i in range(6)]
    # the stitcher's cameras are camera->world rotations whose first column is the camera x axis in the world
    before = max(abs(R[1, 0]) for R in Rs)
    out = isa.wave_correct(Rs, isa.stitching.WAVE_CORRECT_HORIZ)
    after = max(abs(R[1, 0]) for R in out)
    assert before > 0.09 and after < 0.012        # the x axes of a rolled + pitched sweep lie on a cone: a least-squares plane remains
    for R in out:
        assert np.allclose(R @ R.T, np.eye(3), atol=2e-6)            # the reference corrects CV_32F rotations: float round-off
    # relative rotations are preserved (one global rotation applied to all)
    for i in range(5):
        assert _rot_err_deg(Rs[i].T @ Rs[i + 1], out[i].T @ out[i + 1]) < 0.05     # float rotations: acos near 1 magnifies 1e-7 to ~0.03 degree
    assert len(isa.wave_correct(Rs[:1])) == 1


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.mark.parametrize("kind", [0, 1])
def test_wave_correct_matches_oracle(oracle_mod, kind):
    """waveCorrect HORIZ / VERT on CV_32F rotations: library (host logic, no GPU needed) == oracle, bit for bit."""
    import synth
    import image_stitching_amd as isa
    rng = np.random.default_rng(40 + kind)
    for n in (2, 5, 16):
        Rs = [synth.rotation_yxz(math.radians(15.0 * i - 40 + rng.normal(0, 1)), math.radians(rng.normal(0, 3)), math.radians(4.0 + rng.normal(0, 2))) for i in range(n)]
        got = isa.wave_correct(Rs, kind)
        want = oracle_mod.wave_correct(Rs, kind)
        for g, w_ in zip(got, want):
            assert np.array_equal(_bits(g), _bits(w_))
            assert np.array_equal(g, g.astype(np.float32).astype(np.float64))       # CV_32F values


def _reproj_rms(feats, pm, n, cl, strong):
    """RMS distance (px) between the inlier keypoints of image j and those of image i mapped through K_j R_j^-1 R_i K_i^-1."""
    sq, cnt = 0.0, 0
    for i, j in strong:
        mi = pm[i * n + j]
        ki, kj = feats[i].download()[0], feats[j].download()[0]
        Ki = np.array([[cl[i]["focal"], 0, cl[i]["ppx"]], [0, cl[i]["focal"] * cl[i].get("aspect", 1.0), cl[i]["ppy"]], [0, 0, 1]])
        Kj = np.array([[cl[j]["focal"], 0, cl[j]["ppx"]], [0, cl[j]["focal"] * cl[j].get("aspect", 1.0), cl[j]["ppy"]], [0, 0, 1]])
        H = Kj @ np.linalg.inv(cl[j]["R"]) @ cl[i]["R"] @ np.linalg.inv(Ki)
        for m, keep in zip(mi.matches, mi.inliers_mask):
            if not keep:
                continue
            p = H @ np.array([ki["x"][m["query_idx"]], ki["y"][m["query_idx"]], 1.0])
            sq += (kj["x"][m["train_idx"]] - p[0] / p[2]) ** 2 + (kj["y"][m["train_idx"]] - p[1] / p[2]) ** 2
            cnt += 1
    return math.sqrt(sq / cnt)


def _as_params(c):
    return dict(focal=float(c["K"][0, 0]), aspect=float(c["K"][1, 1] / c["K"][0, 0]), ppx=float(c["K"][0, 2]), ppy=float(c["K"][1, 2]), R=c["R"])


@pytest.mark.gpu
def test_bundle_adjustment_recovers_perturbed_cameras(ctx):
    import torch
    import synth
    import image_stitching_amd as isa
    w, h, n = 640, 360, 5
    cams = [synth.make_camera(w, h, 60.0, 11.0 * i - 22.0, 0.8 * ((i % 3) - 1), 0.5 * ((i % 2) - 0.5)) for i in range(n)]
    frames = [torch.from_numpy(synth.render_frame(c)).cuda() for c in cams]
    finder = isa.OrbFeatureFinder(ctx, (w, h))
    feats = finder.detect_batch(frames)
    pm = isa.BestOf2NearestMatcher(ctx, 0.32)(feats)
    strong = [(i, j) for i in range(n) for j in range(i + 1, n) if pm[i * n + j].confidence > 0.95]
    assert len(strong) >= n - 1

    def reproj_rms(cl):
        return _reproj_rms(feats, pm, n, cl, strong)

    truth = [dict(focal=c["f"], ppx=c["K"][0, 2], ppy=c["K"][1, 2], aspect=1.0, R=c["R"]) for c in cams]
    rng = np.random.default_rng(5)
    start = []
    for c in truth:
        d = synth.rotation_yxz(*np.radians(rng.normal(0, 0.6, 3)))           # ~0.6 degree of sensor error per axis
        start.append(dict(focal=c["focal"] * (1 + rng.normal(0, 0.01)), ppx=c["ppx"] + rng.normal(0, 2), ppy=c["ppy"] + rng.normal(0, 2),
                          aspect=1.0, R=d @ c["R"]))
    e_truth, e_start = reproj_rms(truth), reproj_rms(start)
    refined = isa.bundle_adjust_reproj(ctx, feats, pm, start, conf_thresh=0.95)
    # the oracle's restatement on the same features / matches / cameras: every refined parameter bit for bit
    import oracle
    host = [f.download() for f in feats]
    ofe = [dict(img_w=w, img_h=h, xy=np.stack([k["x"], k["y"]], 1), desc=d) for k, d in host]
    for mask_s in ("xxxxx", "_____", "x_xxx"):
        want, iters = oracle.bundle_adjust_reproj(ofe, [pm[k] for k in range(n * n)], start, 0.95, mask_s)
        got = refined if mask_s == "xxxxx" else isa.bundle_adjust_reproj(ctx, feats, pm, start, conf_thresh=0.95, refine_mask=mask_s)
        assert iters >= 2
        for g, o in zip(got, want):
            for key in ("focal", "aspect", "ppx", "ppy"):
                assert g[key] == o[key], (mask_s, key)
            assert np.array_equal(_bits(g["R"]), _bits(o["R"])), mask_s
    e_ref = reproj_rms(refined)
    print("reprojection rms (px): truth %.3f  perturbed %.3f  refined %.3f" % (e_truth, e_start, e_ref))
    assert e_truth < 2.0 and e_start > 5 * e_truth                                 # ORB localisation noise; the perturbation matters
    assert e_ref < 1.5 * e_truth + 0.2                                             # back at the noise floor of the keypoints
    # relative rotations of neighbouring cameras agree with the ground truth
    err_ref = [_rot_err_deg(truth[i]["R"].T @ truth[i + 1]["R"], refined[i]["R"].T @ refined[i + 1]["R"]) for i in range(n - 1)]
    err_start = [_rot_err_deg(truth[i]["R"].T @ truth[i + 1]["R"], start[i]["R"].T @ start[i + 1]["R"]) for i in range(n - 1)]
    print("relative rotation error (deg): perturbed", np.round(err_start, 3), "refined", np.round(err_ref, 3))
    assert max(err_ref) < 0.8 and np.mean(err_ref) < 0.6 * np.mean(err_start)   # focal / principal point are free too: a few tenths remain
    # the centre image of the spanning tree carries the identity
    assert min(_rot_err_deg(np.eye(3), c["R"]) for c in refined) < 0.05          # R_c^-1 * R_c on CV_32F matrices
    for c in refined:
        assert abs(c["focal"] / truth[0]["focal"] - 1) < 0.06     # a yaw sweep observes the focal length weakly: the solver drifts a few % along it
    # refinement mask: nothing but the rotations may move
    fixed = isa.bundle_adjust_reproj(ctx, feats, pm, start, conf_thresh=0.95, refine_mask="_____")
    for a, b in zip(fixed, start):
        assert a["focal"] == b["focal"] and a["ppx"] == b["ppx"] and a["ppy"] == b["ppy"]
    with pytest.raises(isa.MisError):
        isa.bundle_adjust_reproj(ctx, feats, pm, start, conf_thresh=1e9)           # no pair above the threshold


@pytest.mark.gpu
def test_job_with_camera_refinement_from_perturbed_cameras(ctx):
    """The whole job from noisy cameras: with ba_cost_func = "reproj" the panorama comes out close to the one stitched
    from the exact cameras (up to the global rotation the refinement fixes), without it the mosaic is visibly torn."""
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    w, h, n = 640, 360, 5
    # (pitch and roll vary: a pure yaw sweep leaves the aspect ratio almost unobservable and the LM drifts along it)
    exact = [synth.make_camera(w, h, 60.0, 11.0 * i - 22.0, 2.5 * ((i % 3) - 1), 1.5 * ((i % 2) - 0.5)) for i in range(n)]
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(exact)}
    rng = np.random.default_rng(9)
    noisy = []
    for c in exact:
        d = dict(c)
        d["R"] = synth.rotation_yxz(*np.radians(rng.normal(0, 0.5, 3))) @ c["R"]
        noisy.append(d)
    job = StitchJob(ctx, (w, h), noisy, config=isa.StitchConfig.hot_path(ba_cost_func="reproj", ba_refine_mask="xxxxx"))
    out = job.run(frames)
    assert out["indices"] == list(range(n))
    # The adjustment minimises the reprojection error of the inlier matches; with focal, aspect and principal point free
    # it does not have to land on the ground truth (a narrow sweep trades focal length against rotation), so the check is
    # on its own objective: the refined cameras explain the matches at the keypoints' noise floor.
    feats, pm = out["features"], out["matches"]
    strong = [(i, j) for i in range(n) for j in range(i + 1, n) if pm[i * n + j].confidence > 0.95]
    e_noisy = _reproj_rms(feats, pm, n, [_as_params(c) for c in noisy], strong)
    e_exact = _reproj_rms(feats, pm, n, [_as_params(c) for c in exact], strong)
    e_refined = _reproj_rms(feats, pm, n, [_as_params(c) for c in job.cams], strong)
    print("reprojection rms (px): noisy %.2f exact %.2f refined %.2f" % (e_noisy, e_exact, e_refined))
    assert e_noisy > 4 * e_exact and e_refined < 1.2 * e_exact + 0.2
    # the reference's default mask "_____" refines the rotations only: intrinsics stay, the error still collapses
    job2 = StitchJob(ctx, (w, h), noisy, config=isa.StitchConfig.hot_path(ba_cost_func="reproj"))
    out2 = job2.run(frames)
    for a, b in zip(job2.cams, noisy):
        assert np.array_equal(a["K"], b["K"])
    assert _reproj_rms(feats, pm, n, [_as_params(c) for c in job2.cams], strong) < 1.3 * e_exact + 0.3
    pw, ph = out["pano_size"]
    assert pw > 2 * w * 0.8 and out["mask"].float().mean() > 100            # a panorama of sensible extent came out
    # the sharded job's path (match entries gathered from the ranks, table rebuilt on the host) gives the same cameras and panorama
    job3 = StitchJob(ctx, (w, h), noisy, config=isa.StitchConfig.hot_path(ba_cost_func="reproj"), force_collectives=True)
    out3 = job3.run(frames)
    for a, b in zip(job2.cams, job3.cams):
        assert np.array_equal(np.asarray(a["R"]), np.asarray(b["R"])) and np.array_equal(np.asarray(a["K"]), np.asarray(b["K"]))
    assert torch.equal(out3["pano"], out2["pano"]) and torch.equal(out3["mask"], out2["mask"])
