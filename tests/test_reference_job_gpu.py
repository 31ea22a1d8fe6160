"""GPU parity: the job the reference's main() actually runs with its globals untouched (image_stitching/image_stitching.cpp:49-85)
-- reprojection bundle adjustment with the mask "_____" + horizontal wave correction, block gain compensation, DpSeamFinder(COLOR)
at seam_megapix, the compositing loop at compose_megapix (intrinsics and warper scale times compose_work_aspect, frames resized
INTER_LINEAR_EXACT), 8UC3 -> 16SC3, seam mask dilate -> resize -> AND, MultiBandBlender -- as ONE job (StitchJob with
StitchConfig.reference()) against the oracle's run of the same sequence (oracle/job.py: stitch_job_reference), SURVEY rows N1 / N1b / N1c."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _compare(job, out, ref, n_kept):
    eng = job.engine
    assert out["indices"] == ref["indices"] and len(ref["indices"]) == n_kept
    n = int(round(np.sqrt(np.asarray(out["confidence"].cpu()).size)))
    assert np.array_equal(np.asarray(out["confidence"].cpu()).reshape(n, n), ref["confidence"])
    # refined cameras of the kept frames: f64 compared as integers
    for i, c in zip(ref["indices"], ref["cameras"]):
        got = job.cams[i]
        assert np.array_equal(np.asarray(got["R"], np.float64).view(np.uint64), np.asarray(c["R"], np.float64).view(np.uint64)), "R of frame %d" % i
        assert float(got["K"][0, 0]) == c["focal"] and float(got["K"][0, 2]) == c["ppx"] and float(got["K"][1, 2]) == c["ppy"]
    assert float(np.float32(job.scale)) == float(np.float32(ref["scale"]))
    compensator, seam_masks = eng._seam
    for k in range(n_kept):
        assert np.array_equal(seam_masks[k].cpu().numpy(), ref["seam_masks"][k]), "seam mask %d" % k
        assert np.array_equal(compensator.gain_map(k).view(np.uint32), ref["gain_maps"][k].view(np.uint32)), "gain map %d" % k
    assert [tuple(job._compose_rois[i]) for i in ref["indices"]] == [tuple(r) for r in ref["rois"]]
    assert out["num_bands"] == ref["num_bands"] and tuple(out["pano_size"]) == tuple(ref["pano_size"])
    assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"])
    assert np.array_equal(out["pano"].cpu().numpy(), ref["pano"])


def test_reference_job_small_sweep_bit_exact(ctx, oracle_mod):
    """Six 640 x 360 frames from cameras off by ~0.4 degrees (the adjuster has something to do); compose_megapix and seam_megapix
    scaled down with the frames so that both resizes happen (compose scale 0.59, seam scale 0.29)."""
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    from oracle import job as ojob
    w, h, n = 640, 360, 6
    exact = [synth.make_camera(w, h, 60.0, 12.0 * i - 30.0, 2.0 * ((i % 3) - 1), 1.2 * ((i % 2) - 0.5), 0.96 + 0.015 * i) for i in range(n)]
    rng = np.random.default_rng(5)
    noisy = []
    for c in exact:
        d = dict(c)
        d["R"] = synth.rotation_yxz(*np.radians(rng.normal(0, 0.4, 3))) @ c["R"]
        noisy.append(d)
    host = [synth.render_frame(c) for c in exact]
    cfg = isa.StitchConfig.reference(compose_megapix=0.08, seam_megapix=0.02)
    job = StitchJob(ctx, (w, h), noisy, config=cfg)
    out = job.run({i: torch.from_numpy(f).cuda() for i, f in enumerate(host)})
    ref = ojob.stitch_job_reference(host, noisy, compose_megapix=0.08, seam_megapix=0.02)
    assert abs(isa.stitching.compose_geometry(cfg, (w, h), 1.0).compose_scale - 0.589) < 0.01
    _compare(job, out, ref, n)
    # the same configuration through the per-call mirror of main()'s compositing loop (Stitcher.compose) with the refined cameras
    st = isa.Stitcher(ctx, (w, h), cfg)
    pano2, mask2 = st.compose({i: torch.from_numpy(f).cuda() for i, f in enumerate(host)}, job.cams, out["indices"])
    assert np.array_equal(mask2.cpu().numpy(), ref["mask"]) and np.array_equal(pano2.cpu().numpy(), ref["pano"])


def test_reference_job_config3_16x4k_bit_exact(ctx, oracle_mod):
    """BASELINE config 3's sweep (16 x 3840 x 2160) through the reference's default job: features and matching at full resolution,
    seams at 0.1 MP, composition at 0.4 MP (compose scale 0.2196: 843 x 474 frames), ground-truth cameras as the start."""
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    from oracle import job as ojob
    cams = synth.workload("config3")
    w, h = cams[0]["width"], cams[0]["height"]
    dev = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
    torch.cuda.synchronize()
    cfg = isa.StitchConfig.reference()
    assert cfg.compose_megapix == 0.4 and cfg.seam_megapix == 0.1 and cfg.ba_cost_func == "reproj" and cfg.wave_correct == "horiz"
    assert cfg.expos_comp_type == "gain_blocks" and cfg.seam_find_type == "dp_color"
    g = isa.stitching.compose_geometry(cfg, (w, h), 1.0)
    assert g.size == (843, 474)
    job = StitchJob(ctx, (w, h), cams, config=cfg)
    out = job.run(dev)
    ref = ojob.stitch_job_reference([dev[i].cpu().numpy() for i in range(len(cams))], cams)
    _compare(job, out, ref, 16)
    out2 = job.run(dev)             # a second run of the same job object (refined cameras are not carried over)
    assert torch.equal(out2["pano"], out["pano"]) and torch.equal(out2["mask"], out["mask"])
