"""GPU: the HipEngine side of the sharded job (feature packing for the all-gather, raw-pointer tensor views of the blender
pyramids, the column-strip exchange: pack / zero / add of pyramid rectangles, per-strip finalise, strip assembly) exercised at
world size 1 with the collective code paths forced on -- the result must equal the plain single-GPU path bit for bit -- and as
two and three processes on the one GPU of the box."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_forced_collective_paths_match_plain_job(ctx):
    import torch
    import synth
    from image_stitching_amd.distributed import StitchJob
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(4)]
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(cams)}
    plain = StitchJob(ctx, (w, h), cams).run(frames)
    forced = StitchJob(ctx, (w, h), cams, force_collectives=True).run(frames)
    assert forced["indices"] == plain["indices"] == [0, 1, 2, 3]
    assert torch.equal(forced["confidence"].cpu(), plain["confidence"].cpu())
    for a, b in zip(forced["features"], plain["features"]):
        ka, da = a.download()
        kb, db = b.download()
        assert np.array_equal(ka, kb) and np.array_equal(da, db)
    for a, b in zip(forced["matches"], plain["matches"]):
        assert np.array_equal(a.matches, b.matches) and np.array_equal(a.inliers_mask, b.inliers_mask)
    assert torch.equal(forced["pano"], plain["pano"]) and torch.equal(forced["mask"], plain["mask"])


def test_blend_columns_and_rect_exchange_equal_full_blend(ctx):
    """mis_blender_blend_columns on 2^bands-aligned column strips (each from freshly fed accumulators, as each rank has its
    own) reproduces the columns of the full blend; pack -> zero -> add of the need ranges leaves the result unchanged."""
    import ctypes as C
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import HipEngine, StitchJob
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(4)]
    frames = [torch.from_numpy(synth.render_frame(c)).cuda() for c in cams]
    eng = HipEngine(ctx, (w, h))
    scale = isa.Stitcher.warped_image_scale(cams)
    rois = eng.warp_rois(scale, cams)
    corners, sizes = [(r[0], r[1]) for r in rois], [(r[2], r[3]) for r in rois]

    def feed_all():
        with torch.cuda.stream(eng.compose_stream):
            eng.begin_compose(scale, corners, sizes)
            for f, c, r in zip(frames, cams, rois):
                eng.warp_feed(f, c, r)
    feed_all()
    with torch.cuda.stream(eng.compose_stream):
        full, fmask = eng.finalize()
    torch.cuda.synchronize()
    pw, ph = eng.pano_size
    feed_all()
    lsz = eng.level_sizes()
    bands = len(lsz) - 1
    assert bands >= 3
    bounds = StitchJob.strip_bounds(lsz[0][0], bands, 3)
    for k, (x0, x1) in enumerate(bounds):
        if k:
            feed_all()
        need = StitchJob.need_ranges((x0, x1), lsz)
        rects, nbytes = StitchJob.exchange_rects((0, 0, lsz[0][0], lsz[0][1]), need, lsz)
        with torch.cuda.stream(eng.compose_stream):
            buf = eng.pack_rects(rects, nbytes)
            eng.zero_rects([(l, lo, 0, hi, lsz[l][1], 0) for l, (lo, hi) in enumerate(need)])
            eng.add_rects(rects, buf)
            img, msk = eng.finalize_columns(x0, x1)
        torch.cuda.synchronize()
        x1c = min(x1, pw)
        assert torch.equal(img, full[:, x0:x1c]) and torch.equal(msk, fmask[:, x0:x1c]), k


def test_accumulator_views_alias_blender_memory(ctx):
    """dev_tensor views must alias the library's pyramids: a write through torch shows up in blend()."""
    import torch
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import HipEngine
    eng = HipEngine(ctx, (64, 64))
    eng.begin_compose(100.0, [(0, 0)], [(64, 64)])
    acc = eng.accumulators()
    assert len(acc) == eng.num_bands() + 1
    lap0, w0 = acc[0]
    lap0.fill_(7)
    w0.fill_(1.0)
    for lap, w in acc[1:]:
        lap.zero_()
        w.fill_(1.0)
    out, mask = eng.finalize()
    assert int(mask.min()) == 255 and int(out.min()) == 6 and int(out.max()) == 6   # (short)(7 / (1 + 1e-5)) = 6


def test_sift_job_forced_collectives_match_plain_job(ctx):
    """features_type = "sift": float descriptors through the variable-capacity gather and the L2 (MFMA) matcher."""
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 11.0 * i - 16.0, 0.4 * ((i % 3) - 1)) for i in range(4)]
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(cams)}
    cfg = isa.StitchConfig.hot_path(features_type="sift")
    plain = StitchJob(ctx, (w, h), cams, config=cfg).run(frames)
    forced = StitchJob(ctx, (w, h), cams, config=cfg, force_collectives=True).run(frames)
    assert plain["indices"] == forced["indices"] == [0, 1, 2, 3]
    assert torch.equal(forced["confidence"].cpu(), plain["confidence"].cpu())
    k, d = plain["features"][1].download()
    assert d.dtype == np.float32 and d.shape[1] == 128 and len(k) > 300
    for a, b in zip(forced["features"], plain["features"]):
        assert np.array_equal(a.download()[1], b.download()[1])
    assert torch.equal(forced["pano"], plain["pano"]) and torch.equal(forced["mask"], plain["mask"])


def _gpu_rank(rank, world, port, out_path):
    """One rank of a 2-process job on the SAME GPU (gloo rendezvous, device tensors staged through the host): the real
    HipEngine under world size 2 -- frame blocks, feature gather into raw-pointer views, sharded pairs, packed
    region gather, root finalise."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth
        import image_stitching_amd as isa
        from image_stitching_amd.distributed import StitchJob
        w, h = 480, 270
        cams = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(4)]
        ctx = isa.Context(0)
        job = StitchJob(ctx, (w, h), cams, rank=rank, world_size=world, group=dist.group.WORLD)
        frames = {i: torch.from_numpy(synth.render_frame(cams[i])).cuda() for i in job.my_frames}
        out = job.run(frames)
        if rank == 0:
            np.savez(out_path, pano=out["pano"].cpu().numpy(), mask=out["mask"].cpu().numpy(), conf=out["confidence"].cpu().numpy(),
                     indices=np.array(out["indices"]))
        else:
            assert out["pano"] is None
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_rank(ctx, tmp_path):
    import socket
    import torch
    import torch.multiprocessing as mp
    import synth
    from image_stitching_amd.distributed import StitchJob
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(4)]
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(cams)}
    ref = StitchJob(ctx, (w, h), cams).run(frames)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_gpu_rank, args=(2, port, out_path), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    assert list(got["indices"]) == ref["indices"] == [0, 1, 2, 3]
    assert np.array_equal(got["conf"], ref["confidence"].cpu().numpy())
    assert np.array_equal(got["mask"], ref["mask"].cpu().numpy())
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].cpu().numpy().astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02          # f32 weight sums in a different order where >= 3 frames overlap


@pytest.mark.parametrize("forced", [False, True])
def test_job_with_the_reference_default_seam_step_equals_the_stitcher(ctx, forced):
    """StitchConfig() -- the reference's defaults: block gain compensation + DpSeamFinder(COLOR) -- inside the job (also with the
    all-gather of the seam-scale images forced on) against Stitcher.compose, which makes one library call per reference call."""
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    w, h = 640, 360
    cams = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(4)]
    gains = [0.8, 1.0, 1.2, 0.9]
    host = [np.clip(synth.render_frame(c).astype(np.float32) * g, 0, 255).astype(np.uint8) for c, g in zip(cams, gains)]
    dev = [torch.from_numpy(f).cuda() for f in host]
    cfg = isa.StitchConfig(compose_megapix=-1)
    want, wmask = isa.Stitcher(ctx, (w, h), cfg).compose(dev, cams)
    out = StitchJob(ctx, (w, h), cams, config=cfg, force_collectives=forced).run({i: f for i, f in enumerate(dev)})
    assert out["indices"] == [0, 1, 2, 3]
    assert torch.equal(out["mask"], wmask) and torch.equal(out["pano"], want)
    plain = StitchJob(ctx, (w, h), cams).run({i: f for i, f in enumerate(dev)})
    assert not torch.equal(plain["pano"], want)


def _ba_scene():
    import synth
    w, h = 480, 270
    rng = np.random.default_rng(9)
    exact = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 2.5 * ((i % 3) - 1), 1.5 * ((i % 2) - 0.5)) for i in range(4)]
    noisy = []
    for c in exact:
        d = dict(c)
        d["R"] = synth.rotation_yxz(*np.radians(rng.normal(0, 0.5, 3))) @ c["R"]
        noisy.append(d)
    return w, h, exact, noisy


def _gpu_rank_ba(rank, world, port, out_path):
    """Two ranks with camera refinement: the matches of the pairs each rank owns are gathered, every rank solves the same problem."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth
        import image_stitching_amd as isa
        from image_stitching_amd.distributed import StitchJob
        w, h, exact, noisy = _ba_scene()
        ctx = isa.Context(0)
        job = StitchJob(ctx, (w, h), noisy, rank=rank, world_size=world, group=dist.group.WORLD, config=isa.StitchConfig.hot_path(ba_cost_func="reproj"))
        frames = {i: torch.from_numpy(synth.render_frame(exact[i])).cuda() for i in job.my_frames}
        out = job.run(frames)
        if rank == 0:
            np.savez(out_path, pano=out["pano"].cpu().numpy(), mask=out["mask"].cpu().numpy(), R=np.stack([np.asarray(c["R"]) for c in job.cams]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_bundle_adjustment_match_single_rank(ctx, tmp_path):
    import socket
    import torch
    import torch.multiprocessing as mp
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    w, h, exact, noisy = _ba_scene()
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(exact)}
    solo = StitchJob(ctx, (w, h), noisy, config=isa.StitchConfig.hot_path(ba_cost_func="reproj"))
    ref = solo.run(frames)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_gpu_rank_ba, args=(2, port, out_path), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    assert np.array_equal(got["R"], np.stack([np.asarray(c["R"]) for c in solo.cams]))        # the same refined cameras, bit for bit
    assert np.array_equal(got["mask"], ref["mask"].cpu().numpy())
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].cpu().numpy().astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02


def _gpu_rank6(rank, world, port, out_path):
    """One rank of a 3-process job over 6 frames on the same GPU (3 strips, halos crossing two owners)."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth
        import image_stitching_amd as isa
        from image_stitching_amd.distributed import StitchJob
        w, h = 480, 270
        cams = [synth.make_camera(w, h, 60.0, 24.0 * i - 60.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(6)]
        ctx = isa.Context(0)
        job = StitchJob(ctx, (w, h), cams, rank=rank, world_size=world, group=dist.group.WORLD)
        frames = {i: torch.from_numpy(synth.render_frame(cams[i])).cuda() for i in job.my_frames}
        out = job.run(frames)
        if rank == 0:
            np.savez(out_path, pano=out["pano"].cpu().numpy(), mask=out["mask"].cpu().numpy(), indices=np.array(out["indices"]))
    finally:
        dist.destroy_process_group()


def test_three_ranks_on_one_gpu_match_single_rank(ctx, tmp_path):
    import socket
    import torch
    import torch.multiprocessing as mp
    import synth
    from image_stitching_amd.distributed import StitchJob
    w, h = 480, 270
    cams = [synth.make_camera(w, h, 60.0, 24.0 * i - 60.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(6)]
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(cams)}
    ref = StitchJob(ctx, (w, h), cams).run(frames)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_gpu_rank6, args=(3, port, out_path), nprocs=3, join=True, start_method="spawn")
    got = np.load(out_path)
    assert list(got["indices"]) == ref["indices"]
    assert np.array_equal(got["mask"], ref["mask"].cpu().numpy())
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].cpu().numpy().astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02


@pytest.mark.parametrize("stray", [False, True])
def test_speculative_compose_equals_sequential(ctx, stray):
    """The job composes all frames on a second stream while the matcher runs; when the pruning drops a frame
    (stray = True: one camera looks elsewhere) the composition is redone -- either way the result equals the
    sequential order of the reference's main()."""
    import torch
    import synth
    from image_stitching_amd.distributed import StitchJob
    w, h = 480, 270
    yaws = [-20.0, -7.0, 6.0, 140.0 if stray else 19.0]
    cams = [synth.make_camera(w, h, 60.0, y, 0.3 * ((i % 3) - 1)) for i, y in enumerate(yaws)]
    frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(cams)}
    seq_job = StitchJob(ctx, (w, h), cams)
    seq_job.engine.speculative_compose = False
    seq = seq_job.run(frames)
    spec = StitchJob(ctx, (w, h), cams).run(frames)
    assert spec["indices"] == seq["indices"] == ([0, 1, 2] if stray else [0, 1, 2, 3])
    assert spec["pano_size"] == seq["pano_size"]
    assert torch.equal(spec["pano"], seq["pano"]) and torch.equal(spec["mask"], seq["mask"])
    assert torch.equal(spec["confidence"].cpu(), seq["confidence"].cpu())


def test_job_equals_the_per_call_stitcher(ctx):
    """StitchJob (batched entries: mis_orb_detect_batch, mis_compose_frames, speculative composition + collapse on a
    second stream) against Stitcher (one C call per reference call site, everything on one stream): same panorama,
    mask, kept indices and pairwise confidences, bit for bit."""
    import torch
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    w, h = 640, 360
    cams = [synth.make_camera(w, h, 60.0, 12.0 * i - 18.0, 0.5 * ((i % 3) - 1), 0.25 * ((i % 2) - 0.5)) for i in range(5)]
    dev = [torch.from_numpy(synth.render_frame(c)).cuda() for c in cams]
    job = StitchJob(ctx, (w, h), cams).run({i: f for i, f in enumerate(dev)})
    pano, mask, feats, pm, idx = isa.Stitcher(ctx, (w, h), isa.StitchConfig.hot_path(compose_megapix=-1)).stitch(dev, cams)
    assert list(idx) == job["indices"] == [0, 1, 2, 3, 4]
    assert np.array_equal(job["confidence"].cpu().numpy().reshape(-1), np.array([m.confidence for m in pm]))
    assert torch.equal(job["pano"], pano) and torch.equal(job["mask"], mask)


def _rccl_one_rank(port, out_path):
    """One rank, backend nccl (= RCCL), every collective of the N > 1 path issued for real (Comm(always_collective=True))."""
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import synth
    import image_stitching_amd as isa
    from image_stitching_amd.distributed import StitchJob
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = isa.Context(0)
        w, h = 480, 270
        cams = [synth.make_camera(w, h, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(4)]
        frames = {i: torch.from_numpy(synth.render_frame(c)).cuda() for i, c in enumerate(cams)}
        plain = StitchJob(ctx, (w, h), cams).run(frames)
        coll = StitchJob(ctx, (w, h), cams, force_collectives=True, always_collective=True).run(frames)
        ok = coll["indices"] == plain["indices"] and torch.equal(coll["confidence"].cpu(), plain["confidence"].cpu()) and \
            torch.equal(coll["pano"], plain["pano"]) and torch.equal(coll["mask"], plain["mask"])
        ba = isa.StitchConfig.hot_path(ba_cost_func="reproj")
        p2 = StitchJob(ctx, (w, h), cams, config=ba).run(frames)
        c2 = StitchJob(ctx, (w, h), cams, config=ba, force_collectives=True, always_collective=True).run(frames)
        ok = ok and torch.equal(c2["pano"], p2["pano"]) and torch.equal(c2["mask"], p2["mask"])
        torch.cuda.synchronize()
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_of_the_sharded_path_on_one_rank(ctx, tmp_path):
    """The all-gather / all-reduce / all-to-all / object gathers of the N > 1 job through RCCL itself (a one-rank group on the
    box's GPU: two ranks cannot share a device under RCCL): dtypes, split sizes and stream ordering of the real backend; the
    result equals the plain job bit for bit, with and without bundle adjustment."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "rccl.txt"
    p = mp.get_context("spawn").Process(target=_rccl_one_rank, args=(port, str(out)))
    p.start(); p.join(300)
    assert p.exitcode == 0
    assert out.read_text() == "ok"
