"""world_size-2 gloo runs of the sharded panorama job (image_stitching_amd.distributed.StitchJob) on CPU.

The orchestration (frame blocks, feature all-gather, round-robin pairs, confidence all-reduce, column-strip
exchange of the blend pyramids, per-strip finalise, strip all-gather) is engine-agnostic; here the oracle-backed engine from
tests/oracle_engine.py stands in for the HIP engine so the N > 1 path runs without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, NFRAMES = 256, 144, 4


def _cams():
    import synth
    return [synth.make_camera(W, H, 60.0, 13.0 * i - 20.0, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i in range(NFRAMES)]


def _cams_stray_block():
    """Frames 0, 1 overlap; frames 2, 3 (rank 1's whole block at world size 2) look elsewhere and at nothing of each other:
    leaveBiggestComponent keeps [0, 1] and rank 1 is left without a frame."""
    import synth
    yaws = [-6.0, 7.0, 105.0, -150.0]
    return [synth.make_camera(W, H, 60.0, y, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5)) for i, y in enumerate(yaws)]


def _run_job(rank, world, group, seam=False, stray=False):
    import synth
    from image_stitching_amd.distributed import StitchJob
    from image_stitching_amd.stitching import StitchConfig
    from oracle_engine import OracleEngine
    cams = _cams_stray_block() if stray else _cams()
    cfg = StitchConfig(seam_megapix=0.02) if seam else StitchConfig.hot_path()     # seam: the reference's defaults (gain_blocks + dp_color)
    job = StitchJob(None, (W, H), cams, rank=rank, world_size=world, group=group, engine=OracleEngine((W, H), config=cfg), config=cfg)
    frames = {i: synth.render_frame(cams[i]) for i in job.my_frames}
    return job, job.run(frames)


def _worker(rank, world, port, out_path, seam=False, stray=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        job, out = _run_job(rank, world, dist.group.WORLD, seam, stray)
        assert job.my_frames == list(range(rank * 2, rank * 2 + 2))
        if rank == 0:
            np.savez(out_path, pano=out["pano"], mask=out["mask"], conf=out["confidence"].numpy(), indices=np.array(out["indices"]),
                     nfeat=np.array([len(f["kps"]) for f in out["features"]]))
        else:
            assert out["pano"] is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_frame_block_partition():
    from image_stitching_amd.distributed import frame_block
    assert frame_block(16, 0, 1) == list(range(16))
    assert [frame_block(64, r, 8) for r in (0, 7)] == [list(range(8)), list(range(56, 64))]
    assert sum((frame_block(10, r, 4) for r in range(4)), []) == list(range(10))


def test_two_rank_job_matches_single_rank(tmp_path):
    # single-process reference
    _, ref = _run_job(0, 1, None)
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    # integer / index results are identical; every pair was matched by exactly one rank
    assert list(got["indices"]) == list(ref["indices"]) == [0, 1, 2, 3]
    assert np.array_equal(got["conf"], ref["confidence"].numpy())
    assert list(got["nfeat"]) == [len(f["kps"]) for f in ref["features"]]
    assert np.array_equal(got["mask"], ref["mask"])
    # 16SC3 Laplacian sums are exact under any order; f32 weight sums differ in the last bit where >= 3
    # frames overlap -> at most 1 LSB in the blended pixels (north-star tolerance)
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].astype(np.int32))
    assert d.max() <= 1
    assert (d > 0).mean() < 0.02


def test_two_rank_result_is_the_oracle_with_rank_order_sums(tmp_path):
    """The exchange fixes the association of the f32 weight sums: per-rank partial sums (each rank's frames in feed order)
    added in rank order from zero.  An oracle run with exactly that association reproduces the 2-rank panorama bit for bit."""
    import oracle
    import synth
    from oracle import job as ojob
    cams = _cams()
    frames = [synth.render_frame(c) for c in cams]
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    scale = ojob.warped_image_scale(cams)
    rois = [oracle.warp_roi(scale, W, H, c["K"].astype(np.float32), c["R"].astype(np.float32)) for c in cams]
    corners, sizes = [(r[0], r[1]) for r in rois], [(r[2], r[3]) for r in rois]
    x0 = min(c[0] for c in corners); y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, sizes)); y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
    btype, bands, sharp = oracle.blend_config(oracle.BLEND_MULTI_BAND, 5.0, x1 - x0, y1 - y0)
    from oracle_engine import OracleEngine
    engs = []
    for r in range(2):
        e = OracleEngine((W, H))
        e.begin_compose(scale, corners, sizes)
        for i in (2 * r, 2 * r + 1):
            e.warp_feed(frames[i], cams[i], rois[i])
        engs.append(e)
    a0, a1 = engs[0].accumulators(), engs[1].accumulators()
    for (l0, w0), (l1, w1) in zip(a0, a1):
        w0.copy_((torch.zeros_like(w0) + w0) + w1)          # ((0 + p0) + p1)
        l0.copy_((l0.to(torch.int32) + l1.to(torch.int32)).to(torch.int16))   # wraps
    pano, mask = engs[0].finalize()
    assert np.array_equal(got["mask"], mask)
    assert np.array_equal(got["pano"], pano)


def test_strip_plan_covers_what_a_strip_needs():
    """Strip bounds tile the padded panorama on 2^bands boundaries; the need ranges follow pyrUp's reach; the exchange
    rectangles of a rank are its region clipped to them, with 16-byte aligned blocks."""
    from image_stitching_amd.distributed import StitchJob
    sizes = [(1024 >> l, 256 >> l) for l in range(5)]            # 4 bands
    b = StitchJob.strip_bounds(1024, 4, 3)
    assert b == [(0, 336), (336, 672), (672, 1024)] and all(x % 16 == 0 for bb in b for x in bb)
    need = StitchJob.need_ranges(b[1], sizes)
    assert need[0] == (336, 672) and need[1] == (167, 337) and need[2] == (82, 170) and need[4][0] >= 0 and need[4][1] <= 64
    # the recurrence reproduces what blend_columns collapses: every fine column of need[l] has its three coarse columns in need[l + 1]
    for l in range(4):
        lo, hi = need[l]
        assert need[l + 1][0] <= max(0, (lo >> 1) - 1) and need[l + 1][1] >= min(sizes[l + 1][0], ((hi - 1) >> 1) + 2)
    rects, nbytes = StitchJob.exchange_rects((320, 16, 720, 240), need, sizes)
    assert rects[0][:5] == (0, 336, 16, 672, 240) and rects[1][:5] == (1, 167, 8, 337, 120)
    assert all(r[5] % 16 == 0 for r in rects) and nbytes % 16 == 0
    assert StitchJob.exchange_rects(None, need, sizes) == ([], 0)
    assert StitchJob.exchange_rects((0, 0, 320, 64), need, sizes)[0][0][:3] == (4, 19, 0)      # a region left of the strip only meets its coarse halo


def test_two_rank_job_with_seam_step_matches_single_rank(tmp_path):
    """The reference's default seam-scale step (block gains + DpSeamFinder) inside the sharded job: the seam-scale images are
    all-gathered and every rank solves the same problem -- masks identical, pixels within the f32 association of the exchange."""
    _, ref = _run_job(0, 1, None, seam=True)
    _, plain = _run_job(0, 1, None, seam=False)
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out_path, True), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    assert np.array_equal(got["mask"], ref["mask"])
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02
    assert not np.array_equal(ref["pano"], plain["pano"])        # the seam step did change the panorama


def test_two_rank_seam_job_when_one_ranks_block_is_pruned(tmp_path):
    """Reference-default configuration (gain_blocks + dp_color) at N = 2 with rank 1's whole frame block stray: that rank has
    nothing to warp at seam scale or at compose scale but must stay in every collective (it used to leave the job with a
    ValueError from an empty batch, or to hand a CPU tensor to the all-gather, and the other rank hung in the next collective)."""
    _, ref = _run_job(0, 1, None, seam=True, stray=True)
    assert ref["indices"] == [0, 1]
    out_path = str(tmp_path / "rank0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out_path, True, True), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    assert list(got["indices"]) == [0, 1]
    assert np.array_equal(got["mask"], ref["mask"])
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].astype(np.int32))
    assert d.max() <= 1


def test_job_refuses_options_it_does_not_run():
    """Options the library does not implement are refused at construction, never silently ignored."""
    from image_stitching_amd.distributed import StitchJob
    from image_stitching_amd.stitching import StitchConfig
    from oracle_engine import OracleEngine
    cams = _cams()
    for cfg in (StitchConfig.hot_path(expos_comp_type="channels"), StitchConfig.hot_path(seam_find_type="gc_color")):
        with pytest.raises(NotImplementedError):
            StitchJob(None, (W, H), cams, engine=OracleEngine((W, H)), config=cfg)
