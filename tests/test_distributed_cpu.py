"""world_size-2 gloo runs of the sharded panorama job (image_stitching_amd.distributed.StitchJob) on CPU.

The orchestration (frame blocks, feature all-gather, round-robin pairs, confidence all-reduce, packed
region gather of the blend pyramids, root finalise) is engine-agnostic; here the oracle-backed engine from
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


def _run_job(rank, world, group):
    import synth
    from image_stitching_amd.distributed import StitchJob
    from oracle_engine import OracleEngine
    cams = _cams()
    job = StitchJob(None, (W, H), cams, rank=rank, world_size=world, group=group, engine=OracleEngine((W, H)))
    frames = {i: synth.render_frame(cams[i]) for i in job.my_frames}
    return job, job.run(frames)


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        job, out = _run_job(rank, world, dist.group.WORLD)
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


def test_region_pack_roundtrip_and_int16_wrap():
    """Packed rectangle exchange: pack -> add lands on the same pixels at every level; int16 adds wrap."""
    from image_stitching_amd.distributed import StitchJob
    g = torch.Generator().manual_seed(5)
    levels = []
    w, h = 64, 32
    for l in range(3):
        levels.append((torch.randint(-32768, 32767, (h, w * 3), generator=g, dtype=torch.int16), torch.rand((h, w), generator=g)))
        w, h = w // 2, h // 2
    region = (8, 4, 40, 28)
    rects = StitchJob._level_rects(region, levels)
    assert rects == [(8, 4, 40, 28), (4, 2, 20, 14), (2, 1, 10, 7)]
    buf = StitchJob._pack(levels, rects, StitchJob._packed_size(rects) + 64)
    dst = [(torch.full_like(a, 30000), torch.ones_like(b)) for a, b in levels]
    StitchJob._add_packed(dst, rects, buf)
    for (a, b), (da, db), (x0, y0, x1, y1) in zip(levels, dst, rects):
        exp = (a.to(torch.int32) + 30000).to(torch.int16)                  # two's-complement wrap
        assert torch.equal(da[y0:y1, 3 * x0:3 * x1], exp[y0:y1, 3 * x0:3 * x1])
        assert torch.equal(db[y0:y1, x0:x1], b[y0:y1, x0:x1] + 1)
        outside = torch.ones_like(db, dtype=torch.bool)
        outside[y0:y1, x0:x1] = False
        assert bool((db[outside] == 1).all()) and bool((da[outside.repeat_interleave(3, 1)] == 30000).all())


def test_job_refuses_options_it_does_not_run():
    """Options the sharded job cannot honour are refused at construction, never silently ignored."""
    from image_stitching_amd.distributed import StitchJob
    from image_stitching_amd.stitching import StitchConfig
    from oracle_engine import OracleEngine
    cams = _cams()
    for cfg in (StitchConfig(expos_comp_type="gain_blocks"), StitchConfig(seam_find_type="voronoi")):
        with pytest.raises(NotImplementedError):
            StitchJob(None, (W, H), cams, engine=OracleEngine((W, H)), config=cfg)
    with pytest.raises(NotImplementedError):
        StitchJob(None, (W, H), cams, rank=0, world_size=2, engine=OracleEngine((W, H)), config=StitchConfig(ba_cost_func="reproj"))
