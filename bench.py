#!/usr/bin/env python3
"""bench.py -- 4K frames stitched per second on N MI355X (BASELINE.json metric).

A "step" is one full pass of the hot path over one panorama job: ORB detect+describe of every frame,
all-pairs 2-NN + RANSAC matching, connected-component pruning, spherical warp and multi-band blend
(compose_megapix = -1: true 4K warp + blend), with the frames already resident in HBM and the
ground-truth cameras standing in for the EXIF path.  N = 1 runs BASELINE config 3 (16 x 4K sweep);
N > 1 runs BASELINE config 4 (64 x 4K, the same 64-frame job for every N > 1: strong scaling) and adds
`same_workload_on_1_gpu`, the unsharded run of that job on rank 0, as the base of the speed-up.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` (warp kernel, HIP events)
and `cpu_baseline` (the oracle timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The job's streams (feature lanes, compose, the matcher's side chains) are sized for the runtime's default of 4 hardware
# queues per process: streams beyond that share queues and serialise (3 / 5 / 8 queues cost the step
# 1-3 ms).  Pinned here, before the HIP runtime starts, so that an inherited setting cannot change what is measured.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, help="config2 | config3 | config4 (default: config3, config4 when --gpus > 1)")
    ap.add_argument("--features", default=None, help="orb | sift (default: orb; sift for config5, BASELINE.json configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-stage timing table to stderr")
    ap.add_argument("--no-single-base", action="store_true", help="N > 1: skip the unsharded run of the same workload on rank 0")
    ap.add_argument("--roofline-launches", type=int, default=200)
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (for a rocprofv3 --stats pass in which no other kernel overlaps the warp kernel)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    import image_stitching_amd as isa
    import synth
    from image_stitching_amd import distributed as misdist

    # MIS_BENCH_REHEARSAL=1: the N > 1 code path on a box with ONE GPU (every rank on cuda:0, gloo instead of RCCL) --
    # a functional rehearsal of this script's multi-rank branch, not a measurement
    rehearsal = os.environ.get("MIS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        pg = dist.group.WORLD
    workload = args.workload or ("config3" if world == 1 else "config4")
    cams = synth.workload(workload)
    n = len(cams)
    W, H = cams[0]["width"], cams[0]["height"]
    ctx = isa.Context(local_rank)
    features = args.features or ("sift" if workload == "config5" else "orb")
    cfg = isa.StitchConfig(features_type=features)
    job = misdist.StitchJob(ctx, (W, H), cams, rank=rank, world_size=world, group=pg, config=cfg)
    # synthetic frames of this rank's shard, rendered straight into HBM
    frames = {i: synth.render_frame_gpu(cams[i], device="cuda:%d" % local_rank) for i in job.my_frames}
    torch.cuda.synchronize()

    def step():
        return job.run(frames)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    if args.roofline_only:
        roof = measure_roofline(ctx, job, frames, cams, args.roofline_launches)
        print(json.dumps({"roofline": roof}), flush=True)
        return
    for _ in range(args.warmup):
        out = step()          # same object lifetimes as the timed loop: the allocators reach their steady state here
    # the interpreter's cyclic garbage collector is paused for the timed region (as timeit does): a full collection
    # walks every torch / ctypes object and costs ~40 ms, twice the step being measured
    import gc
    gc.collect()
    gc.disable()
    barrier()
    trace = [] if os.environ.get("BENCH_TRACE") else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        out = step()
        if trace is not None:
            trace.append((time.perf_counter() - ts) * 1e3)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if trace is not None:
        log("per-step ms: " + " ".join("%.2f" % v for v in trace))
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = n * args.steps / dt

    # N > 1: rank 0 also runs the SAME job alone (no sharding) so that the strong-scaling base of this workload
    # is on the same line; the other ranks wait at the barrier below.  Outside the timed region.
    single = None
    if world > 1 and not args.no_single_base:
        if rank == 0:
            all_frames = {i: frames[i] if i in frames else synth.render_frame_gpu(cams[i], device="cuda:%d" % local_rank) for i in range(n)}
            solo = misdist.StitchJob(ctx, (W, H), cams, config=cfg)
            solo.run(all_frames)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                solo.run(all_frames)
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t1) / 2
            single = {"value": round(n / t1, 3), "ms_per_step": round(t1 * 1e3, 3), "n_gpus": 1, "workload": workload}
            del all_frames, solo
        barrier()

    breakdown = None
    if args.breakdown and rank == 0:
        breakdown = job.breakdown(frames)
        for k, v in breakdown.items():
            log("  %-28s %9.3f ms" % (k, v))

    roof = None
    cpu = None
    if rank == 0:
        roof = measure_roofline(ctx, job, frames, cams, args.roofline_launches)
        if not args.no_cpu_baseline and features == "orb" and world == 1:     # the CPU baseline is an N = 1 item
            cpu = cpu_baseline(cams, workload)
        res = {
            "metric": "4K frames stitched/sec", "value": round(value, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %d x %dx%d frames, %s + all-pairs 2-NN/RANSAC + spherical warp + multiband blend, compose_megapix=-1"
                                   % (workload, n, W, H, "SIFT (128-D f32, L2 on fp16 MFMA)" if features == "sift" else "ORB 4000"),
                       "frames": n, "frame_size": [W, H], "pairs": n * (n - 1) // 2, "pano_size": list(out["pano_size"]),
                       "num_bands": out["num_bands"], "parallelism": "frames sharded %d/GPU" % (n // world)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if single:
            res["same_workload_on_1_gpu"] = single
            res["speedup_vs_1_gpu_same_workload"] = round(value / single["value"], 3)
        if breakdown:
            res["breakdown_ms"] = {k: round(v, 3) for k, v in breakdown.items()}
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def measure_roofline(ctx, job, frames, cams, launches):
    """Warp kernel (K10): algorithmic bytes per launch / average launch duration.  The duration is measured with
    HIP events on the context's stream around `launches` back-to-back launches of the kernel for one frame, all
    enqueued from inside the library (mis_warp_spherical_fused_timed), so that host launch pacing does not enter."""
    import image_stitching_amd as isa
    i = job.my_frames[len(job.my_frames) // 2]
    cam = cams[i]
    warper = isa.SphericalWarper(ctx, job.scale)
    roi = warper.warpRoi((cam["width"], cam["height"]), cam["K"], cam["R"])
    S = cam["width"] * cam["height"]
    P = roi[2] * roi[3]
    algo = 3 * S + 6 * P + 1 * P           # source read once, 16SC3 + mask written once
    dst, msk = warper.alloc_fused(roi)
    warper.warp_fused_timed(frames[i], cam["K"], cam["R"], roi, dst, msk, launches)        # warm-up
    us = sum(warper.warp_fused_timed(frames[i], cam["K"], cam["R"], roi, dst, msk, launches) for _ in range(3)) / 3.0
    t = us * 1e-6
    ach = algo / t / 1e9
    # HBM traffic per launch from the PMC passes of profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs; FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md prescribes): not collectable from inside
    # this process, so it is read from the committed summary when that belongs to this kernel build.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_warp_pmc.json")) as f:
            pm = json.load(f)
        if pm.get("frame_size") == [cam["width"], cam["height"]]:
            traffic = pm["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "hbm", "kernel": "warp_fused_kernel", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(ach / 8000.0, 4), "traffic": traffic, "algorithmic_bytes_per_launch": algo,
            "avg_launch_us": round(t * 1e6, 2), "launches": launches}


def cpu_baseline(cams, workload):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample:
    two adjacent frames of the same workload through every stage; per-frame and per-pair times are
    extrapolated to the full job (n frames, n(n-1)/2 pairs, panorama area)."""
    import oracle
    import synth
    n = len(cams)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # a container's CPU quota (cgroup v2) is the real core count available to the oracle's OpenMP team
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    cores = int(os.environ.get("OMP_NUM_THREADS", cores))
    os.environ["OMP_NUM_THREADS"] = str(cores)   # read by libgomp when the oracle library is first loaded
    a, b = n // 2 - 1, n // 2
    sub = [cams[a], cams[b]]
    fr = [synth.render_frame(c) for c in sub]
    W, H = sub[0]["width"], sub[0]["height"]
    t = {}
    t0 = time.perf_counter()
    orb = oracle.Orb(W, H)
    feats = []
    for f in fr:
        k, d = orb.run(f)
        feats.append(dict(img_w=W, img_h=H, xy=np.stack([k["x"], k["y"]], 1), desc=d))
    t["features"] = (time.perf_counter() - t0) / 2
    t0 = time.perf_counter()
    oracle.match_pair(feats[0], feats[1])
    t["pair_near"] = time.perf_counter() - t0
    # a pair without overlap costs more (RANSAC never becomes confident and runs all its iterations): time one as well
    far = cams[(b + n // 2) % n]
    kf, df = orb.run(synth.render_frame(far))
    ffar = dict(img_w=W, img_h=H, xy=np.stack([kf["x"], kf["y"]], 1), desc=df)
    t0 = time.perf_counter()
    oracle.match_pair(feats[0], ffar)
    t["pair_far"] = time.perf_counter() - t0
    scale = float(np.float32(sub[0]["K"][1, 1]))
    t0 = time.perf_counter()
    items = []
    for c, f in zip(sub, fr):
        K, R = c["K"].astype(np.float32), c["R"].astype(np.float32)
        img, tl = oracle.warp_spherical(f, scale, K, R)
        msk, _ = oracle.warp_spherical(np.full((H, W), 255, np.uint8), scale, K, R, oracle.INTER_NEAREST, oracle.BORDER_CONSTANT)
        items.append((img.astype(np.int16), msk, tl))
    t["warp"] = (time.perf_counter() - t0) / 2
    corners = [i[2] for i in items]
    sizes = [(i[0].shape[1], i[0].shape[0]) for i in items]
    # band count of the FULL job's panorama (the sample's own panorama is smaller)
    full_rois = [oracle.warp_roi(scale, W, H, c["K"].astype(np.float32), c["R"].astype(np.float32)) for c in cams]
    px, py, pw, ph = oracle.lib() and _result_roi(full_rois)
    _, bands, _ = oracle.blend_config(oracle.BLEND_MULTI_BAND, 5.0, pw, ph)
    bl = oracle.Blender(oracle.BLEND_MULTI_BAND, bands, 0.0)
    bl.prepare(corners, sizes)
    t0 = time.perf_counter()
    for img, msk, tl in items:
        bl.feed(img, msk, tl)
    t["feed"] = (time.perf_counter() - t0) / 2
    _, _, sw, sh, _, _ = bl.roi()
    t0 = time.perf_counter()
    bl.blend()
    t["finalize_sample"] = time.perf_counter() - t0
    fin = t["finalize_sample"] * (pw * ph) / float(sw * sh)
    # pairs whose warped ROIs intersect are costed like the adjacent sample pair, the others like the distant one
    def overlap(r1, r2):
        return r1[0] < r2[0] + r2[2] and r2[0] < r1[0] + r1[2] and r1[1] < r2[1] + r2[3] and r2[1] < r1[1] + r1[3]
    n_near = sum(1 for i in range(n) for j in range(i + 1, n) if overlap(full_rois[i], full_rois[j]))
    n_far = n * (n - 1) // 2 - n_near
    total = n * (t["features"] + t["warp"] + t["feed"]) + n_near * t["pair_near"] + n_far * t["pair_far"] + fin
    return {"value": round(n / total, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "frames %d,%d (+ a distant one) of %s through every oracle stage; extrapolated: %d x (features %.2fs + warp %.2fs + feed %.2fs) + "
                      "%d overlapping pairs x %.3fs + %d other pairs x %.3fs + finalize %.2fs (scaled by panorama area) = %.1fs"
                      % (a, b, workload, n, t["features"], t["warp"], t["feed"], n_near, t["pair_near"], n_far, t["pair_far"], fin, total)}


def _result_roi(rois):
    x0 = min(r[0] for r in rois); y0 = min(r[1] for r in rois)
    x1 = max(r[0] + r[2] for r in rois); y1 = max(r[1] + r[3] for r in rois)
    return x0, y0, x1 - x0, y1 - y0


if __name__ == "__main__":
    main()
