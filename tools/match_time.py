"""Where the match stage's wall time goes: the C call alone vs the Python wrapper (config 3)."""
import ctypes as C, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd import _capi as capi
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
feats = job.stage_features(frames)
n = len(feats)
m = job.engine.matcher
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    arr = (capi.MisFeatures * n)()
    for k, f in enumerate(feats):
        C.memmove(C.byref(arr[k]), C.byref(f.raw), C.sizeof(capi.MisFeatures))
    mis = (capi.MisMatchesInfo * (n * n))()
    t1 = time.perf_counter()
    ctx.check(ctx.lib.mis_match_all_pairs(ctx.h, arr, n, C.byref(m.params), mis))
    t2 = time.perf_counter()
    from image_stitching_amd.stitching import _unpack_mi
    out = [_unpack_mi(mis[i]) for i in range(n * n)]
    t3 = time.perf_counter()
    import numpy as np
    conf = torch.from_numpy(np.array([mis[i].confidence for i in range(n * n)], np.float64)).view(n, n)
    ctx.lib.mis_matches_free(mis, n * n)
    t4 = time.perf_counter()
    idx = job.stage_prune(conf)
    t5 = time.perf_counter()
    print("prep %.2f  C call %.2f  unpack %.2f  free %.2f  conf+prune %.2f  total %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)))
