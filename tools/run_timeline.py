"""Host timeline of StitchJob.run (config 3): where the wall time between the stages goes.  python tools/run_timeline.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
if len(sys.argv) > 1: sys.setswitchinterval(float(sys.argv[1]))
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        T.setdefault(label + ".in", []).append(time.perf_counter())
        r = f(*a, **k)
        T.setdefault(label + ".out", []).append(time.perf_counter())
        return r
    setattr(obj, name, g)
wrap(job, "stage_features", "features")
wrap(job, "stage_gather", "gather")
wrap(job, "stage_match", "match")
wrap(job.engine, "match", "match_call")
wrap(job, "stage_prune", "prune")
wrap(job, "_compose_on_side_stream", "compose_thread")
wrap(job, "stage_reduce", "reduce")
wrap(job, "stage_finalize", "finalize")
wrap(job.engine, "sync", "final_sync")
for _ in range(4): job.run(frames)
T.clear()
N = 10
t_runs = []
for _ in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter(); job.run(frames); torch.cuda.synchronize(); t_runs.append((t0, time.perf_counter()))
import numpy as np
def rel(label):
    return np.mean([T[label][k] - t_runs[k][0] for k in range(N)]) * 1e3
print("step %.2f ms" % (np.mean([b - a for a, b in t_runs]) * 1e3))
for l in ("features.in", "features.out", "gather.out", "compose_thread.in", "match.in", "match_call.in", "match_call.out", "match.out", "prune.out", "compose_thread.out", "reduce.in", "finalize.in", "finalize.out", "final_sync.in", "final_sync.out"):
    if l in T: print("%-22s %7.2f ms" % (l, rel(l)))
