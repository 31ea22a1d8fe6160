"""Per-step wall time of the config-3 job (diagnostics): python tools/step_times.py [steps]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ts = []
for k in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    job.run(frames)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms:", " ".join("%.2f" % t for t in ts))
bd = job.breakdown(frames, reps=5)
print("breakdown sum %.2f ms:" % sum(bd.values()), {k: round(v, 2) for k, v in bd.items()})
ts = []
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    job.run(frames)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms after:", " ".join("%.2f" % t for t in ts))
ts = []
out = None
for k in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = job.run(frames)          # the previous result stays alive during the run, as in bench.py
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms, result kept alive:", " ".join("%.2f" % t for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(8):
    out = job.run(frames)
torch.cuda.synchronize()
print("8 steps back to back, no sync between: %.2f ms per step" % ((time.perf_counter() - t0) * 1e3 / 8))
