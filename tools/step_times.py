"""Per-step wall time of a job (diagnostics): python tools/step_times.py [steps] [config3|config5]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
if os.environ.get("SWITCH"): sys.setswitchinterval(float(os.environ["SWITCH"]))
ctx = isa.Context(0)
wl = sys.argv[2] if len(sys.argv) > 2 else "config3"
cams = synth.workload(wl)
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
size = (cams[0]["width"], cams[0]["height"]) if "width" in cams[0] else ((7680, 4320) if wl == "config5" else (3840, 2160))
job = StitchJob(ctx, size, cams, config=isa.StitchConfig.hot_path(features_type="sift" if wl == "config5" else "orb"))
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
