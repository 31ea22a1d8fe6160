"""Look for step-time outliers: python tools/spikes.py [steps]"""
import sys, os, time, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gc_events = []
gc.callbacks.append(lambda phase, info: gc_events.append((time.perf_counter(), phase, info.get("generation"))))
ts = []
out = None
for k in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = job.run(frames)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g = [e for e in gc_events if t0 <= e[0] <= t1 and e[1] == "start"]
    ts.append((t1 - t0) * 1e3)
    st = torch.cuda.memory_stats()
    print("step %2d %.2f ms  gc gens %s  torch allocs %d  reserved %.0f MB" % (k, ts[-1], [e[2] for e in g], st.get("num_device_alloc", 0), st["reserved_bytes.all.current"] / 1e6), flush=True)
