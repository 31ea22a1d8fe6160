"""One ORB detect at a time on a 4K frame (no lane overlap): per-kernel durations under rocprofv3.  python tools/orb_single.py [reps]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
cam = synth.make_camera(3840, 2160, 60.0, 15.0)
fr = synth.render_frame_gpu(cam)
f = isa.OrbFeatureFinder(ctx, (3840, 2160))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f.detect(fr); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    f.detect(fr)
torch.cuda.synchronize()
print("single-lane ORB on 4K: %.1f us per frame" % ((time.perf_counter() - t0) / reps * 1e6))
