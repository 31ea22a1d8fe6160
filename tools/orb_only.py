import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
cam = synth.workload("config3")[8]
frame = synth.render_frame_gpu(cam)
f = isa.OrbFeatureFinder(ctx, (3840, 2160))
for _ in range(6):
    try:
        ft = f.detect(frame)
    except Exception as e:
        print("err", e)
print(len(ft) if 'ft' in dir() else -1)
