"""One SIFT detect of an 8K frame, a few times (for the PMC passes of tools/pmc_sift.sh): python tools/sift_single.py [n]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
w, h = 7680, 4320
fr = synth.render_frame_gpu(synth.make_camera(w, h, 60.0, 15.0))
f = isa.SiftFeatureFinder(ctx, (w, h))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    f.detect(fr)
torch.cuda.synchronize()
f.close()
